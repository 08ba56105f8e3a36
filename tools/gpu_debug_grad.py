import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O

for N in (100, 128, 200, 300, 700):
    d = 3
    X, y, ls = O.synthetic_table(N, d, seed=3)
    spec = O.make_spec(d, range(d))
    theta = O.pack_theta(spec, ls, 1.2, 0.25)
    eng = engine.Engine(0)
    eng.set_data(X, y); eng.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)))); eng.set_theta(theta)
    eng.factorize()
    val, g = eng.nlml(grad=True)
    Z = eng.copy_factor()
    S = O.sigma_matrix(spec, theta, X, "direct")
    Zr = np.linalg.inv(S)
    errZ = np.max(np.abs(np.tril(Z) - np.tril(Zr))) / np.max(np.abs(Zr))
    vr, gr = O.nlml_and_grad(spec, theta, X, y)
    print(N, "Z rel err", errZ, "grad", g, "ref", gr)
    if errZ > 1e-8:
        E = np.abs(np.tril(Z) - np.tril(Zr)) > 1e-8 * np.max(np.abs(Zr))
        nb = (N + 127) // 128
        print("  bad tiles:", [(i, j, int(E[i*128:(i+1)*128, j*128:(j+1)*128].sum())) for i in range(nb) for j in range(i + 1) if E[i*128:(i+1)*128, j*128:(j+1)*128].any()])
    eng.close()
