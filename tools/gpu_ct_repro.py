"""Bit-reproducibility probe of the persistent tile Cholesky: repeats one factorisation and reports where runs differ."""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O
N, d = int(os.environ.get('CT_N', '5400')), 4
X, y, ls = O.synthetic_table(N, d, seed=8)
ref = None
for fresh in range(2):
    e = engine.Engine(0)
    e.set_data(X, y)
    e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
    e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
    e.set_chol_scheme(3)
    for rep in range(4):
        e.factorize()
        L = np.tril(e.copy_factor())
        v = e.copy_v()
        nl = e.nlml()
        if ref is None:
            ref = (L, v, nl)
            continue
        dL = np.abs(L - ref[0])
        bad = np.argwhere(dL > 0)
        msg = f"engine {fresh} rep {rep}: L differs in {len(bad)} entries (max {dL.max():.3e})"
        if len(bad):
            tiles = sorted({(int(r) // 128, int(c) // 128) for r, c in bad})
            msg += f"; first tiles {tiles[:8]} of {len(tiles)}; first entry {tuple(bad[0])}"
        msg += f"; v differs in {(v != ref[1]).sum()} (max {np.abs(v - ref[1]).max():.3e}); nlml diff {nl - ref[2]:.3e}"
        print(msg, flush=True)
    e.close()
