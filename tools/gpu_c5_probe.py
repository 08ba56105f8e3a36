"""C5-size problem (N = 100,000, d = 8, RBF-ARD, M = 10^4) on ONE MI355X at fixed hyper-parameters:
factorisation, grid prediction and one MAP objective+gradient evaluation, with the size-independent
checks the parity suite uses (with_noise shifts the variance by exactly sigma^2; L L^T = Sigma on
sampled rows).  Writes gpurun_out/c5_single_gpu.json."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from gumbi_amd import engine as E  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = 8
X, y, ls = bench.synthetic_table(N, d)
Xs = bench.synthetic_grid(d, 100)
theta = np.concatenate([ls, [1.0, 0.2]])
eng = E.Engine(0)
eng.set_data(X, y)
eng.set_kernel(E.KernelSpec(D=d, idx_cont=list(range(d)), kind="ExpQuad"))
eng.set_theta(theta)
out = {"N": N, "d": d, "M": len(Xs)}
eng.factorize()  # warm-up: allocates the 80 GB factor buffer
t0 = time.perf_counter(); eng.factorize(); t1 = time.perf_counter()
out["factorize_s"] = round(t1 - t0, 3)
out["factorize_tflops"] = round(N**3 / 3 / (t1 - t0) / 1e12, 2)
out["nlml"] = eng.nlml()
t0 = time.perf_counter(); mu, var = eng.predict(Xs); t1 = time.perf_counter()
out["predict_s"] = round(t1 - t0, 3)
out["predict_tflops"] = round(N * N * len(Xs) / (t1 - t0) / 1e12, 2)
mu2, var2 = eng.predict(Xs[:512], with_noise=False)
out["with_noise_shift_err"] = float(np.max(np.abs((var[:512] - var2) - 0.2**2)))
rows = np.array([0, 1, N // 3, N // 2, N - 2, N - 1])
L = eng.copy_factor(int(N - 300), 300, 0, N)            # last 300 rows of the factor (row-major)
L = np.tril(L, k=N - 300)                                # only the lower triangle is meaningful
S_rows = (L @ L.T)[-3:, -3:]                             # Sigma[N-3:, N-3:] from the factor
from oracle import gp_oracle as O  # noqa: E402  (checker only)
spec = O.make_spec(d, range(d))
S_ref = O.sigma_matrix(spec, theta, X[-3:], dist_mode="direct")
out["LLt_vs_sigma_err"] = float(np.max(np.abs(S_rows - S_ref)))
eng.factorize(); eng.nlml(grad=True)  # warm-up: allocates the 80 GB inverse buffer
t0 = time.perf_counter(); eng.factorize(); val, g = eng.nlml(grad=True); t1 = time.perf_counter()
out["map_eval_s"] = round(t1 - t0, 3)
out["map_eval_tflops"] = round(N**3 / (t1 - t0) / 1e12, 2)
out["grad_finite"] = bool(np.all(np.isfinite(g)))
out["finite"] = bool(np.all(np.isfinite(mu)) and np.all(var > 0))
Path("gpurun_out").mkdir(exist_ok=True)
Path("gpurun_out/c5_single_gpu.json").write_text(json.dumps(out))
print(json.dumps(out))
