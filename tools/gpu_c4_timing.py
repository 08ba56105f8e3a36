"""C4 (2 outputs x N = 20k, d = 4, ExpQuad, heteroskedastic output noise) through the Kronecker engine: wall time of
one MAP evaluation (P factorisations + gradients) and of a grid prediction, beside the single-output engine on
ONE of the N x N systems (the floor: P x that).   python tools/gpu_c4_timing.py [N]"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import gp_oracle as O
from gumbi_amd.engine import Engine, KernelSpec
from gumbi_amd.regression.icm import IcmEngine
from test_gpu_configs import icm_problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
d = 4
X, y, spec, theta = icm_problem(n, d)
ks = KernelSpec(D=d + 1, idx_cont=list(range(d)), kind="ExpQuad", out_col=d, n_out=2, hetero_noise=True)
kron = IcmEngine(0); kron.set_data(X, y); kron.set_kernel(ks); kron.set_theta(theta)
Xs1 = O.synthetic_grid(d, res=100)
Xs = np.vstack([np.column_stack([Xs1, np.full(len(Xs1), p)]) for p in range(2)])
kron.factorize(); kron.nlml(grad=True); kron.predict(Xs[:256])
for rep in range(3):
    th = theta * (1.0 + 0.01 * rep)
    t0 = time.perf_counter(); kron.set_theta(th); kron.factorize(); v, g = kron.nlml(grad=True); t1 = time.perf_counter()
    mu, var = kron.predict(Xs); t2 = time.perf_counter()
    print(f"kron  N={n}: MAP evaluation {1e3*(t1-t0):.1f} ms ({2*float(n)**3/(t1-t0)/1e12:.1f} TF/s)   predict 2 x 10^4 points {1e3*(t2-t1):.1f} ms   nlml {v:.6f}")
e = Engine(0); e.set_data(np.ascontiguousarray(X[:n, :d]), y[:n]); e.set_kernel(KernelSpec(D=d, idx_cont=list(range(d)), kind="ExpQuad"))
e.set_theta(np.concatenate([theta[:d], [1.0, 0.3]])); e.factorize(); e.nlml(grad=True)
for rep in range(3):
    t0 = time.perf_counter(); e.factorize(); e.nlml(grad=True); t1 = time.perf_counter()
    print(f"single N={n}: factorise + gradient {1e3*(t1-t0):.1f} ms ({float(n)**3/(t1-t0)/1e12:.1f} TF/s)")
