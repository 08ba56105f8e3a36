"""Soak of the persistent tile Cholesky: many factorisations at ragged sizes, gradient / prediction launches in between (other
cache states), optionally a second process loading the same GPU (CT_SOAK_PEER=1: uneven load, the case that exposes a missing
release / acquire); every run must reproduce the first run's bits and agree with the recursion to rounding."""
import hashlib, os, subprocess, sys, time
sys.path.insert(0, '.')
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O

if os.environ.get("CT_SOAK_ROLE") == "peer":  # background load: big GEMM-heavy factorisations of another size
    X, y, ls = O.synthetic_table(9000, 4, seed=1)
    e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=4, idx_cont=[0, 1, 2, 3])); e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
    t_end = time.time() + float(os.environ.get("CT_SOAK_SECONDS", "40"))
    n = 0
    while time.time() < t_end:
        e.factorize(); e.nlml(grad=True); n += 1
    print("peer did", n, "evaluations", flush=True)
    sys.exit(0)

peer = None
if os.environ.get("CT_SOAK_PEER") == "1":
    peer = subprocess.Popen([sys.executable, __file__], env=dict(os.environ, CT_SOAK_ROLE="peer"))
    time.sleep(8)
reps = int(os.environ.get("CT_SOAK_REPS", "25"))
bad = 0
for N, d in [(2100, 3), (3333, 4), (5200, 4), (6016, 2), (7777, 5), (10000, 4), (12345, 3)]:
    X, y, ls = O.synthetic_table(N, d, seed=N)
    Xs = np.random.default_rng(0).standard_normal((500, d))
    e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)), kind="Matern52"))
    e.set_theta(np.concatenate([ls, [1.1, 0.2]]))
    e.set_chol_scheme(0); e.factorize(); L0 = np.tril(e.copy_factor(r0=N - 600, nr=600), N - 600); nl0 = e.nlml()
    e.set_chol_scheme(3)
    first = None
    t0 = time.time()
    for rep in range(reps):
        e.factorize()
        L = np.tril(e.copy_factor(r0=N - 600, nr=600), N - 600)
        key = (hashlib.sha1(L.tobytes()).hexdigest(), e.copy_v().tobytes(), np.float64(e.nlml()).tobytes())
        if first is None:
            first = key
            assert np.max(np.abs(L - L0)) / np.max(np.abs(L0)) < 1e-12 and abs(e.nlml() - nl0) < 1e-9 * abs(nl0), "tile kernel vs recursion"
        elif key != first:
            bad += 1
            print(f"N={N} rep {rep}: DIFFERENT BITS", flush=True)
        if rep % 3 == 1:
            e.nlml(grad=True)   # consumes / restores the factor, leaves other lines in the caches
        if rep % 3 == 2:
            e.predict(Xs)
    print(f"N={N}: {reps} factorisations, {bad} mismatches so far, {1e3 * (time.time() - t0) / reps:.1f} ms per round trip", flush=True)
    e.close()
if peer is not None:
    peer.wait()
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
