"""Soak of the persistent evaluation launch: many gmb_evaluate calls at several sizes and kernels, every result compared bit for
bit with the first one of its series (any lost flag would show as a time-out error, any race as different bits)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O

reps = int(os.environ.get('SOAK_REPS', '150'))
cases = [(392, 1, 'ExpQuad'), (1000, 3, 'Matern52'), (1024, 2, 'Matern32'), (2560, 4, 'ExpQuad'), (5200, 4, 'Matern52'), (10000, 4, 'ExpQuad'), (16384, 8, 'Matern52')]
t_all = time.time()
for N, d, kind in cases:
    X, y, ls = O.synthetic_table(N, d, seed=N)
    spec = O.make_spec(d, range(d), kind=kind)
    theta = O.pack_theta(spec, ls, 1.0, 0.25)
    e = engine.Engine(0)
    e.set_data(X, y)
    e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)), kind=kind))
    val0, g0 = e.evaluate(theta)
    a0 = e.copy_alpha()
    n = max(10, min(reps, int(reps * 2560 / N)))
    bad = 0
    t0 = time.time()
    for i in range(n):
        val, g = e.evaluate(theta * (1.0 + 0.0 * i))
        if np.float64(val).tobytes() != np.float64(val0).tobytes() or g.tobytes() != g0.tobytes():
            bad += 1
        if i % 7 == 0 and e.copy_alpha().tobytes() != a0.tobytes():
            bad += 1
    print(f"N={N} d={d} {kind}: {n} evaluations in {time.time() - t0:.1f} s, {bad} differing results", flush=True)
    # the other two persistent launches (tile Cholesky alone, tile solve of a prediction with no fit before it) and the GEMM-form
    # prediction behind an evaluation: the same bits every time
    Xs = np.random.default_rng(N).standard_normal((4200, d))
    e.factorize()
    mu0, var0 = e.predict(Xs)
    L0 = e.copy_factor(max(0, N - 300), min(300, N), 0, min(300, N)).tobytes()
    e.evaluate(theta)
    mug0, varg0 = e.predict(Xs)
    bad2, m = 0, max(5, n // 6)
    for i in range(m):
        e.factorize()
        mu, var = e.predict(Xs)
        bad2 += int(mu.tobytes() != mu0.tobytes() or var.tobytes() != var0.tobytes() or e.copy_factor(max(0, N - 300), min(300, N), 0, min(300, N)).tobytes() != L0)
        e.evaluate(theta)
        mug, varg = e.predict(Xs)
        bad2 += int(mug.tobytes() != mug0.tobytes() or varg.tobytes() != varg0.tobytes())
    print(f"    factorize + predict (solve form) and evaluate + predict (GEMM form): {m} rounds, {bad2} differing results; forms agree to "
          f"{np.max(np.abs(mug0 - mu0)) / np.max(np.abs(mu0)):.1e} (mean) / {np.max(np.abs(varg0 - var0)):.1e} (variance)", flush=True)
    e.close()
print(f"soak done in {time.time() - t_all:.0f} s")
