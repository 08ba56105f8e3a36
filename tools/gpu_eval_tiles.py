"""Persistent evaluation launch (csrc/eval_tiles.hpp) against the launch tree: gradient / alpha / NLML agreement with the
oracle and between the schedules, and wall time per evaluation.  ET_SIZES=10000x4,... picks the cases; ET_LAG the lag."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O

sizes = [tuple(int(v) for v in t.split('x')) for t in os.environ.get('ET_SIZES', '300x2,1000x3,1024x2,2500x4,5200x4,10000x4').split(',')]
schemes = [int(s) for s in os.environ.get('ET_SCHEMES', '0,1,2').split(',')]
reps = int(os.environ.get('ET_REPS', '5'))
check = os.environ.get('ET_ORACLE', '1') == '1'
kind = os.environ.get('ET_KIND', 'ExpQuad')
for N, d in sizes:
    X, y, ls = O.synthetic_table(N, d)
    spec = O.make_spec(d, range(d), kind=kind)
    theta = O.pack_theta(spec, ls, 1.0, 0.2)
    e = engine.Engine(0)
    e.set_data(X, y)
    e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)), kind=kind))
    ref = None
    if check and N <= 6000:
        val_r, grad_r = O.nlml_and_grad(spec, theta, X, y)
        ref = (val_r, grad_r)
    line = []
    first = None
    for scheme in schemes:
        e.set_grad_scheme(scheme)
        try:
            val, g = e.evaluate(theta, grad=True)
            alpha = e.copy_alpha()
            # the factor must still serve predictions
            Xs = np.random.default_rng(1).standard_normal((300, d))
            mu, var = e.predict(Xs)
        except Exception as ex:  # noqa: BLE001
            print(f"N={N} scheme {scheme}: FAILED {ex}", flush=True)
            continue
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            e.evaluate(theta, grad=True)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        tm = e.timings()
        s = f"scheme {scheme}: {best:.3f} ms/eval (chol_ms {tm['chol_ms']:.3f} grad_ms {tm['grad_ms']:.3f})"
        if ref is not None:
            s += f" g-vs-oracle {np.max(np.abs(g - ref[1])) / max(1.0, np.max(np.abs(ref[1]))):.1e} nlml {abs(val - ref[0]) / abs(ref[0]):.1e}"
        if first is None:
            first = (val, g, alpha, mu, var)
        else:
            s += (f" dg {np.max(np.abs(g - first[1])) / max(1.0, np.max(np.abs(first[1]))):.1e} dalpha {np.max(np.abs(alpha - first[2])) / np.max(np.abs(first[2])):.1e}"
                  f" dmu {np.max(np.abs(mu - first[3])) / np.max(np.abs(first[3])):.1e} dvar {np.max(np.abs(var - first[4])):.1e}")
        line.append(s)
    print(f"N={N} d={d}: " + " | ".join(line), flush=True)
    # separate calls: factorize, then nlml(grad) -- the launch behind a final factor
    e.set_grad_scheme(-1)
    e.set_theta(theta)
    e.factorize()
    v2, g2 = e.nlml(grad=True)
    mu2, var2 = e.predict(Xs)
    print(f"   factorize + nlml(grad): dg {np.max(np.abs(g2 - first[1])) / max(1.0, np.max(np.abs(first[1]))):.1e} dmu {np.max(np.abs(mu2 - first[3])) / np.max(np.abs(first[3])):.1e}", flush=True)
    e.close()
