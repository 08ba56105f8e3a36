"""Where does one bench step (find_MAP + predict at C2) spend its wall time?  Wraps the engine
calls of the MAP objective with host timers (each call synchronises, so host time = device time
+ launch / synchronisation overhead)."""
import sys
import time
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
gp = bench.build_gp(cfg, 0)
eng = gp.engine
acc = defaultdict(float)
cnt = defaultdict(int)


def wrap(obj, name):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] += time.perf_counter() - t0
            cnt[name] += 1

    setattr(obj, name, timed)


for nm in ("set_theta", "factorize", "nlml"):
    wrap(eng, nm)
wrap(gp, "_objective")
wrap(gp, "_log_prior")
gp.find_MAP(maxeval=200)  # warm-up
for rep in range(2):
    acc.clear()
    cnt.clear()
    t0 = time.perf_counter()
    gp.find_MAP(maxeval=200)
    total = time.perf_counter() - t0
    print(f"find_MAP {total*1e3:.1f} ms, {gp.n_eval} evaluations")
    for k in ("_objective", "set_theta", "factorize", "nlml", "_log_prior"):
        print(f"  {k:12s} n={cnt[k]:3d}  total {acc[k]*1e3:8.1f} ms  mean {acc[k]/max(cnt[k],1)*1e3:7.3f} ms")
    print(f"  outside _objective (scipy L-BFGS-B + final factorize): {(total - acc['_objective'])*1e3:.1f} ms")
for prof in (False, True):
    eng.set_profiling(prof)
    for _ in range(2):
        t0 = time.perf_counter(); eng.factorize(); t1 = time.perf_counter(); eng.nlml(grad=True); t2 = time.perf_counter()
    tm = eng.timings()
    print(f"profiling={prof}: factorize {1e3*(t1-t0):.2f} ms (kbuild {tm['kbuild_ms']:.2f} chol {tm['chol_ms']:.2f})  nlml+grad {1e3*(t2-t1):.2f} ms (grad_ms {tm['grad_ms']:.2f})")
