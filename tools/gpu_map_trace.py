import sys; sys.path.insert(0, '.')
import numpy as np
import bench
cfg = dict(bench.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "c3"]); cfg["N"] = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
for jac in (False, True):
    gp = bench.build_gp(cfg, 0)
    gp.map_includes_jacobian = jac
    gp.find_MAP(maxeval=200)
    r = gp.opt_result
    print(f"jacobian={jac}: evals {gp.n_eval} nit {r.nit} status {r.status} msg {r.message}  f {r.fun:.6f}  |proj g| {np.max(np.abs(r.jac)):.3e}")
    print("   theta", np.round(gp._theta_fitted, 4), " nlml trace", np.round(gp.nlml_trace[:12], 3))
    gp.engine.close()
