"""Replay the reference's Simple_Regression notebook (docs/source/notebooks/examples/
Simple_Regression.pct.py:33-71) with the HIP backend and print what the notebook printed."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402

import gumbi_amd as gmb  # noqa: E402

df = pd.read_pickle(ROOT / "tests" / "golden" / "example_dataset.pkl").query('Metric=="mean"')
outputs = ["a", "b", "c", "d", "e", "f"]
log_vars = ["Y", "b", "c", "d", "f"]
logit_vars = ["X", "e"]
ds = gmb.DataSet(df, outputs=outputs, log_vars=log_vars, logit_vars=logit_vars)
ds.tidy = ds.tidy[ds.tidy.Color.isin(["cyan", "magenta"]) & (ds.tidy.Pair == "burrata+barbaresco")]
gp = gmb.GP(ds, outputs=["d"])
gp.fit(continuous_dims=["X", "Y", "lg10_Z"], linear_dims=["X", "Y", "lg10_Z"])
print("N =", len(gp.model.y), "evals", gp.n_eval)
print({k: np.round(v, 5) for k, v in gp.MAP.items() if not k.endswith("_log__")})
point = gp.parray(lg10_Z=8, X=0.5, Y=88)
pred = gp.predict_points(point)
print("point:", float(np.asarray(pred.μ).ravel()[0]), float(np.asarray(pred.σ2).ravel()[0]), " notebook: 0.7526282 0.00204789")
gp.prepare_grid(at=gp.parray(lg10_Z=8, X=0.5))
gp.predict_grid()
y = gp.predictions
print("grid[:10] mu ", np.round(np.asarray(y.μ).ravel()[:10], 8))
print("notebook  mu  [0.95353955 0.94923129 0.94544874 0.94220088 0.93948256 0.93727268 0.93553307 0.93420812 0.93322533 0.93249681]")
print("grid[:10] s2 ", np.round(np.asarray(y.σ2).ravel()[:10], 8))
print("notebook  s2  [0.02777067 0.02648205 0.02492182 0.02307904 0.02096868 0.01863859 0.01617249 0.01368664 0.01131927 0.009213]")

# sensitivity of the optimum to the starting point (is the posterior multimodal at the 1% level?)
rng = np.random.default_rng(0)
th0 = gp._initial_theta()
pos = gp._positive_mask()
best = []
for trial in range(8):
    start = th0.copy()
    start[pos] = start[pos] * np.exp(rng.normal(0, 0.7, pos.sum()))
    start[~pos] = start[~pos] + rng.normal(0, 0.5, (~pos).sum())
    u0 = start.copy(); u0[pos] = np.log(start[pos])
    from scipy.optimize import minimize
    res = minimize(gp._objective, u0, args=(pos,), jac=True, method="L-BFGS-B", options={"maxfun": 500})
    th = np.where(pos, np.exp(res.x), res.x)
    gp.engine.set_theta(th); gp.engine.factorize(); gp._theta_fitted = th; gp.MAP = gp._theta_to_dict(th)
    p = gp.predict_points(point)
    print(f"trial {trial}: obj {res.fun:.6f}  point mu {float(np.asarray(p.μ).ravel()[0]):.6f} s2 {float(np.asarray(p.σ2).ravel()[0]):.6f}  ls {np.round(th[:3],4)} eta {th[3]:.4f} sigma {th[4]:.4f}")

# hypothesis: pm.find_MAP (PyMC >= 4) optimises logp WITHOUT the transforms' Jacobians
def obj_nojac(u, pos):
    f, g = gp._objective(u, pos)
    f = f + np.sum(u[pos])          # remove the  -sum(u)  Jacobian term
    g = g.copy(); g[pos] += 1.0
    return f, g
u0 = th0.copy(); u0[pos] = np.log(th0[pos])
res = minimize(obj_nojac, u0, args=(pos,), jac=True, method="L-BFGS-B", options={"maxfun": 500})
th = np.where(pos, np.exp(res.x), res.x)
gp.engine.set_theta(th); gp.engine.factorize(); gp._theta_fitted = th; gp.MAP = gp._theta_to_dict(th)
p = gp.predict_points(point)
print("NO-JACOBIAN MAP: point", float(np.asarray(p.μ).ravel()[0]), float(np.asarray(p.σ2).ravel()[0]), " notebook: 0.7526282 0.00204789")
gp.prepare_grid(at=gp.parray(lg10_Z=8, X=0.5)); gp.predict_grid()
print("grid mu", np.round(np.asarray(gp.predictions.μ).ravel()[:10], 8))
print("grid s2", np.round(np.asarray(gp.predictions.σ2).ravel()[:10], 8))
print({k: np.round(v, 5) for k, v in gp.MAP.items() if not k.endswith("_log__")})
