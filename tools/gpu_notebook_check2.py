"""Replay the reference's Multioutput_Regression notebook (run by its author on PyMC >= 5: its first
cell shows a pytensor warning) with the HIP backend, under both readings of pm.find_MAP's objective."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
from scipy.optimize import minimize  # noqa: E402

import gumbi_amd as gmb  # noqa: E402

df = pd.read_pickle(ROOT / "tests" / "golden" / "example_dataset.pkl")
df = df[(df.Name == "binary-pollen") & (df.Color == "cyan") & (df.Metric == "mean")]
ds = gmb.DataSet(df, outputs=["a", "b", "c", "d", "e", "f"], log_vars=["Y", "b", "c", "d", "f"], logit_vars=["X", "e"])
fit_params = ["a", "b", "c", "d", "e"]
NB_MU = np.array([[-9.59442479, 0.65605058, 0.00646403, 0.81416271, 0.15214448],
                  [-8.05656298, 0.66609041, 0.00635764, 0.81267686, 0.16440518],
                  [-6.40414117, 0.67787309, 0.00620618, 0.8105507, 0.17662809],
                  [-4.75033515, 0.68729924, 0.00617787, 0.81008143, 0.19510875],
                  [-2.94787273, 0.69658766, 0.00619329, 0.81021742, 0.21940875]])
NB_S2 = np.array([[0.01676639, 1.22016392e-04, 0.00134064, 1.22105406e-05, 2.80013388e-04],
                  [0.00462947, 1.03825281e-04, 0.00114775, 1.03319592e-05, 8.16338238e-05],
                  [0.00455407, 9.92785623e-05, 0.00110227, 9.91092000e-06, 8.03754540e-05],
                  [0.00462973, 1.04073081e-04, 0.00115023, 1.03548470e-05, 8.16411508e-05],
                  [0.01685376, 1.22594426e-04, 0.00134647, 1.22658105e-05, 2.81471085e-04]])


def report(gp, tag):
    gp.prepare_grid(limits=gp.parray(lg10_Z=[1, 9]), resolution=5)
    gp.predict_grid()
    mv = gp.predictions
    mu = np.stack([np.asarray(mv.get(p).μ).ravel() for p in fit_params], axis=1)
    s2 = np.stack([np.asarray(mv.get(p).σ2).ravel() for p in fit_params], axis=1)
    print(tag, "max rel err mu", np.max(np.abs(mu - NB_MU) / np.abs(NB_MU)), " max rel err s2", np.max(np.abs(s2 - NB_S2) / NB_S2))
    print(np.round(mu, 6))
    print(np.round(s2, 8))


gp = gmb.GP(ds, outputs=fit_params)
gp.fit(continuous_dims="lg10_Z", linear_dims="lg10_Z")
print("N", len(gp.model.y), "evals", gp.n_eval)
report(gp, "WITH-JACOBIAN ")
pos = gp._positive_mask()
th0 = gp._initial_theta()


def obj_nojac(u, pos):
    f, g = gp._objective(u, pos)
    g = g.copy()
    g[pos] += 1.0
    return f + np.sum(u[pos]), g


u0 = th0.copy()
u0[pos] = np.log(th0[pos])
res = minimize(obj_nojac, u0, args=(pos,), jac=True, method="L-BFGS-B", options={"maxfun": 5000})
th = np.where(pos, np.exp(res.x), res.x)
gp.engine.set_theta(th); gp.engine.factorize(); gp._theta_fitted = th; gp.MAP = gp._theta_to_dict(th)
print("evals", res.nfev)
report(gp, "NO-JACOBIAN   ")
