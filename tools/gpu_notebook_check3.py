"""Which variant of the model reproduces Simple_Regression.ipynb's printed numbers?"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, pandas as pd
import gumbi_amd as gmb
df = pd.read_pickle(ROOT / "tests" / "golden" / "example_dataset.pkl").query('Metric=="mean"')
ds = gmb.DataSet(df, outputs=["a", "b", "c", "d", "e", "f"], log_vars=["Y", "b", "c", "d", "f"], logit_vars=["X", "e"])
ds.tidy = ds.tidy[ds.tidy.Color.isin(["cyan", "magenta"]) & (ds.tidy.Pair == "burrata+barbaresco")]
NB = np.array([0.95353955, 0.94923129, 0.94544874, 0.94220088, 0.93948256, 0.93727268, 0.93553307, 0.93420812, 0.93322533, 0.93249681])
for label, kw, jac in (("ARD, no jac", {}, False), ("ARD, jac", {}, True), ("non-ARD, no jac", {"ARD": False}, False), ("non-ARD, jac", {"ARD": False}, True)):
    gp = gmb.GP(ds, outputs=["d"])
    gp.map_includes_jacobian = jac
    gp.fit(continuous_dims=["X", "Y", "lg10_Z"], linear_dims=["X", "Y", "lg10_Z"], **kw)
    p = gp.predict_points(gp.parray(lg10_Z=8, X=0.5, Y=88))
    gp.prepare_grid(at=gp.parray(lg10_Z=8, X=0.5)); gp.predict_grid()
    mu = np.asarray(gp.predictions.μ).ravel()[:10]
    print(f"{label:18s} point {float(np.asarray(p.μ).ravel()[0]):.6f} {float(np.asarray(p.σ2).ravel()[0]):.6f} (nb 0.752628 0.002048)  grid max rel err {np.max(np.abs(mu-NB)/NB):.4f}  ls {np.round(gp.MAP['ls_total'],3)}")
    gp.engine.close()
