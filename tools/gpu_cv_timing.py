"""Wall time of GP.cross_validate (k independent fits + predictions) against one fit of the same table."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, pandas as pd
import gumbi_amd as gmb
from oracle import gp_oracle as O
N, d = int(sys.argv[1]), 3
X, y, ls = O.synthetic_table(N, d, seed=2)
cols = [f"x{i}" for i in range(d)]
df = pd.DataFrame(X, columns=cols); df["y"] = y
ds = gmb.DataSet(df, outputs=["y"])
gp = gmb.GP(ds, outputs=["y"])
gp.fit(continuous_dims=cols, continuous_kernel="ExpQuad")
for rep in range(2):
    t0 = time.perf_counter(); gp.fit(continuous_dims=cols, continuous_kernel="ExpQuad"); t1 = time.perf_counter()
    cv = gp.cross_validate(n_train=int(0.8 * N), seed=1 + rep)
    t2 = time.perf_counter()
    print(f"N={N}: one fit {1e3*(t1-t0):.1f} ms ({gp.n_eval} evaluations); one cross-validation split (fit on 80 %, predict 20 %) {1e3*(t2-t1):.1f} ms")
