"""C1 (BASELINE.json configs[0], the reference's own CPU-runnable case: mpg ~ horsepower, N = 392, d = 1)
end to end through the front end on the synthetic stand-in of SURVEY.md section 8d: DataSet ->
GP.fit (MAP) -> prepare_grid -> predict_grid, wall clock, and the per-evaluation latency."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, pandas as pd
import gumbi_amd as gmb

rng = np.random.default_rng(2021)
hp = np.exp(rng.normal(4.6, 0.35, 392))
mpg = np.exp(7.1 - 0.85 * np.log(hp) + rng.normal(0, 0.12, 392))
df = pd.DataFrame({"horsepower": hp, "mpg": mpg})
for rep in range(3):
    t0 = time.perf_counter()
    ds = gmb.DataSet(df, outputs=["mpg"], log_vars=["mpg", "horsepower"])
    gp = gmb.GP(ds)
    t1 = time.perf_counter()
    gp.fit(outputs=["mpg"], continuous_dims=["horsepower"])
    t2 = time.perf_counter()
    gp.prepare_grid()
    y = gp.predict_grid()
    t3 = time.perf_counter()
    print(f"rep {rep}: dataset {1e3*(t1-t0):.1f} ms  fit {1e3*(t2-t1):.1f} ms ({gp.n_eval} evaluations, {1e3*(t2-t1)/max(gp.n_eval,1):.2f} ms each)  "
          f"grid+predict {1e3*(t3-t2):.1f} ms  total {1e3*(t3-t0):.1f} ms")
