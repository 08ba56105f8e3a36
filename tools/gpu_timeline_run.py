"""Workload for a rocprofv3 --kernel-trace timeline: a few factorize / gradient / predict calls
at one size (argv: N d what), the last of each is what tools/timeline_dump.py looks at."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402

from gumbi_amd import engine  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4
what = sys.argv[3] if len(sys.argv) > 3 else "factorize"
X, y, ls = O.synthetic_table(N, d)
Xs = O.synthetic_grid(d)
e = engine.Engine(0)
e.set_data(X, y)
e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
import os  # noqa: E402

theta = np.concatenate([ls, [1.0, 0.2]])
if os.environ.get("TL_CHOL"):
    e.set_chol_scheme(int(os.environ["TL_CHOL"]))
if os.environ.get("TL_GRAD"):
    e.set_grad_scheme(int(os.environ["TL_GRAD"]))
for _ in range(3):
    if what == "evaluate":  # what find_MAP calls: ONE C call per objective + gradient evaluation
        e.evaluate(theta)
        continue
    e.factorize()
    if what == "grad":
        e.nlml(grad=True)
    if what == "predict":
        e.predict(Xs)
e.close()
