"""Task trace of the persistent evaluation launch: where the time of a fused evaluation goes, by task kind.
ET_N, ET_D, ET_SCHEME (2 = fused, 1 = behind the factorisation), ET_LAG; ET_DUMP=1 lists every task (small launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O

N, d = int(os.environ.get('ET_N', '10000')), int(os.environ.get('ET_D', '4'))
scheme = int(os.environ.get('ET_SCHEME', '2'))
lag = int(os.environ.get('ET_LAG', '-1'))
X, y, ls = O.synthetic_table(N, d)
spec = O.make_spec(d, range(d), kind='ExpQuad')
theta = O.pack_theta(spec, ls, 1.0, 0.2)
e = engine.Engine(0)
e.set_data(X, y)
e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
e.set_grad_scheme(scheme, lag)
for _ in range(3):
    e.evaluate(theta)
e.chol_task_trace(1)
e.evaluate(theta)
tr = e.eval_task_trace(with_chol=(scheme != 1), lag=lag)
e.chol_task_trace(0)
if tr is None:
    print("no trace"); sys.exit(0)
tasks, st = tr
st = st * 1e6
nct = (N + 127) // 128
end = st.max()
print(f"N={N}: {len(tasks)} tasks, launch span {end:.0f} us")
names = {0: 'CHOL', 1: 'INV', 2: 'ZZ', 3: 'FIN'}
if os.environ.get('ET_DUMP'):  # small launches: every task -- taken / contraction done / solve input ready / published (us)
    for (k, I, J), t in sorted(zip(tasks.tolist(), st.tolist()), key=lambda kt: kt[1][3]):
        print(f"  {names[k]:4s} ({I},{J})  taken {t[0]:7.1f}  contracted {t[1]:7.1f}  input {t[2]:7.1f}  published {t[3]:7.1f}")
for k in (0, 1, 2):
    m = tasks[:, 0] == k
    if not m.any():
        continue
    I, J = tasks[m, 1], tasks[m, 2]
    s = st[m]
    nkb = np.where(k == 0, J, np.where(k == 1, J - I, nct - I))
    dur = s[:, 3] - s[:, 0]
    con = s[:, 1] - s[:, 0]
    ok = nkb > 0
    print(f"{names[k]}: {m.sum()} tasks, first taken {s[:,0].min():.0f} us, last published {s[:,3].max():.0f} us, sum of task time {dur.sum()/1e3:.1f} ms"
          f" ({dur.sum()/256/1e3:.2f} ms per CU), k-blocks {nkb.sum()}, contraction us per k-block: median {np.median(con[ok]/nkb[ok]):.2f} "
          f"weighted {con[ok].sum()/nkb[ok].sum():.2f}; post-contraction per task median {np.median(s[:,3]-s[:,1]):.1f} us")
    # long tasks only (>= 20 k-blocks): rate without the per-task overhead
    big = nkb >= 20
    if big.any():
        print(f"     tasks with >= 20 k-blocks: {con[big].sum()/nkb[big].sum():.2f} us per k-block; wait for solve input median {np.median((s[:,2]-s[:,1])[big]):.1f} us")
# utilisation by tenths of the launch: busy = inside a task
edges = np.linspace(0, end, 11)
busy = []
for a, b in zip(edges[:-1], edges[1:]):
    ov = np.clip(np.minimum(st[:, 3], b) - np.maximum(st[:, 0], a), 0, None).sum()
    busy.append(ov / (256 * (b - a)))
print("fraction of 256 CUs inside a task, by tenths:", " ".join(f"{v:.2f}" for v in busy))
for k in (0, 1, 2):
    m = tasks[:, 0] == k
    if m.any():
        sh = []
        for a, b in zip(edges[:-1], edges[1:]):
            ov = np.clip(np.minimum(st[m, 3], b) - np.maximum(st[m, 0], a), 0, None).sum()
            sh.append(ov / (256 * (b - a)))
        print(f"   {names[k]:4s}:", " ".join(f"{v:.2f}" for v in sh))
# INV detail
m = (tasks[:, 0] == 1) & (tasks[:, 2] > tasks[:, 1])
if m.any():
    s = st[m]; I = tasks[m, 1]; J = tasks[m, 2]
    w = s[:, 2] - s[:, 1]; sv = s[:, 3] - s[:, 2]
    q = lambda a: " ".join(f"{v:.1f}" for v in np.percentile(a, [10, 50, 90, 99]))
    print("INV off-diagonal: wait-for-L(c,c) p10/50/90/99:", q(w), "| solve+publish:", q(sv))
    for lo, hi in ((0, 20), (20, 40), (40, 60), (60, 80)):
        mm = (J >= lo) & (J < hi)
        if mm.any():
            print(f"   columns {lo}..{hi}: wait {q(w[mm])} | solve+publish {q(sv[mm])} | contraction per k-block {((s[mm,1]-s[mm,0]).sum()/np.maximum(1,(J-I)[mm]).sum()):.2f}")
m = (tasks[:, 0] == 0) & (tasks[:, 1] > tasks[:, 2])
if m.any():
    s = st[m]
    print("CHOL off-diagonal: wait p10/50/90/99:", q(s[:, 2] - s[:, 1]), "| solve+publish:", q(s[:, 3] - s[:, 2]))
