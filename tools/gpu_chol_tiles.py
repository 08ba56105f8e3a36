"""Persistent tile Cholesky (csrc/chol_tiles.hpp) against the stream schedules: factor / v / log-det agreement and
chol_ms per schedule; CT_SIZES=10000x4,... picks the cases, CT_TRACE=1 prints the per-column chain of the traced run."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O

sizes = [tuple(int(v) for v in t.split('x')) for t in os.environ.get('CT_SIZES', '300x2,1000x3,2500x4,4096x4,10000x4').split(',')]
schemes = [int(s) for s in os.environ.get('CT_SCHEMES', '0,2,3').split(',')]
reps = int(os.environ.get('CT_REPS', '5'))
for N, d in sizes:
    X, y, ls = O.synthetic_table(N, d)
    e = engine.Engine(0)
    e.set_data(X, y)
    e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
    e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
    ref = None
    line = []
    for scheme in schemes:
        e.set_chol_scheme(scheme)
        try:
            e.factorize()
        except Exception as ex:  # noqa: BLE001
            print(f"N={N} scheme {scheme}: FAILED {ex}")
            continue
        nr = min(N, 2048)
        rows = e.copy_factor(r0=N - nr, nr=nr, c0=0, nc=N)
        v = e.copy_v()
        nl = e.nlml()
        best = 1e9
        for _ in range(reps):
            e.factorize()
            best = min(best, e.timings()['chol_ms'])
        if os.environ.get('CT_PREDICT') == '1':  # the triangular solve of the predict path follows the same switch
            Xs = np.random.default_rng(5).standard_normal((int(os.environ.get('CT_M', '10000')), d))
            mu, var = e.predict(Xs)
            tp = 1e9
            for _ in range(reps):
                e.predict(Xs)
                tp = min(tp, e.timings()['predict_ms'])
            if scheme == schemes[0]:
                pref = (mu, var)
            pd = f" predict {tp:.2f} ms ({N * N * len(Xs) / tp / 1e9:.1f} TF/s) dmu {np.max(np.abs(mu - pref[0])) / np.max(np.abs(pref[0])):.1e} dvar {np.max(np.abs(var - pref[1])):.1e}"
        else:
            pd = ""
        if ref is None:
            ref = (rows, v, nl)
            line.append(f"scheme {scheme}: {best:.3f} ms ({N**3/3/best/1e9:.1f} TF/s) nlml {nl:.12g}" + pd)
        else:
            dl = np.max(np.abs(np.tril(rows, N - nr) - np.tril(ref[0], N - nr))) / np.max(np.abs(ref[0]))
            dv = np.max(np.abs(v - ref[1])) / np.max(np.abs(ref[1]))
            line.append(f"scheme {scheme}: {best:.3f} ms ({N**3/3/best/1e9:.1f} TF/s) dL {dl:.1e} dv {dv:.1e} dnlml {abs(nl-ref[2])/abs(ref[2]):.1e}" + pd)
    print(f"N={N} d={d}: " + " | ".join(line), flush=True)
    if os.environ.get('CT_TRACE') == '1' and 3 in schemes:
        e.set_chol_scheme(3)
        e.chol_task_trace(1)
        e.factorize()
        tiles, st = e.chol_task_trace(0)
        st = st * 1e6  # us
        if os.environ.get('CT_TRACE_SAVE'):  # raw stamps for offline analysis (tools/ct_schedule_sim.py)
            np.savez(os.environ['CT_TRACE_SAVE'] + f"_{N}.npz", tiles=np.asarray(tiles), stamps_us=st)
        nct = (N + 127) // 128
        diag = {int(j): st[i] for i, (ii, j) in enumerate(tiles) if ii == j}
        sub = {int(j): st[i] for i, (ii, j) in enumerate(tiles) if ii == j + 1}
        print(f"  trace: last publish {st[:, 3].max():.1f} us; tasks {len(tiles)}")
        print("  col: diag[taken ksum-done leaf-in published]  sub[taken ksum-done diag-ready published]  chain step")
        prev = 0.0
        for j in range(nct):
            dj = diag[j]
            sj = sub.get(j)
            step = dj[3] - prev
            prev = dj[3]
            if j < 12 or j % 8 == 0 or j >= nct - 6:
                print(f"  {j:3d}: diag {dj[0]:8.1f} {dj[1]:8.1f} {dj[2]:8.1f} {dj[3]:8.1f}  sub " +
                      (f"{sj[0]:8.1f} {sj[1]:8.1f} {sj[2]:8.1f} {sj[3]:8.1f}" if sj is not None else "-") + f"  step {step:6.1f}")
        # contraction time per k-block of the bulk tiles (I >= J + 2), by column: minimum / median / maximum (waits included)
        per = {}
        for i, (ii, j) in enumerate(tiles):
            if j > 0 and ii >= j + 2:
                per.setdefault(int(j), []).append((st[i, 1] - st[i, 0]) / j)
        print("  us per k-block of bulk tiles (min / median / max), by column:")
        print("   " + "  ".join(f"{j}: {min(v):.1f}/{np.median(v):.1f}/{max(v):.1f}" for j, v in sorted(per.items()) if j % 6 == 0 or j < 4))
        busy = (st[:, 3] - st[:, 0]).sum()
        ks = (st[:, 1] - st[:, 0]).sum()
        print(f"  sum of task spans {busy/1e3:.2f} ms, of contraction spans {ks/1e3:.2f} ms over {min(len(tiles), 512)} workgroups")
    e.close()
