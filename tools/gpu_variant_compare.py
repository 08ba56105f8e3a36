"""Per-launch comparison of GEMM tile shapes on the real launch list of factorize / predict /
gradient (serialised: no look-ahead, no parallel inverse), argv: N.  Prints, per distinct launch
shape, the time under forced 128x128, forced 64x64, forced 128x64 and the default selection."""
import os, subprocess, sys
from collections import defaultdict

N = sys.argv[1] if len(sys.argv) > 1 else "10000"
settings = {"v0": {"GMB_GEMM_VARIANT": "0"}, "v1": {"GMB_GEMM_VARIANT": "1"}, "v2": {"GMB_GEMM_VARIANT": "2"}, "model": {}}
res = {}
for name, env in settings.items():
    path = f"/tmp/trace_{name}.txt"
    if os.path.exists(path):
        os.remove(path)
    e = dict(os.environ, GMB_TRACE_FILE=path, GMB_LOOKAHEAD="0", GMB_PAR_INVERSE="0", **env)
    subprocess.run([sys.executable, "tools/gpu_trace_run.py", N], env=e, capture_output=True, text=True)
    res[name] = [ln.split() for ln in open(path) if ln.strip()]
names = {0: "chol", 1: "leaf", 2: "trsm", 3: "pred", 4: "grad", 5: "strip", 6: "pstrip"}
n = min(len(v) for v in res.values())
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
chosen = {}
for i in range(n):
    kind, mt, nt, k, flags = (int(v) for v in res["v0"][i][:5])
    if kind in (1, 5, 6):
        continue
    key = (names[kind], mt, nt, k, flags & 0xff)
    cnt[key] += 1
    for name in settings:
        agg[key][name] += float(res[name][i][5])
    chosen[key] = int(res["model"][i][4]) >> 8
tot = {name: sum(a[name] for a in agg.values()) for name in settings}
print("totals (ms):", {k: round(v, 2) for k, v in tot.items()}, " best-of-per-shape:", round(sum(min(a[v] for v in ("v0", "v1", "v2")) for a in agg.values()), 2))
print(f"{'kind':5s} {'mt':>4s} {'nt':>4s} {'k':>6s} {'fl':>3s} {'n':>4s} {'v0':>8s} {'v1':>8s} {'v2':>8s} {'model':>8s} pick")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1]["v0"])[:60]:
    print(f"{key[0]:5s} {key[1]:4d} {key[2]:4d} {key[3]:6d} {key[4]:3d} {cnt[key]:4d} {a['v0']:8.3f} {a['v1']:8.3f} {a['v2']:8.3f} {a['model']:8.3f} v{chosen[key]}")
