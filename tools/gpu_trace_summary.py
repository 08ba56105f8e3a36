"""Summarise a GMB_TRACE_FILE per-launch log: efficiency of gemm launches by shape."""
import sys
from collections import defaultdict

rows = [ln.split() for ln in open(sys.argv[1]) if ln.strip()]
agg = defaultdict(lambda: [0, 0.0, 0.0])
names = {0: "panel", 1: "leaf", 2: "trsm", 3: "pred", 4: "grad", 5: "strip", 6: "strip", 7: "bulk"}
for kind, mt, nt, k, flags, ms, gf in rows:
    key = (names[int(kind)], int(mt), int(nt), int(k), int(flags))
    a = agg[key]
    a[0] += 1
    a[1] += float(ms)
    a[2] += float(gf)
tot = sum(a[1] for a in agg.values())
print(f"total {tot:.2f} ms")
print(f"{'kind':5s} {'mt':>4s} {'nt':>4s} {'k':>6s} {'fl':>3s} {'n':>5s} {'ms_tot':>9s} {'ms_avg':>8s} {'TF/s':>7s} {'%':>5s}")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    tf = a[2] / a[1] if a[1] > 0 else 0.0
    print(f"{key[0]:5s} {key[1]:4d} {key[2]:4d} {key[3]:6d} {key[4]:3d} {a[0]:5d} {a[1]:9.3f} {a[1]/a[0]:8.4f} {tf:7.2f} {100*a[1]/tot:5.1f}")
