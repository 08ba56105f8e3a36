"""Feasibility probe of a software XCD partition: does the latency-bound leaf
factorisation keep its stand-alone speed when an MFMA-saturating kernel occupies the REST of the chip?

Three cases, each timing 200 back-to-back 128 x 128 leaf factorisations on the engine's stream while
a background thread runs the MFMA peak microbenchmark (null stream) for ~tens of ms:
  alone        no background load
  shared       background load on all eight XCDs (what a concurrent trailing update does today)
  partitioned  background workgroups leave XCD 7 at once (GMB_PEAK_SKIP_XCD=7); the leaf is launched
               with 8 workgroups of which only the one on XCD 7 works (GMB_PROBE_XCD=7)
Each case runs in its own process (the switches are read once).  argv: none."""
import os, subprocess, sys

CODE = r'''
import sys, time, threading; sys.path.insert(0, '.')
import numpy as np, torch
from gumbi_amd import engine
eng = engine.Engine(0)
rng = np.random.default_rng(0)
G = rng.standard_normal((128, 300)); S = G @ G.T / 300 + 0.5 * np.eye(128)
dev = torch.device("cuda:0")
tA0 = torch.tensor(S.T.copy(), device=dev)
tAs = [tA0.clone() for _ in range(200)]
tI = torch.zeros(8 * 256, dtype=torch.float64, device=dev); tL = torch.zeros(64, dtype=torch.float64, device=dev)
tinfo = torch.zeros(1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
_s = eng.stream
es = torch.cuda.ExternalStream(_s() if callable(_s) else _s, device=dev)
def leaves():
    t0 = time.perf_counter()
    for t in tAs:
        eng.blk_potrf(t.data_ptr(), 128, 128, tI.data_ptr(), 0, tinfo.data_ptr())
    es.synchronize()   # the engine's stream only: a device-wide sync would wait for the background kernel
    return (time.perf_counter() - t0) / len(tAs) * 1e6
for t in tAs[:5]:
    eng.blk_potrf(t.data_ptr(), 128, 128, tI.data_ptr(), 0, tinfo.data_ptr())
torch.cuda.synchronize()
bg = sys.argv[1] == "1"
res = {}
if bg:
    th = threading.Thread(target=lambda: res.setdefault("tf", engine.mfma_f64_peak(0)))
    th.start()
    time.sleep(0.02)   # let the background kernel get resident
us = leaves()
if bg:
    th.join()
L = torch.tril(tAs[-1].T).cpu().numpy()   # column-major block -> (row, col)
err = np.max(np.abs(L @ L.T - S)) / np.max(np.abs(S))
print("leaf %.1f us each   background %s   L L^T error %.1e" % (us, ("%.1f TF/s" % res["tf"][0]) if bg else "-", err))
'''
cases = [("alone", "0", {}),
         ("shared", "1", {"GMB_PEAK_ITERS": "400000"}),
         ("partitioned", "1", {"GMB_PEAK_ITERS": "400000", "GMB_PEAK_SKIP_XCD": "7", "GMB_PROBE_XCD": "7"}),
         ("leaf on XCD 7 only, no load", "0", {"GMB_PROBE_XCD": "7"})]
# the engine owns four streams, torch one more and the background load runs on the null stream: with the
# runtime's default of four hardware queues two of them would share a queue and serialise
hwq = os.environ.get("PROBE_HW_QUEUES", "8")
for name, bg, env in cases:
    out = subprocess.run([sys.executable, "-c", CODE, bg], env=dict(os.environ, GPU_MAX_HW_QUEUES=hwq, **env),
                         capture_output=True, text=True)
    print("%-28s %s" % (name, out.stdout.strip().splitlines()[-1] if out.returncode == 0 and out.stdout.strip() else out.stderr[-500:]))
