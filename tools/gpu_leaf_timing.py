import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["GMB_LEAF_DBG"] = "1"
import numpy as np, torch
from gumbi_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(0)
G = rng.standard_normal((128, 300)); S = G @ G.T / 300 + 0.5 * np.eye(128)
dev = torch.device("cuda:0")
for rep in range(3):
    tA = torch.tensor(S.T.copy(), device=dev); tI = torch.zeros(8 * 256, dtype=torch.float64, device=dev)
    tL = torch.zeros(64, dtype=torch.float64, device=dev); tinfo = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    eng.blk_potrf(tA.data_ptr(), 128, 128, tI.data_ptr(), tL.data_ptr(), tinfo.data_ptr())
    torch.cuda.synchronize()
    st = tL.cpu().numpy()[1:7]
    full = tL.cpu().numpy()
    st = full[1:7]
    st = full[1:6]
    print("stamps (us): load %.2f  factor %.2f  writeback %.2f  dinv16-out %.2f  total %.2f" % tuple(list(np.diff(st) / 100.0) + [(st[-1] - st[0]) / 100.0]))

st = full[1:]
t1 = st[1]
for s_ in range(7):
    b_done, b_bar, c_done = st[8 + 3 * s_], st[9 + 3 * s_], st[10 + 3 * s_]
    print(f"  step {s_}: phaseB(wave0) {(b_done - t1)/100:.2f} us  barrier-wait {(b_bar - b_done)/100:.2f}  phaseC(wave0: tile+factor16) {(c_done - b_bar)/100:.2f}")
    t1 = c_done

d16 = full[41:52]
print("diag16 (cycles): load %d | factor steps %s | inverse steps %s | store %d | total %d" % (
    d16[1] - d16[0], [int(v) for v in np.diff(d16[1:6])], [int(v) for v in np.diff(d16[5:10])], d16[10] - d16[9], d16[10] - d16[0]))
