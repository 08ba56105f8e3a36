"""Does independent bulk MFMA work hide under the chain-bound Cholesky?  One engine factorises
N = 10k while a second engine (its own stream) runs a plain GEMM of the size of the triangular
inverse's root product (5000^3): alone, alone, then together -- if together ~ max(alone) the
gradient's first root product could be started during the second half of the factorisation; if
together ~ sum it cannot.  (argv: N gemm_edge)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from gumbi_amd import engine
from oracle import gp_oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 5120
d = 4
X, y, ls = O.synthetic_table(N, d)
e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)))); e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
g = engine.Engine(0)
dev = torch.device("cuda:0")
A = torch.randn(E, E, dtype=torch.float64, device=dev); B = torch.randn(E, E, dtype=torch.float64, device=dev)
Cm = torch.zeros(E, E, dtype=torch.float64, device=dev)
def gemm():
    g.blk_gemm_nt(Cm.data_ptr(), E, A.data_ptr(), E, B.data_ptr(), E, E, E, E, 1.0, 0.0)
def sync():
    torch.cuda.synchronize()
e.factorize(); gemm(); sync()
for rep in range(3):
    sync(); t0 = time.perf_counter(); e.factorize(); sync(); tf = time.perf_counter() - t0
    sync(); t0 = time.perf_counter(); gemm(); sync(); tg = time.perf_counter() - t0
    sync(); t0 = time.perf_counter(); gemm(); e.factorize(); sync(); tb = time.perf_counter() - t0
    print(f"factorize alone {tf*1e3:.2f} ms   gemm {E}^3 alone {tg*1e3:.2f} ms ({2*E**3/tg/1e12:.1f} TF/s)   together {tb*1e3:.2f} ms   (sum {1e3*(tf+tg):.2f})")
