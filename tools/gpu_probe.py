"""One-shot GPU diagnostics (run on the MI355X box through gpurun): runtime mapping, MFMA f64
peak, and per-phase timings of factorize / gradient / predict at a few sizes."""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from gumbi_amd import engine  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402


def libs():
    return sorted({ln.split()[-1] for ln in open(f"/proc/{os.getpid()}/maps") if "amdhip" in ln or "hsa-runtime" in ln})


def main():
    sizes = [int(s) for s in (sys.argv[1:] or ["2000", "10000"])]
    print("devices:", engine.device_count(), "runtime:", libs())
    print("mfma f64 peak (TFLOP/s, cycles/MFMA/wave):", [tuple(round(v, 2) for v in engine.mfma_f64_peak(0)) for _ in range(3)])
    for N in sizes:
        d = 4
        X, y, ls = O.synthetic_table(N, d)
        Xs = O.synthetic_grid(d)
        spec = engine.KernelSpec(D=d, idx_cont=list(range(d)), kind="ExpQuad")
        theta = np.concatenate([ls, [1.0, 0.2]])
        eng = engine.Engine(0)
        eng.set_data(X, y)
        eng.set_kernel(spec)
        eng.set_theta(theta)
        for prof in (False, True):
            eng.set_profiling(prof)
            for _ in range(2):
                t0 = time.perf_counter()
                eng.factorize()
                t1 = time.perf_counter()
                mu, var = eng.predict(Xs)
                t2 = time.perf_counter()
                eng.factorize()
                t3 = time.perf_counter()
                val, g = eng.nlml(grad=True)
                t4 = time.perf_counter()
                eng.factorize()
            tm = eng.timings()
            print(f"N={N} prof={prof}: factorize {1e3*(t1-t0):.2f} ms (kbuild {tm['kbuild_ms']:.3f}, chol {tm['chol_ms']:.3f}) "
                  f"predict {1e3*(t2-t1):.2f} ms (dev {tm['predict_ms']:.3f}) grad {1e3*(t4-t3):.2f} ms (dev {tm['grad_ms']:.3f})")
            if prof:
                cg = tm["chol_gemm_flops"] / max(tm["chol_gemm_ms"], 1e-9) / 1e9
                pg = tm["predict_gemm_flops"] / max(tm["predict_gemm_ms"], 1e-9) / 1e9
                gg = tm["grad_gemm_flops"] / max(tm["grad_gemm_ms"], 1e-9) / 1e9
                print(f"   chol: gemm {tm['chol_gemm_ms']:.3f} ms / {tm['chol_gemm_launches']} launches = {cg:.1f} TF/s; "
                      f"leaf {tm['chol_leaf_ms']:.3f} ms; trsm {tm['chol_trsm_ms']:.3f} ms")
                print(f"   predict gemm {tm['predict_gemm_ms']:.3f} ms = {pg:.1f} TF/s;  grad gemm {tm['grad_gemm_ms']:.3f} ms = {gg:.1f} TF/s")
                print(f"   kbuild {tm['kbuild_bytes']/max(tm['kbuild_ms'],1e-9)/1e6:.1f} GB/s algorithmic")
        eng.close()


if __name__ == "__main__":
    main()
