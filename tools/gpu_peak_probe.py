import os, subprocess, sys
for per_cu in (1, 2, 4):
    for scale in ("1.0", "0.0"):
        env = dict(os.environ, GMB_PEAK_BLOCKS_PER_CU=str(per_cu), GMB_PEAK_SCALE=scale)
        out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0,'.'); from gumbi_amd import engine; print([round(engine.mfma_f64_peak(0)[0],1) for _ in range(3)])"], env=env, capture_output=True, text=True)
        print(f"blocks/CU={per_cu} scale={scale}:", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
