"""cProfile of the user-level sequence at C1 size (DataSet -> GP.fit -> prepare_grid -> predict_grid, N = 392, d = 1):
where the ~17 ms go once the engine is warm.   python tools/gpu_c1_profile.py"""
import cProfile, pstats, sys, time
sys.path.insert(0, '.')
import bench
cfg = bench.CONFIGS["c1"]
bench.end_to_end_fit(cfg, 0, 0)  # warm: library load, HIP context
ts = []
for _ in range(5):
    t0 = time.perf_counter(); r = bench.end_to_end_fit(cfg, 0, 0); ts.append(time.perf_counter() - t0)
print("end to end, 5 runs (ms):", [round(1e3 * t, 1) for t in ts], r)
pr = cProfile.Profile(); pr.enable(); bench.end_to_end_fit(cfg, 0, 0); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
