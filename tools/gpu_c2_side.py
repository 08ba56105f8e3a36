import sys, json
sys.path.insert(0, '.')
import torch, bench
clock = bench.Clock(None, torch.device("cuda", 0))
print(json.dumps(bench.c2_side_section(0, clock))[:1500])
