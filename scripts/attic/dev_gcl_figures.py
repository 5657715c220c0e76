import json, sys, torch
sys.path.insert(0, ".")
import bench
print(json.dumps(bench.sampled_gcl_figures(torch.device("cuda", 0)), indent=1))
