"""Dev: eager vs graphed NeighborLoader GGD regime (bench.py sampled_gcl_figures alone)."""
import json, sys
sys.path.insert(0, ".")
import torch
import bench
print(json.dumps(bench.sampled_gcl_figures(torch.device("cuda", 0)), indent=1))
