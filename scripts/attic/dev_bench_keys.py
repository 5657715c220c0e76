"""Dev: the step time and the tagged calls' launch times of a bench line (file argument)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
tags = {k["tag"]: round(k["avg_launch_ms"], 4) for k in [d["roofline"]] + d["kernels"]}
print(round(d["ms_per_step"], 3), round(d["ms_gcl_step"], 3), round(d["ms_kge_step"], 3), tags)
