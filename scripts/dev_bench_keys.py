import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["ms_per_step_median"], d["ms_per_step_fresh_structures"], d["sampled_regime"]["graphed_steps_per_s"], d["ms_gcl_step"], d["ms_kge_step"])
