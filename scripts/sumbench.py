import json,sys
d=json.load(open(sys.argv[1])); print(round(d["value"]), round(d["ms_per_step"],1), d["config"]["eig"]["outer_iterations"], d["config"]["eig"]["g_products"], {k:round(v,1) for k,v in d["stage_ms_per_step"].items()}); print({k:(round(v["ms_per_step"],2), round(v["launches_per_step"])) for k,v in d["kernels"].items()})
