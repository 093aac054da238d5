import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"])
for k,v in sorted(d["gpu_time"]["stages_ms_per_step"].items()):
    if 'wino4' in k: print(k,v)
