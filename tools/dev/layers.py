import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"])
tf=d["gpu_time"].get("stages_tflops_direct_equiv",{})
for k,v in sorted(d["gpu_time"]["stages_ms_per_step"].items()):
    if len(sys.argv)>2 or 'wino4' in k: print(k,v,tf.get(k,''))
