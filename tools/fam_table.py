import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print("ms/step %.2f"%d["ms_per_step"])
for k,v in d["kernels"].items():
    print("%-52s %5.1f/step %8.1f us %7.3f ms/step %9.1f %s frac %.3f"%(k,v["launches_per_step"],v["avg_us"],v["ms_per_step"],v["achieved"],v["unit"],v["frac"]))
