"""Summarise rocprofv3 --pmc SQ passes per kernel: mean counter value per dispatch (gpurun_out/msf_<tag>/p*/...)."""
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
res = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][:60]
        res[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items() if "ms_fused" in k or "gemm" in k}
json.dump(summary, open(os.path.join(out, "sq.json"), "w"), indent=1)
for k, d in summary.items():
    print(k)
    wc = d.get("SQ_WAVE_CYCLES", 0)
    for c, v in sorted(d.items()):
        print("   %-28s %14.0f %s" % (c, v, ("%.3f of WAVE_CYCLES" % (v / wc)) if wc and c.startswith("SQ_") else ""))
