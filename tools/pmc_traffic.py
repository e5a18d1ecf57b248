"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/prof_pmc.sh) -> HBM bytes per launch per GEMM family.

Counter units are KiB; FETCH_SIZE is doubled (gfx950 reports half of wide streaming reads, see the HBM section of
/opt/skills/guides/MI355X_MICROARCH.md).  Family = layout + N tile, the same naming as the profiler spans of
prifit_amd/nn_ops.gemm ("gemm_nn_bn128"): the template arguments of gemm_kernel<BM, BN, WM, WN, LAY, ...>."""
import csv, glob, json, os, re, sys

def load(d, counter):
    f = glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        raise SystemExit("no counter_collection.csv under %s/%s" % (d, counter))
    per = {}
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        m = re.match(r"void gemm_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)", name)
        ms = re.match(r"void gemm_stream_kernel<(\d+), (\d+), (true|false)", name)
        if m:
            fam = "gemm_%s_bn%s" % (("nt", "nn", "tn")[int(m.group(5))], m.group(2))
        elif ms:       # profiler span names of prifit_amd/nn_ops.gemm: gemm_stream_nt / gemm_stream_nn
            fam = "gemm_stream_%s" % ("nt" if ms.group(3) == "true" else "nn")
        elif "gemm_stream_tn_kernel" in name:
            fam = "gemm_stream_tn"
        elif re.match(r"void gemm_pers_kernel<(\d+)", name):   # persistent form of the 128 x 128 kernel: same span family
            fam = "gemm_%s_bn128" % ("nt", "nn")[int(re.match(r"void gemm_pers_kernel<(\d+)", name).group(1))]
        elif "sa_group_kernel" in name:
            fam = "sa_group_linear"
        else:
            fam = name.split("(")[0].replace("void ", "")[:60]
        a = per.setdefault(fam, [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return per

d = sys.argv[1]
fetch, write = load(d, "FETCH_SIZE"), load(d, "WRITE_SIZE")
fam = {}
for k, (fs, n) in fetch.items():
    ws, nw = write.get(k, (0.0, n))
    fam[k] = {"hbm_bytes_per_launch": (2.0 * fs / n + ws / max(nw, 1)) * 1024.0, "dispatches": n,
              "fetch_kb_raw_avg": fs / n, "write_kb_avg": ws / max(nw, 1)}
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/prof_pmc.sh) over `python3 bench.py "
                 "--steps 2 --warmup 1 --no-cpu-baseline`; FETCH_SIZE doubled per MI355X_MICROARCH.md",
       "families": {k: v for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])}}
json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
for k, v in list(out["families"].items())[:12]:
    print("%-40s %4d launches  %8.1f MB/launch" % (k, v["dispatches"], v["hbm_bytes_per_launch"] / 1e6))
