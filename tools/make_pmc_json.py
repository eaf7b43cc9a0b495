"""Condenses the two rocprofv3 --pmc passes of gpu_prof.sh (FETCH_SIZE in gpurun_out/pmc1, WRITE_SIZE in gpurun_out/pmc2) into
the per-kernel HBM traffic file that bench.py reads for roofline.traffic.  gfx950: read bytes = 2 x FETCH_SIZE (KiB) for
coalesced streams (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
usage: python tools/make_pmc_json.py out.json engine-name kernel-name-prefix"""
import collections
import csv
import glob
import json
import sys

out, engine, kprefix = sys.argv[1:4]
acc = {}
for tag, d in (("FETCH_SIZE", "gpurun_out/pmc1"), ("WRITE_SIZE", "gpurun_out/pmc2")):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    a = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") != tag:
            continue
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        a[k][0] += float(row["Counter_Value"])
        a[k][1] += 1
    acc[tag] = a
kern = {}
for k, (v, n) in acc["FETCH_SIZE"].items():
    w = acc["WRITE_SIZE"].get(k, [0.0, 1])
    kern[k] = {"FETCH_SIZE_KiB_per_dispatch": v / n, "dispatches": n, "WRITE_SIZE_KiB_per_dispatch": w[0] / max(w[1], 1),
               "hbm_bytes_per_dispatch_corrected": (2.0 * v / n + w[0] / max(w[1], 1)) * 1024.0}
match = [k for k in kern if k.startswith(kprefix)]
doc = {"note": "rocprofv3 --pmc passes (separate runs, --kernel-trace only) of the bench command named in the calling script (tools/gpu_prof*.sh); "
               "FETCH_SIZE/WRITE_SIZE are KiB as reported; gfx950: read bytes = 2 x FETCH_SIZE.",
       "kernels": dict(sorted(kern.items(), key=lambda kv: -kv[1]["hbm_bytes_per_dispatch_corrected"])[:24]),
       "engine_kernels": {engine: match[0]} if match else {}}
json.dump(doc, open(out, "w"), indent=1)
print(out, doc["engine_kernels"], {k: round(kern[k]["hbm_bytes_per_dispatch_corrected"] / 1e9, 3) for k in list(doc["kernels"])[:6]})
