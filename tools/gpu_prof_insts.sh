#!/bin/bash
# instruction mix of the dominant kernel (rocprofv3 PMC pass, kernel-trace only): summary only travels back
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc3 /tmp/pmc3
ARGS="--grid 216 --steps 10 --warmup 3 --cpu-iters 0 --no-solve --coarse pou"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d /tmp/pmc3 -o run -- python3 bench.py $ARGS > gpurun_out/pmc3/bench.json 2> gpurun_out/pmc3/bench.log || { tail -20 gpurun_out/pmc3/bench.log; exit 1; }
python3 - <<'PY'
import csv, collections, glob, json
f = glob.glob("/tmp/pmc3/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
    a = acc[k][row["Counter_Name"]]
    a[0] += float(row["Counter_Value"]); a[1] += 1
out = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in acc.items() if k.startswith("ddm::k_trsv_pipe") or k.startswith("ddm::k_spmv_stream")}
json.dump(out, open("gpurun_out/pmc3/insts_per_dispatch.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
