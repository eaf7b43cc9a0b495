set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof gpurun_out/pmc1 gpurun_out/pmc2
ARGS="--grid ${GRID:-216} --steps 10 --warmup 3 --cpu-iters 0 --no-solve --coarse ${COARSE:-pou}"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o run -- python3 bench.py $ARGS > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.log || { tail -20 gpurun_out/prof/bench.log; exit 1; }
head -12 gpurun_out/prof/run_kernel_stats.csv | cut -c1-160
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc1 -o run -- python3 bench.py $ARGS > gpurun_out/pmc1/bench.json 2> gpurun_out/pmc1/bench.log || { tail -20 gpurun_out/pmc1/bench.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc2 -o run -- python3 bench.py $ARGS > gpurun_out/pmc2/bench.json 2> gpurun_out/pmc2/bench.log || { tail -20 gpurun_out/pmc2/bench.log; exit 1; }
ls gpurun_out/pmc1 gpurun_out/pmc2
python3 - <<'PY'
import csv, collections
for tag, d in (("FETCH_SIZE","gpurun_out/pmc1"),("WRITE_SIZE","gpurun_out/pmc2")):
    import glob
    f = glob.glob(d+"/*counter_collection.csv")
    if not f: print("no counter file in", d); continue
    acc = collections.defaultdict(lambda: [0.0,0])
    for row in csv.DictReader(open(f[0])):
        if row.get("Counter_Name") != tag: continue
        k = row["Kernel_Name"].split("(")[0][:60]
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    print(tag)
    for k,(v,n) in sorted(acc.items(), key=lambda kv:-kv[1][0])[:8]:
        print(f"  {k:60s} total {v:14.1f}  per-dispatch {v/n:12.1f}  n={n}")
PY
