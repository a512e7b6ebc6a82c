#!/bin/bash
# round 2, call K: HBM traffic (FETCH_SIZE, own PMC pass) of the flat GEMM that is now the default
O=$GRAFT_REPO_ROOT/gpurun_out/r2k
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- \
  python $GRAFT_REPO_ROOT/bench.py --workload flat --steps 2 --warmup 1 --cpu-seconds 0 > $O/fetch.log 2>&1
echo "fetch rc=$?"; tail -2 $O/fetch.log | cut -c1-400
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/r2k/fetch/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:6]:
        print(k, "launches", len(v), "FETCH_SIZE per launch", sum(v) / len(v))
PY
find gpurun_out/r2k -name "*kernel_trace.csv" -size +20M -delete
