#!/bin/bash
# On an MI355X box (gpurun -- 'bash scripts/gpu_full_run.sh [outdir]'): the whole GPU suite (torch's bundled HIP runtime, as
# the driver runs it, and once more torch-free on the system runtime), the default bench line end to end (wall time
# printed), the N > 1 code path of bench.py in a world of one rank (C3 shape and the C4 mode: sharded coarse), smoke().
# The summaries a round keeps are copied from the out directory into profiles/ by hand.
O=${1:-gpurun_out/full_run}
mkdir -p $O
export PYTHONFAULTHANDLER=1
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.txt 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - S ))s"; tail -3 $O/pytest.txt
if [ -z "$SKIP_SYSTEM_RUNTIME_PASS" ]; then
S=$(date +%s)
MI355_HIP_RUNTIME=system timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_system_runtime.txt 2>&1
echo "pytest (MI355_HIP_RUNTIME=system, torch-free) rc=$? wall=$(( $(date +%s) - S ))s"; tail -3 $O/pytest_system_runtime.txt
fi
S=$(date +%s)
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$? wall=$(( $(date +%s) - S ))s"; tail -3 $O/bench_default.err | cut -c1-300
python -c "
import json
d = json.load(open('$O/bench_default.json'))
print(json.dumps(d['summary']))"
for mode in "" "--shard-coarse --batch-per-gpu 1024"; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 \
    --force-sharded-path $mode --steps 6 --warmup 2 > $O/bench_sharded_world1$(echo $mode | tr -d ' -').json 2> $O/bench_sharded.err
  echo "sharded world-1 [$mode] rc=$?"; tail -2 $O/bench_sharded.err | cut -c1-300
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 1 \
  --force-sharded-path --workload c4 --c4-rows 200000000 --steps 6 --warmup 2 > $O/bench_c4_sharded_world1.json 2> $O/bench_c4_sharded.err
echo "c4 sharded world-1 rc=$?"; tail -2 $O/bench_c4_sharded.err | cut -c1-300
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*sharded*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], round(d["value"]), d["config"]["workload"], d["multi_gpu"]["coarse"], d["multi_gpu"].get("sharded_equals_unsharded"), d["multi_gpu"]["all_ranks_returned_the_same_results"])
    except Exception as e:
        print(f, "unreadable", e)
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
