#!/bin/bash
# round 3, call R: probe-weighted shard plan: loopback tests, the loopback_world8 leg and the sharded world-of-one bench path
O=gpurun_out/r3r
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_gpu_loopback.py tests/test_gpu_sharded.py tests/test_abi.py tests/test_build_host.py -m "gpu or not gpu" -x -q 2>&1 | tail -3
timeout 900 python bench.py --recall-rows 0 --recall2-rows 0 --c5-rows 0 --cpu-seconds 0 > $O/bench_loopback.json 2> $O/bench_loopback.err
echo "bench rc=$?"; tail -2 $O/bench_loopback.err | cut -c1-300
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3r/bench_loopback.json"))
lb = d["secondary"]["loopback_world8"]
print("C3", round(d["value"]))
print(json.dumps(lb["step_model"])[:500])
print("rows scanned", lb["overlapped"]["rows_scanned_by_rank"], lb["overlapped"]["load_imbalance_max_over_mean"], lb["overlapped"]["every_rank_equals_unsharded"])
print([round(p["ms_per_step_wall"], 3) for p in lb["stage_us_per_step_by_rank_alone"]], lb["shard_plan"])
PY
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 \
    bench.py --gpus 1 --force-sharded-path --steps 10 --warmup 2 --recall-rows 0 --recall2-rows 0 --secondary 0 --cpu-seconds 0 > $O/bench_sharded.json 2> $O/bench_sharded.err
echo "sharded rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r3r/bench_sharded.json')); print(round(d['value']), d['multi_gpu']['sharded_equals_unsharded'], d['multi_gpu']['exchange_overlapped_with_next_scan'])"
