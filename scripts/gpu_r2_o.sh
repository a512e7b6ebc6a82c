#!/bin/bash
# round 2, call O: the N > 1 code path of bench.py in a world of one rank, launched exactly as the driver
# launches N > 1 (torch.distributed.run): process group + RCCL communicator behind the C ABI in one
# process, mi355_search_sharded with torch tensors, stats, teardown, exit code
O=gpurun_out/r2o
mkdir -p $O
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 1 --steps 5 --warmup 1 --force-sharded-path --recall-rows 0 --secondary 0 --cpu-seconds 0 > $O/bench_sharded1.json 2> $O/bench_sharded1.err
echo "torchrun bench rc=$?"; grep -v "amdgpu.ids" $O/bench_sharded1.err | tail -5; cat $O/bench_sharded1.json | head -c 3000; echo
