#!/bin/bash
# round 3, call A: the whole GPU suite (loopback worlds, device-side second pass, overlap), host facts for the
# C5 line, the sharded bench path in a world of one (overlapped / serial), PMC refresh of the shipped scan kernel
O=gpurun_out/r3a
mkdir -p $O
export PYTHONFAULTHANDLER=1
S=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - S ))s"; tail -15 $O/pytest.txt
free -g | head -3; nproc; grep -i hugepagesize /proc/meminfo; cat /sys/kernel/mm/transparent_hugepage/enabled
timeout 300 python tests/tools/host_map_probe.py > $O/host_map_probe.txt 2>&1; echo "probe rc=$?"; cat $O/host_map_probe.txt
for mode in "" "--no-overlap"; do
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 \
    bench.py --gpus 1 --force-sharded-path $mode --steps 20 --warmup 3 --recall-rows 0 --secondary 0 --cpu-seconds 0 > $O/bench_sharded$mode.json 2> $O/bench_sharded$mode.err
  echo "sharded $mode rc=$?"; tail -2 $O/bench_sharded$mode.err; head -c 3000 $O/bench_sharded$mode.json; echo
done
timeout 400 python bench.py --steps 20 --warmup 3 --recall-rows 0 --secondary 0 --cpu-seconds 0 > $O/bench_plain.json 2> $O/bench_plain.err
echo "plain rc=$?"; head -c 2500 $O/bench_plain.json; echo
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
pass() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $R/$O/pmc_$name -o p --output-format csv -- \
    python $R/bench.py --steps 2 --warmup 1 --recall-rows 0 --secondary 0 --cpu-seconds 0 > $R/$O/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
pass fetch FETCH_SIZE
pass lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
cd $R
python scripts/pmc_summary.py $O k_scan_skew | tee $O/pmc_summary.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*.csv" -size +30M -delete
