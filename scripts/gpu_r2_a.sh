#!/bin/bash
# round 2, GPU call A: the whole GPU suite, then flat GEMM A/B (4 M rows, then full C2), then C3.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r2a/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 > gpurun_out/r2a/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
for v in 2 4 5; do
  for g in 0; do
    timeout 200 python bench.py --workload flat --flat-rows 4000000 --steps 8 --warmup 2 --cpu-seconds 0 --flat-gemm $v --flat-grid $g > gpurun_out/r2a/flat4m_v${v}_g${g}.json 2> gpurun_out/r2a/flat4m_v${v}_g${g}.err
    echo "flat4m variant $v grid $g: $(python -c "import json,sys; d=json.load(open('gpurun_out/r2a/flat4m_v${v}_g${g}.json')); print(round(d['ms_per_step'],3),'ms', round(d['roofline']['achieved'],1),'TF')" 2>&1 | tail -1)"
  done
done
for v in 2 4; do
  timeout 300 python bench.py --workload flat --steps 8 --warmup 2 --cpu-seconds 12 --flat-gemm $v > gpurun_out/r2a/flat_c2_v${v}.json 2> gpurun_out/r2a/flat_c2_v${v}.err
  echo "flat C2 variant $v: $(python -c "import json,sys; d=json.load(open('gpurun_out/r2a/flat_c2_v${v}.json')); print(round(d['ms_per_step'],3),'ms', round(d['roofline']['achieved'],1),'TF', d.get('cpu_baseline',{}).get('parity'))" 2>&1 | tail -1)"
done
timeout 400 python bench.py --steps 10 --warmup 2 --recall-rows 0 --cpu-seconds 8 > gpurun_out/r2a/c3.json 2> gpurun_out/r2a/c3.err
echo "C3: $(python -c "import json; d=json.load(open('gpurun_out/r2a/c3.json')); print(round(d['value']),'QPS', d['roofline']['frac'], d.get('cpu_baseline',{}).get('parity'))" 2>&1 | tail -1)"
