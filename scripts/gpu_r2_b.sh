#!/bin/bash
# round 2, GPU call B: whole GPU suite (no -x), flat GEMM A/B incl. the DMA-in-M variant
cd "$(dirname "$0")/.."
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $O/pytest.log
for v in 2 4 6; do
  timeout 200 python bench.py --workload flat --flat-rows 4000000 --steps 8 --warmup 2 --cpu-seconds 0 --flat-gemm $v > $O/flat4m_v${v}.json 2> $O/flat4m_v${v}.err
  echo "flat4m variant $v: $(python -c "import json,sys; d=json.load(open('$O/flat4m_v${v}.json')); print(round(d['ms_per_step'],3),'ms', round(d['roofline']['achieved'],1),'TF')" 2>&1 | tail -1)"
done
for v in 2 6; do
  for m in l2 cosine; do
  timeout 300 python bench.py --workload flat --steps 8 --warmup 2 --cpu-seconds 10 --flat-gemm $v --flat-metric $m > $O/flat_c2_v${v}_$m.json 2> $O/flat_c2_v${v}_$m.err
  echo "flat C2 variant $v $m: $(python -c "import json,sys; d=json.load(open('$O/flat_c2_v${v}_$m.json')); print(round(d['ms_per_step'],3),'ms', round(d['roofline']['achieved'],1),'TF', d.get('cpu_baseline',{}).get('parity'))" 2>&1 | tail -1)"
  done
done
