#!/bin/bash
# usage: tools/sweep.sh "<variant list>" "<streams list>" "<micro-batch list>"
for v in $1; do for st in $2; do for mb in $3; do
  r=$(VDR_GEMM_VARIANT=$v timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams $st --micro-batch $mb 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_time_sum_ms_per_step'])")
  echo "variant $v streams $st mb $mb : img/s ms/step ksum = $r"
done; done; done
