#!/bin/bash
# per-class GEMM times of the MedSAM encoder under different tile variants (B = 1 and 4)
for b in 1 4; do
  for v in 0 16 19 20 14; do
    echo "== batch $b variant $v"
    VDR_GEMM_VARIANT=$v timeout -k 10 200 python bench.py --model medsam --batch $b --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items() if k.startswith('gemm') or k=='attention'})"
  done
done
