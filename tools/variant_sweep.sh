#!/bin/bash
# whole-forward bench under per-class GEMM tile variants (ring2 = 19/21/16, ring3 = 22/23/24)
run() {
  echo "== qkv=$1 proj=$2 fc1=$3 fc2=$4"
  VDR_GEMM_VARIANT_QKV=$1 VDR_GEMM_VARIANT_PROJ=$2 VDR_GEMM_VARIANT_FC1=$3 VDR_GEMM_VARIANT_FC2=$4 timeout -k 10 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items() if k.startswith('gemm_') and k!='gemm_patch'})"
}
run 23 22 22 22
run 23 24 22 24
run 23 24 24 22
run 23 22 22 22
