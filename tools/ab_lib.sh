#!/bin/bash
L=vit-deep-radiomics_amd/vdr/libvdr.so
for r in 1 2; do
  for w in base exp; do
    cp tools/micro/libvdr_$w.so.bin $L
    echo -n "$w: "
    timeout -k 10 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items() if k.startswith('gemm_')})"
  done
done
cp tools/micro/libvdr_base.so.bin $L
