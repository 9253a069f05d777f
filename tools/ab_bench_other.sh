#!/bin/bash
# A/B of two library builds (ab/libvdr_a.so, ab/libvdr_b.so) on the secondary configurations, alternating processes:
#   bash tools/ab_bench_other.sh "<bench args>" ["<bench args>" ...]
mkdir -p gpurun_out/ab2
L=vit-deep-radiomics_amd/vdr/libvdr.so
i=0
for args in "$@"; do
  i=$((i+1))
  for r in 1 2 3; do for n in a b; do
    cp ab/libvdr_$n.so $L
    timeout -k 10 150 python bench.py --no-cpu-baseline $args > gpurun_out/ab2/c${i}_${n}_$r.json 2>/dev/null || exit 1
  done; done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab2/c*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in d["kernels"].items() if k.startswith("gemm_f") or k in ("gemm_qkv", "attention")})
PY
