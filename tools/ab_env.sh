#!/bin/bash
# Whole-forward A/B of tuning knobs: bench.py on the tuning build (tools/ab_base/libvdr_tuning.so = `make TUNING=1 OUT=...`)
# in alternating processes on one box.   bash tools/ab_env.sh "VDR_GEMM_8P=1" "VDR_GEMM_8P=3" ...   (3 rounds each)
mkdir -p gpurun_out/abenv
cp tools/ab_base/libvdr_tuning.so vit-deep-radiomics_amd/vdr/libvdr.so
for r in 1 2 3; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e timeout -k 10 120 python bench.py --no-cpu-baseline ${BENCH_ARGS:---steps 40 --warmup 10} > gpurun_out/abenv/c${i}_$r.json 2> gpurun_out/abenv/err.log || { tail -5 gpurun_out/abenv/err.log; exit 1; }
  done
done
python - "$@" <<'PY'
import json, glob, sys
for i, e in enumerate(sys.argv[1:], 1):
    v = []
    for f in sorted(glob.glob(f"gpurun_out/abenv/c{i}_?.json")):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d["kernels"]
        v.append((d["value"], *[k.get(n, {}).get("ms_per_step") for n in ("gemm_qkv", "gemm_fc1", "gemm_proj", "gemm_fc2", "attention")]))
    print(f"{e:30s} img/s", [round(x[0]) for x in v], "qkv", [x[1] for x in v], "fc1", [x[2] for x in v], "proj", [x[3] for x in v], "fc2", [x[4] for x in v], "attn", [x[5] for x in v])
PY
