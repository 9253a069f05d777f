#!/bin/bash
# Whole-forward A/B of bench.py flag sets on the shipped library, alternating processes on one box, 3 rounds each:
#   bash tools/ab_flags.sh "" "--ln-fin-fused" ...     (BENCH_ARGS="--model ... --batch ..." selects another configuration)
mkdir -p gpurun_out/abflags
for r in 1 2 3; do
  i=0
  for f in "$@"; do
    i=$((i+1))
    timeout -k 10 180 python bench.py --no-cpu-baseline ${BENCH_ARGS:---steps 40 --warmup 10} $f > gpurun_out/abflags/c${i}_$r.json 2> gpurun_out/abflags/err.log || { tail -5 gpurun_out/abflags/err.log; exit 1; }
  done
done
python - "$@" <<'PY'
import json, glob, sys
for i, e in enumerate(sys.argv[1:], 1):
    v = []
    for f in sorted(glob.glob(f"gpurun_out/abflags/c{i}_?.json")):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d["kernels"]
        v.append((d["value"], d["ms_per_step"], *[k.get(n, {}).get("ms_per_step") for n in ("gemm_qkv", "gemm_fc1", "gemm_proj", "gemm_fc2", "attention", "layernorm")]))
    print(f"{e or '(default)':24s} img/s", [round(x[0]) for x in v], "ms", [x[1] for x in v], "qkv", [x[2] for x in v], "fc1", [x[3] for x in v], "proj", [x[4] for x in v], "fc2", [x[5] for x in v], "attn", [x[6] for x in v], "ln", [x[7] for x in v])
PY
