#!/bin/bash
# Whole-forward A/B of several builds of libvdr.so: bench.py in alternating processes on one box (the builds sit under tools/ab_base/,
# git-ignored but not gpurun-ignored: ab/ does not travel).   bash tools/ab_bench_libs.sh "pytest -k expr" lib1.so lib2.so ...   (the LAST library stays installed)
mkdir -p gpurun_out/ab
L=vit-deep-radiomics_amd/vdr/libvdr.so
K="$1"; shift
for lib in "$@"; do
  n=$(basename $lib .so)
  cp $lib $L && timeout -k 10 500 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -k "$K" > gpurun_out/ab/tests_$n.log 2>&1 || { tail -30 gpurun_out/ab/tests_$n.log; exit 1; }
  echo "$n: $(tail -1 gpurun_out/ab/tests_$n.log)"
done
for r in 1 2 3; do
  for lib in "$@"; do
    n=$(basename $lib .so)
    cp $lib $L && timeout -k 10 120 python bench.py --no-cpu-baseline --steps 40 --warmup 10 > gpurun_out/ab/${n}_$r.json 2> gpurun_out/ab/err.log || exit 1
  done
done
python - <<'PY'
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/ab/*_?.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d["kernels"]
    acc[f.split("/")[-1].rsplit("_", 1)[0]].append((d["value"], k["gemm_proj"]["ms_per_step"], k["gemm_fc2"]["ms_per_step"], k["gemm_qkv"]["ms_per_step"], k["gemm_fc1"]["ms_per_step"], k["layernorm"]["ms_per_step"]))
for n, v in acc.items():
    print(n, "img/s", [round(x[0]) for x in v], "proj", [x[1] for x in v], "fc2", [x[2] for x in v], "qkv", [x[3] for x in v], "fc1", [x[4] for x in v], "ln", [x[5] for x in v])
PY
