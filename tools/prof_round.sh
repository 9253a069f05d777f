#!/bin/bash
# rocprofv3 kernel-trace summaries for the three benchmark lines recorded under profiles/ (run on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # name, bench args...
  name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python3 bench.py "$@" --no-cpu-baseline > gpurun_out/prof_$name.json.log 2>&1
  f=$(find gpurun_out/prof_$name -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" gpurun_out/prof_${name}_kernel_stats.csv; fi
  tail -1 gpurun_out/prof_$name.json.log | cut -c1-200
}
run vitb
run medsam_b1 --model medsam --batch 1
run vitg_fp8 --model dinov2_giant14_224 --batch 32 --fp8
run vitb_fp8 --fp8
