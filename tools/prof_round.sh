#!/bin/bash
# rocprofv3 kernel-trace summaries for the benchmark lines recorded under profiles/ (run on the GPU box):
#   bash tools/prof_round.sh [names...]   default: all (names: vitb medsam_b1 medsam_b16 vitl_dense vitg_fp8 vitg_bf16)
set -e
WANT="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4prof
run() {  # name, bench args...
  name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4prof/$name -- python3 bench.py "$@" --no-cpu-baseline > gpurun_out/r4prof/$name.json.log 2>&1
  f=$(find gpurun_out/r4prof/$name -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" gpurun_out/r4prof/${name}_kernel_stats.csv; fi
  rm -rf gpurun_out/r4prof/$name
  tail -1 gpurun_out/r4prof/$name.json.log | cut -c1-200
}
for spec in "vitb" "medsam_b1 --model medsam --batch 1" "medsam_b16 --model medsam --batch 16 --steps 10" \
            "vitl_dense --model vit_large14_336 --batch 64 --out dense --steps 10" \
            "vitg_fp8 --model dinov2_giant14_224 --batch 32 --fp8 --steps 10" "vitg_bf16 --model dinov2_giant14_224 --batch 32 --steps 10"; do
  NAME=${spec%% *}
  ARGS=${spec#"$NAME"}
  if [ -z "$WANT" ] || echo " $WANT " | grep -q " $NAME "; then run $NAME $ARGS; fi
done
