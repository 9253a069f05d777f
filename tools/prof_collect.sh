#!/bin/bash
# Assemble profiles/rNN_* from what tools/prof_round.sh + tools/pmc_round.sh (+ a plain bench run saved as
# gpurun_out/bench_default.log, tools/patch_bench.py -> gpurun_out/r2/patch_bench.json) left under gpurun_out/.
#   bash tools/prof_collect.sh r02        (run in the repo root; PROF_DIR=r4prof for the directory tools/prof_round.sh writes now)
set -e
R=${1:-r02}
one() {  # name, bench args, output file
  [ -f gpurun_out/${PROF_DIR:-prof}/$1_kernel_stats.csv ] || return 0
  { echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py $2 --no-cpu-baseline   (tools/prof_round.sh, one MI355X)"
    echo "# bench line of the same run:"
    echo "# $(grep "^{\"metric\"" gpurun_out/${PROF_DIR:-prof}/$1.json.log | tail -1 | cut -c1-330)"
    grep -v "at::native\|rocclr" gpurun_out/${PROF_DIR:-prof}/$1_kernel_stats.csv | cut -c1-260; } > profiles/$3
}
one vitb "" ${R}_bench_kernel_stats.txt
one medsam_b1 "--model medsam --batch 1" ${R}_medsam_b1_kernel_stats.txt
one medsam_b16 "--model medsam --batch 16 --steps 10" ${R}_medsam_b16_kernel_stats.txt
one vitl_dense "--model vit_large14_336 --batch 64 --out dense --steps 10" ${R}_vitl_dense_kernel_stats.txt
one vitg_fp8 "--model dinov2_giant14_224 --batch 32 --fp8 --steps 10" ${R}_vitg_fp8_kernel_stats.txt
one vitg_bf16 "--model dinov2_giant14_224 --batch 32 --steps 10" ${R}_vitg_bf16_kernel_stats.txt
cp gpurun_out/pmc_summary.txt profiles/${R}_pmc_traffic.txt
cp gpurun_out/pmc_summary.json profiles/${R}_pmc_traffic.json
[ -f gpurun_out/r2/patch_bench.json ] && cp gpurun_out/r2/patch_bench.json profiles/${R}_patch_embed_gbs.json
if [ -f gpurun_out/bench_default.log ]; then
  grep "^{\"metric\"" gpurun_out/bench_default.log | tail -1 > profiles/${R}_bench_default.json.log
  python3 tools/roofline_table.py profiles/${R}_bench_default.json.log > profiles/${R}_roofline_table.md
fi
ls -la profiles/${R}_*
