#!/bin/bash
# Assemble profiles/r01_* from what tools/prof_round.sh + tools/pmc_round.sh left under gpurun_out/ (run in the repo root)
set -e
one() {  # name, bench args, output file
  { echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py $2 --no-cpu-baseline   (tools/prof_round.sh, one MI355X)"
    echo "# bench line of the same run:"
    echo "# $(grep "^{\"metric\"" gpurun_out/prof_$1.json.log | tail -1 | cut -c1-330)"
    cat gpurun_out/prof_$1_kernel_stats.csv; } > profiles/$3
}
one vitb "" r01_bench_kernel_stats.txt
one medsam_b1 "--model medsam --batch 1" r01_medsam_b1_kernel_stats.txt
one vitg_fp8 "--model dinov2_giant14_224 --batch 32 --fp8" r01_vitg_fp8_kernel_stats.txt
one vitb_fp8 "--fp8" r01_vitb_fp8_kernel_stats.txt
cp gpurun_out/pmc_summary.txt profiles/r01_pmc_traffic.txt
cp gpurun_out/pmc_summary.json profiles/r01_pmc_traffic.json
if [ -f gpurun_out/bench_default.log ]; then
  tail -1 gpurun_out/bench_default.log > profiles/r01_bench_default.json.log
  python3 tools/roofline_table.py profiles/r01_bench_default.json.log > profiles/r01_roofline_table.md
fi
