#!/bin/bash
# rocprofv3 PMC passes over the default bench (one counter group per pass, no trace domains), summarised into
# profiles/rNN_pmc_traffic.{txt,json} by tools/pmc_summarize.py.  Run on the GPU box:  tools/pmc_round.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pass() {  # name, counters...
  name=$1; shift
  timeout -k 10 500 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --full-last-block > gpurun_out/pmc_$name.log 2>&1
  f=$(find gpurun_out/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/pmc_$name.csv
  echo "pass $name: $(wc -l < gpurun_out/pmc_$name.csv) rows"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
pass sq2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python3 tools/pmc_summarize.py gpurun_out/pmc_fetch.csv gpurun_out/pmc_write.csv gpurun_out/pmc_sq.csv gpurun_out/pmc_summary gpurun_out/pmc_sq2.csv
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq gpurun_out/pmc_sq2
