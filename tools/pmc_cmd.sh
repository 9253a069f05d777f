#!/bin/bash
# rocprofv3 PMC passes (wave-time shares, pipe activity, instruction mix) over any command, per kernel matching a regex:
#   tools/pmc_cmd.sh <tag> <kernel-regex> -- python3 bench.py --model medsam --batch 16 --steps 2 --warmup 1 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; RX=$2; shift 3
mkdir -p gpurun_out/r2
pass() {
  name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc $PMC --output-format csv -d gpurun_out/r2/pmc_${TAG}_$name -- "$@" > gpurun_out/r2/pmc_${TAG}_$name.log 2>&1 || return 1
  f=$(find gpurun_out/r2/pmc_${TAG}_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/r2/pmc_${TAG}_$name.csv
  rm -rf gpurun_out/r2/pmc_${TAG}_$name
}
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" pass a "$@" &&
PMC="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" pass b "$@" &&
PMC="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" pass c "$@" &&
PMC="FETCH_SIZE" pass d "$@" &&
PMC="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" pass e "$@"
python3 - "$TAG" "$RX" <<'PY'
import csv, glob, collections, re, sys
tag, rx = sys.argv[1], re.compile(sys.argv[2])
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"gpurun_out/r2/pmc_{tag}_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if rx.search(k):
            d[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    print(k)
    for c, xs in sorted(v.items()):
        print(f"   {c:34s} mean/dispatch {sum(xs)/len(xs):14.5g}   (n={len(xs)})")
PY
