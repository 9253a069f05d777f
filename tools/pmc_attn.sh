#!/bin/bash
# rocprofv3 PMC passes over the attention A/B (tools/attn_ab.py): where do the waves of each variant spend their time?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2
VARS=${1:-2,3}
pass() {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/r2/pmc_at_$name -- python3 tools/attn_ab.py --rounds 2 --reps 2 --variants $VARS > gpurun_out/r2/pmc_at_$name.log 2>&1 || return 1
  f=$(find gpurun_out/r2/pmc_at_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/r2/pmc_at_$name.csv
  rm -rf gpurun_out/r2/pmc_at_$name
}
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
pass b SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA &&
pass c SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES &&
pass d SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES SQ_LEVEL_WAVES
python3 - <<'PY'
import csv, glob, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/r2/pmc_at_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" in k:
            d[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    print(k)
    for c, xs in sorted(v.items()):
        print(f"   {c:34s} mean/dispatch {sum(xs)/len(xs):14.4g}   (n={len(xs)})")
PY
