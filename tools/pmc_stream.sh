#!/bin/bash
# rocprofv3 PMC passes over tools/stream_check.py (ring4 variant 26 and the stream kernel, variant 30, at the headline shapes):
# wave-time shares, matrix-pipe and LDS activity, instruction mix, per kernel.   tools/pmc_stream.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-stream}
OUT=gpurun_out/r3
mkdir -p $OUT
pass() {
  name=$1
  timeout -k 10 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc_${TAG}_$name -- python3 tools/stream_check.py --rounds 1 > $OUT/pmc_${TAG}_$name.log 2>&1 || return 1
  f=$(find $OUT/pmc_${TAG}_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/pmc_${TAG}_$name.csv
  rm -rf $OUT/pmc_${TAG}_$name
}
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" pass a &&
PMC="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" pass b &&
PMC="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" pass c
python3 - "$TAG" "$OUT" <<'PY'
import csv, glob, collections, re, sys
tag, out = sys.argv[1], sys.argv[2]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{out}/pmc_{tag}_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if re.search(r"gemm_(stream|ring4)_kernel", k) and int(r.get("Grid_Size", "0") or 0) >= 100000:
            d[k[:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    print(k)
    for c, xs in sorted(v.items()):
        print(f"   {c:34s} mean/dispatch {sum(xs)/len(xs):14.5g}   (n={len(xs)})")
    g = v.get("GRBM_GUI_ACTIVE"); m = v.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if g and m:
        print(f"   MFMA busy {100.0 * (sum(m)/len(m)) / (1024.0 * (sum(g)/len(g)) / 8.0):.1f} %")
PY
