cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d gpurun_out/r2/pmc_kb -- python3 tools/kbench.py --gemm-only --rounds 2 --variants 22,26,28,24 --packed 1 > gpurun_out/r2/pmc_kb.log 2>&1
f=$(find gpurun_out/r2/pmc_kb -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(d.items()):
    if "gemm" in k:
        print(f"{k:72s} conflict/active {100*v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):6.2f}%  parked {100*v['SQ_WAIT_ANY']/max(v['SQ_WAVE_CYCLES'],1):5.1f}%  idx_active {v['SQ_LDS_IDX_ACTIVE']:.3g}")
PY
rm -rf gpurun_out/r2/pmc_kb
