#!/bin/bash
# where do the attention kernel's wave cycles go?  SQ counters over the default bench, attention kernels only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/sq_counters.txt
want="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"
have=""
for c in $want; do grep -qx "$c" gpurun_out/sq_counters.txt && have="$have $c"; done
echo "collecting:$have"
timeout -k 10 400 rocprofv3 --pmc $have --output-format csv -d gpurun_out/pmc_attn -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_attn.log 2>&1
f=$(find gpurun_out/pmc_attn -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for row in csv.DictReader(open(sys.argv[1], newline="")):
    m = re.search(r"(attn_persist_kernel<\d>|gemm_ring3_kernel<[^>]*>)", row["Kernel_Name"])
    if not m: continue
    acc[m.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in acc.items():
    wc = v.get("SQ_WAVE_CYCLES", 1.0)
    print(k, {c: round(x / wc, 3) for c, x in v.items() if c != "SQ_WAVE_CYCLES"}, "wave_cycles %.3g" % wc)
PY
