#!/bin/bash
# column-group tile order: time and FETCH_SIZE of the fc1 GEMM for several group widths (VDR_GEMM_GN; 0 = row-major)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for gn in 3 4 6 5 4 6; do
  export VDR_GEMM_GN=$gn
  echo -n "GN=$gn: "; timeout -k 10 200 python3 bench.py --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items() if k.startswith('gemm_') and k!='gemm_patch'})"
done
for gn in 3 4 6; do
  export VDR_GEMM_GN=$gn
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcgn_$gn -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcgn_$gn.log 2>&1
  f=$(find gpurun_out/pmcgn_$gn -name "*counter_collection.csv" | head -1)
  echo "== FETCH GN=$gn"; python3 - "$f" <<'PY'
import csv, re, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1], newline="")):
    m = re.search(r"gemm_ring\d_kernel<[^>]*>", row["Kernel_Name"])
    if m: acc[m.group(0)].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(f"  {k:40s} n={len(v):3d} read MB (x2 corrected) = {2*sum(v)/len(v)*1024/1e6:8.1f}")
PY
done
