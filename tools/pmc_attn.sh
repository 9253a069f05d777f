#!/bin/bash
# PMC rows for the attention kernels of the secondary configurations (the headline's are in tools/pmc_round.sh):
#   bash tools/pmc_attn.sh <name> <bench args...>     e.g.  bash tools/pmc_attn.sh vitl --model vit_large14_336 --batch 64 --out dense
# Four rocprofv3 --pmc passes (one counter group each, no trace domains), summarised per kernel whose name matches "attn".
set -e
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_attn
pass() {
  p=$1; shift
  timeout -k 10 500 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_attn/${NAME}_$p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $BARGS > gpurun_out/pmc_attn/${NAME}_$p.log 2>&1
  f=$(find gpurun_out/pmc_attn/${NAME}_$p -name "*counter_collection.csv" | head -1)
  cp "$f" gpurun_out/pmc_attn/${NAME}_$p.csv
  rm -rf gpurun_out/pmc_attn/${NAME}_$p
}
BARGS="$*"
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
pass sq2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python3 - "$NAME" "$BARGS" <<'PY'
import collections, csv, re, sys
name, bargs = sys.argv[1:3]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("fetch", "write", "sq", "sq2"):
    for row in csv.DictReader(open(f"gpurun_out/pmc_attn/{name}_{p}.csv")):
        k = row["Kernel_Name"]
        if "attn" in k:
            per[re.sub(r"\(.*", "", k).replace("void vdr::", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
s = lambda v: sum(v) if v else 0.0
out = [f"# rocprofv3 --pmc passes (tools/pmc_attn.sh) over  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline {bargs}",
       "# kernel | dispatches | read MB (FETCH_SIZE x 2: gfx950 correction) | write MB | MFMA busy % | L2 hit % | parked % | issue-stalled % | active % | LDS conflict %"]
for k, c in sorted(per.items()):
    n = len(c["FETCH_SIZE"])
    wc = s(c["SQ_WAVE_CYCLES"]) or 1.0
    ia = s(c["SQ_LDS_IDX_ACTIVE"]) or 1.0
    out.append(f"{k} | {n} | {2.0 * s(c['FETCH_SIZE']) / n * 1024 / 1e6:.1f} | {s(c['WRITE_SIZE']) / max(len(c['WRITE_SIZE']), 1) * 1024 / 1e6:.1f} | "
               f"{100.0 * s(c['SQ_VALU_MFMA_BUSY_CYCLES']) / (1024.0 * s(c['GRBM_GUI_ACTIVE']) / 8.0):.1f} | "
               f"{100.0 * s(c['TCC_HIT_sum']) / ((s(c['TCC_HIT_sum']) + s(c['TCC_MISS_sum'])) or 1.0):.1f} | "
               f"{100.0 * s(c['SQ_WAIT_ANY']) / wc:.1f} | {100.0 * s(c['SQ_WAIT_INST_ANY']) / wc:.1f} | {100.0 * s(c['SQ_ACTIVE_INST_ANY']) / wc:.1f} | "
               f"{100.0 * s(c['SQ_LDS_BANK_CONFLICT']) / ia:.1f}")
open(f"gpurun_out/pmc_attn/{name}_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
