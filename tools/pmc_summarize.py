#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files (tools/pmc_round.sh) per kernel class.
usage: pmc_summarize.py fetch.csv write.csv sq.csv out_prefix [sq2.csv]  ->  out_prefix.txt / out_prefix.json"""
import csv
import json
import re
import sys
from collections import defaultdict

CLASSES = [  # (label, regex on the demangled kernel name: gemm_ring4[p]_kernel<WAVES_M, WAVES_N, EPI, TAG>; TAG 1 = K > N;
    # gemm_8p_kernel<EPI, FOLD> = tile variant 31, the default of the qkv and fc1 launches since round 4)
    ("gemm_qkv (EPI_BIAS, 8-phase)", r"gemm_8p_kernel<\(?(vdr::)?(Epilogue)?\)?0, "),
    ("gemm_fc1 (EPI_BIAS_GELU, 8-phase)", r"gemm_8p_kernel<\(?(vdr::)?(Epilogue)?\)?1, "),
    ("gemm_fc1 (EPI_BIAS_GELU)", r"gemm_ring\dp?_kernel<\d+, \d+, 1, \d+>"),
    ("gemm_proj (EPI_BIAS_RESID, K = N)", r"gemm_ring\dp?_kernel<\d+, \d+, 2, 0>"),
    ("gemm_fc2 (EPI_BIAS_RESID, K > N)", r"gemm_ring\dp?_kernel<\d+, \d+, 2, 1>"),
    ("gemm_qkv (EPI_BIAS)", r"gemm_ring\dp?_kernel<\d+, \d+, 0, \d+>"),
    ("attention", r"attn_(persist_)?kernel"),
    ("ln_finalize", r"ln_finalize_kernel"),
]
# algorithmic MB per launch at the headline shape (M = 50432): A + W + C (+ residual)
M = 50432
ALG = {"gemm_fc1 (EPI_BIAS_GELU)": (M * 768 + 3072 * 768 + M * 3072) * 2 / 1e6,
       "gemm_qkv (EPI_BIAS)": (M * 768 + 2304 * 768 + M * 2304) * 2 / 1e6,
       "gemm_qkv (EPI_BIAS, 8-phase)": (M * 768 + 2304 * 768 + M * 2304) * 2 / 1e6,
       "gemm_fc1 (EPI_BIAS_GELU, 8-phase)": (M * 768 + 3072 * 768 + M * 3072) * 2 / 1e6,
       "gemm_proj (EPI_BIAS_RESID, K = N)": (M * 768 + 768 * 768 + 2 * M * 768) * 2 / 1e6,   # A + W + residual in + out
       "gemm_fc2 (EPI_BIAS_RESID, K > N)": (M * 3072 + 768 * 3072 + 2 * M * 768) * 2 / 1e6,
       "attention": (M * 2304 + M * 768) * 2 / 1e6}


def load(path):
    per = defaultdict(lambda: defaultdict(list))  # class -> counter -> [value per dispatch]
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            for label, rx in CLASSES:
                if re.search(rx, name):
                    per[label][row["Counter_Name"]].append(float(row["Counter_Value"]))
                    break
    return per


def main():
    fetch, write, sq, out = sys.argv[1:5]
    pf, pw, ps = load(fetch), load(write), load(sq)
    p2 = load(sys.argv[5]) if len(sys.argv) > 5 else None
    res = {}
    lines = ["# rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ/TCC/GRBM, one pass each; tools/pmc_round.sh) over",
             "#   python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --full-last-block   (every launch of a class full-size; one MI355X)",
             "# FETCH_SIZE / WRITE_SIZE are KB per dispatch.  On gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane)",
             "# coalesced reads (MI355X_MICROARCH.md, HBM): 'read MB' doubles it; WRITE_SIZE is exact for 16-B stores.",
             "# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); values are means per dispatch.",
             "# wave time (sq2 pass, quad-cycle units, shares of SQ_WAVE_CYCLES): parked = SQ_WAIT_ANY (s_waitcnt / barrier),",
             "#   issue-stalled = SQ_WAIT_INST_ANY, active = SQ_ACTIVE_INST_ANY; LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.",
             "# kernel | dispatches | read MB (corrected) | write MB | algorithmic MB | MFMA busy % | L2 hit % | parked % | issue-stalled % | active % | LDS conflict %"]
    for label, _ in CLASSES:
        f = pf[label].get("FETCH_SIZE", [])
        w = pw[label].get("WRITE_SIZE", [])
        if not f:
            continue
        n = len(f)
        read_mb = 2.0 * sum(f) / n * 1024 / 1e6
        write_mb = sum(w) / max(len(w), 1) * 1024 / 1e6
        busy = ps[label].get("SQ_VALU_MFMA_BUSY_CYCLES", [])
        act = ps[label].get("GRBM_GUI_ACTIVE", [])
        hit, miss = ps[label].get("TCC_HIT_sum", []), ps[label].get("TCC_MISS_sum", [])
        mfma = 100.0 * sum(busy) / (1024.0 * sum(act) / 8.0) if busy and act and sum(act) > 0 else 0.0
        l2 = 100.0 * sum(hit) / (sum(hit) + sum(miss)) if hit and (sum(hit) + sum(miss)) > 0 else 0.0
        extra = [0.0, 0.0, 0.0, 0.0]
        if p2 is not None and p2[label].get("SQ_WAVE_CYCLES"):
            q = p2[label]
            wc = sum(q["SQ_WAVE_CYCLES"])
            extra = [100.0 * sum(q.get(k, [0.0])) / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")]
            ia = sum(q.get("SQ_LDS_IDX_ACTIVE", [0.0]))
            extra.append(100.0 * sum(q.get("SQ_LDS_BANK_CONFLICT", [0.0])) / ia if ia > 0 else 0.0)
        res[label] = {"read_mb_corrected": round(read_mb, 1), "write_mb": round(write_mb, 1),
                      "algorithmic_mb": round(ALG.get(label, 0.0), 1), "mfma_busy_pct": round(mfma, 1), "l2_hit_pct": round(l2, 1),
                      "parked_pct": round(extra[0], 1), "issue_stalled_pct": round(extra[1], 1), "active_pct": round(extra[2], 1),
                      "lds_conflict_pct": round(extra[3], 1)}
        lines.append(f"{label:32s} | {n:3d} | {read_mb:8.1f} | {write_mb:7.1f} | {ALG.get(label, 0.0):7.1f} | {mfma:5.1f} | {l2:5.1f} | "
                     f"{extra[0]:5.1f} | {extra[1]:5.1f} | {extra[2]:5.1f} | {extra[3]:5.1f}")
    # the kernels these counters belong to: bench.py drops `roofline.traffic` when the library it runs was built from
    # other sources (vdr.source_id() = sha256 over csrc/)
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-deep-radiomics_amd"))
    try:
        import vdr
        res["_source_id"] = vdr.source_id()
        lines.append(f"# kernel sources: vdr.source_id() = {res['_source_id']}")
    except Exception as e:  # noqa: BLE001
        lines.append(f"# kernel sources: unknown ({e})")
    open(out + ".txt", "w").write("\n".join(lines) + "\n")
    json.dump(res, open(out + ".json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
