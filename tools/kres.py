#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks: one line per kernel."""
import re, sys, subprocess
for path in sys.argv[1:]:
    cur = {}
    for line in open(path, errors="ignore"):
        m = re.search(r"remark: [^:]+:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
        if not m: continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            if cur: print(cur)
            name = t.split(":", 1)[1].strip()
            try: name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
            except Exception: pass
            cur = {"k": re.sub(r"\(.*", "", name)}
        else:
            k, _, v = t.partition(":")
            k = k.strip()
            if k in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "TotalSGPRs"):
                cur[k.split()[0]] = v.strip()
    if cur: print(cur)
