#!/usr/bin/env python3
"""Scan gfx950 ISA for a store-data hazard hipcc (ROCm 7.2) does not cover.

Found with tools/micro/gemm_ring4d_experiment.hip (round 3): `buffer_store_dwordx4 v[a:a+3], voff, rsrc, sN offen` -- a
MUBUF store of more than 64 bits whose soffset is an SGPR -- followed at once by a VALU write of one of its data registers
can store the NEW register value (LLVM's hazard recognizer inserts the wait state only when the soffset is not a
register).  One wait state -- any instruction -- between the two is enough.  This script lists every such store that is followed, within
`--window` instructions, by a vector instruction writing one of its data registers.

    python tools/hazard_scan.py                 # compiles csrc/*.hip to ISA with hipcc (-S) and scans them
    python tools/hazard_scan.py file.s ...      # scans the given ISA listings
Exit code 1 if a match is found."""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vit-deep-radiomics_amd", "csrc")
STORE = re.compile(r"\s*(buffer_store_dwordx[34])\s+v\[(\d+):(\d+)\],\s*\S+,\s*s\[\d+:\d+\],\s*(s\d+|ttmp\d+|m0)\b")
VWRITE = re.compile(r"(v_\w+|ds_read\w*|global_load\w*|buffer_load\w*)\s+(?:v(\d+)|v\[(\d+):(\d+)\])")


def scan(path: str, window: int):
    lines = open(path).read().splitlines()
    kern, total, hits = "?", 0, []
    for i, l in enumerate(lines):
        if l.startswith("_Z") and ":" in l:
            kern = l.split(":")[0]
        m = STORE.match(l)
        if not m:
            continue
        total += 1
        lo, hi = int(m.group(2)), int(m.group(3))
        seen, j = 0, i + 1
        while j < len(lines) and seen < window:
            t = lines[j].strip()
            j += 1
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            seen += 1
            w = VWRITE.match(t)
            if w and not t.startswith("v_cmp"):
                a = int(w.group(2)) if w.group(2) else int(w.group(3))
                b = int(w.group(2)) if w.group(2) else int(w.group(4))
                if a <= hi and b >= lo:
                    hits.append((kern, i + 1, l.strip(), t))
                    break
    return total, hits


def compile_isa(sources, jobs: int = 4):
    """hipcc -S of the given csrc files (the library's flags), `jobs` at a time; returns the .s paths"""
    tmp = tempfile.mkdtemp(prefix="vdr_isa_")
    outs, running = [], []
    for src in sources:
        out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
        flags = []
        running.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast"] + flags +
                                        ["-S", "--cuda-device-only", src, "-o", out], cwd=CSRC, stderr=subprocess.DEVNULL))
        outs.append(out)
        if len(running) >= jobs:
            if running.pop(0).wait() != 0:
                raise RuntimeError("hipcc -S failed")
    for pr in running:
        if pr.wait() != 0:
            raise RuntimeError("hipcc -S failed")
    return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--window", type=int, default=2, help="instructions behind the store that are checked")
    a = ap.parse_args()
    files = a.files
    if not files:
        files = compile_isa(sorted(glob.glob(os.path.join(CSRC, "*.hip"))))
    bad = 0
    for f in files:
        total, hits = scan(f, a.window)
        for kern, ln, st, wr in hits:
            print(f"{os.path.basename(f)}:{ln} {kern[:70]}\n    {st}\n    -> {wr}")
        print(f"{os.path.basename(f)}: {total} wide buffer stores with an SGPR soffset, {len(hits)} followed by a write of their data")
        bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
