#!/usr/bin/env python3
"""Scan gfx950 ISA for a store-data hazard hipcc (ROCm 7.2) does not cover.

Found with tools/micro/gemm_ring4d_experiment.hip (round 3): `buffer_store_dwordx4 v[a:a+3], voff, rsrc, sN offen` -- a
MUBUF store of more than 64 bits whose soffset is an SGPR -- followed at once by a VALU write of one of its data registers
can store the NEW register value (LLVM's hazard recognizer inserts the wait state only when the soffset is not a
register).  One wait state -- any instruction -- between the two is enough.  This script lists every such store that is followed, within
`--window` instructions, by a vector instruction writing one of its data registers.

Second scan (round 4, the cause of the ring4d memory-access faults): inside an `asm` statement hipcc inserts NO hazard
wait states.  `VALU writes an SGPR -> a vector-memory instruction reads that SGPR` needs 5 wait states on gfx9 / CDNA; the
faulting build had `v_readlane_b32 s78, v126, 4 / v_readlane_b32 s79, v126, 5 / v_mov / v_mov / ;;#ASMSTART
global_atomic_add v127, v0, v102, s[78:79] sc0` -- the queue pointer reloaded from an SGPR spill two instructions ahead of
an opaque atomic, which then used the STALE pair as its base address.  scan_asm_sgpr lists every memory instruction between
;;#ASMSTART / ;;#ASMEND whose SGPR operands (or M0) were written by a VALU (v_readlane, v_readfirstlane, v_cmp ...) fewer than
`need` wait states earlier (s_nop N counts N + 1).

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


SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")
VALU_SWRITE = re.compile(r"\s*(v_readlane_b32|v_readfirstlane_b32|v_cmp\w*|v_cmpx\w*|v_add_co\w*|v_sub_co\w*|v_div_scale\w*)\s+(s\d+|s\[\d+:\d+\]|vcc)")
MEMOP = re.compile(r"\s*(global_\w+|buffer_\w+|flat_\w+|scratch_\w+|ds_\w+|s_load\w*|s_buffer_load\w*)\b(.*)")


def _sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan_asm_sgpr(path: str, need: int = 5):
    """memory instructions inside asm statements that read an SGPR a VALU wrote fewer than `need` wait states before"""
    lines = open(path).read().splitlines()
    kern, hits, n_asm_mem = "?", [], 0
    recent = []  # (wait states since, set of SGPRs written by a VALU)
    in_asm = False
    for i, l in enumerate(lines):
        t = l.strip()
        if l.startswith("_Z") and ":" in l:
            kern = l.split(":")[0]
            recent = []
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        if in_asm:
            m = MEMOP.match(t)
            if m:
                n_asm_mem += 1
                used = _sregs(m.group(2))
                for age, regs in recent:
                    if age < need and used & regs:
                        hits.append((kern, i + 1, t, sorted(used & regs), age))
                        break
        w = VALU_SWRITE.match(t)
        step = 1
        nop = re.match(r"s_nop\s+(\d+)", t)
        if nop:
            step = int(nop.group(1)) + 1
        recent = [(a + step, r) for a, r in recent if a + step < need + 2]
        if w:
            recent.append((0, _sregs(w.group(2))))
    return n_asm_mem, hits


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
        n_mem, h2 = scan_asm_sgpr(f)
        for kern, ln, ins, regs, age in h2:
            print(f"{os.path.basename(f)}:{ln} {kern[:70]}\n    {ins}\n    reads s{regs} written by a VALU {age} wait state(s) earlier (5 needed; hipcc adds none inside asm)")
        print(f"{os.path.basename(f)}: {n_mem} memory instructions inside asm statements, {len(h2)} behind a VALU write of their SGPR operands")
        bad += len(h2)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
