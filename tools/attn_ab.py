"""Interleaved A/B of the attention kernel variants at the headline shape (+ bitwise equality between them).
    python tools/attn_ab.py [--batch 256] [--seq 197] [--heads 12] [--variants 2,3] [--rounds 30]"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-deep-radiomics_amd"))
ops = importlib.import_module("vdr.ops")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seq", type=int, default=197)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--variants", default="2,3")
    ap.add_argument("--rounds", type=int, default=30)
    ap.add_argument("--reps", type=int, default=12)
    a = ap.parse_args()
    vs = [int(v) for v in a.variants.split(",")]
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(a.batch * a.seq, 3 * a.heads * 64, generator=g).bfloat16().cuda()
    outs = {v: ops.attention(qkv, a.batch, a.seq, a.heads, variant=v) for v in vs}
    torch.cuda.synchronize()
    for v in vs[1:]:
        same = torch.equal(outs[v], outs[vs[0]])
        d = (outs[v].float() - outs[vs[0]].float()).abs().max().item()
        print(f"variant {v} vs {vs[0]}: bitwise equal {same}, max |diff| {d:.3e}", flush=True)
    times = {v: [] for v in vs}
    for rnd in range(a.rounds):
        order = vs if rnd % 2 == 0 else vs[::-1]
        evs = []
        for v in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                ops.attention(qkv, a.batch, a.seq, a.heads, variant=v)
            e1.record()
            evs.append((v, e0, e1))
        torch.cuda.synchronize()
        for v, e0, e1 in evs:
            times[v].append(e0.elapsed_time(e1) / a.reps)
    fl = 4.0 * a.seq * a.seq * 64 * a.heads * a.batch
    for v in vs:
        ts = sorted(times[v])
        med = ts[len(ts) // 2]
        print(f"attention B{a.batch} N{a.seq} H{a.heads} variant {v}: median {med * 1e3:7.1f} us  min {ts[0] * 1e3:7.1f} us  "
              f"{fl / (med * 1e-3) / 1e12:6.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
