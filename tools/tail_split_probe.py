"""Would running the last partial round of a GEMM's tiles as 128x128 tiles on a second stream pay?
Times one launch (variant 26, M = 50432) against the same rows split into a main part (whole 512-tile rounds of
128x256 tiles) on one stream and the remaining rows as 128x128 tiles (variant 28) on another, joined by events.
    python tools/tail_split_probe.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-deep-radiomics_amd"))
ops = importlib.import_module("vdr.ops")
vdr = importlib.import_module("vdr")


def main():
    M = 50432
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for name, N, K in (("proj", 768, 768), ("fc2", 768, 3072), ("fc1", 3072, 768), ("qkv", 2304, 768)):
        epi = vdr.EPI_BIAS_RESID if N == 768 else vdr.EPI_BIAS
        x = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        Wp = ops.pack_linear_weight(W)
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda").bfloat16() if epi == vdr.EPI_BIAS_RESID else None
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        tn = (N + 255) // 256
        tiles = ((M + 127) // 128) * tn
        full_rounds = tiles // 512
        rows_main = (full_rounds * 512 // tn) * 128
        rows_main = min(rows_main, M)

        def whole():
            ops.linear(x, Wp, b, resid=r, epilogue=epi, variant=26, out=out, packed=True)

        def split():
            cur = torch.cuda.current_stream()
            e0 = torch.cuda.Event()
            e0.record(cur)
            s2.wait_event(e0)
            ops.linear(x[:rows_main], Wp, b, resid=None if r is None else r[:rows_main], epilogue=epi, variant=26, out=out2[:rows_main], packed=True)
            with torch.cuda.stream(s2):
                ops.linear(x[rows_main:], Wp, b, resid=None if r is None else r[rows_main:], epilogue=epi, variant=28, out=out2[rows_main:], packed=True)
                e1 = torch.cuda.Event()
                e1.record(s2)
            cur.wait_event(e1)

        whole(); split()
        torch.cuda.synchronize()
        same = torch.equal(out, out2)
        tw, ts = [], []
        for rnd in range(14):
            for fn, acc in ((whole, tw), (split, ts)) if rnd % 2 == 0 else ((split, ts), (whole, tw)):
                fn()
                a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                fn()
                z.record()
                torch.cuda.synchronize()
                acc.append(a.elapsed_time(z) / 2)
        tw.sort(); ts.sort()
        print(f"{name}: tiles {tiles} = {full_rounds} rounds + {tiles - full_rounds * 512}; main rows {rows_main}; one launch {tw[len(tw)//2]*1e3:.1f} us, "
              f"split {ts[len(ts)//2]*1e3:.1f} us; bitwise equal {same}", flush=True)


if __name__ == "__main__":
    main()
