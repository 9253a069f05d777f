#!/usr/bin/env python3
"""Per-kernel-class roofline table from one bench.py JSON line (HIP-event timings of the profiled warm-up step):
   python tools/roofline_table.py profiles/r01_bench_default.json.log > profiles/r01_roofline_table.md"""
import json
import sys

PEAK_TF, PEAK_GBS = 2500.0, 8000.0  # MI355X_MICROARCH.md: dense bf16 MFMA, HBM3E spec
HBM_CLASSES = {"im2col", "layernorm", "final_ln", "assemble"}
NOTE = {"gemm_fc1": "mlp.fc1 + erf-GELU (LayerNorm folded in)", "gemm_fc2": "mlp.fc2 + residual (+ LN partial sums)",
        "gemm_qkv": "attn.qkv (LayerNorm folded in)", "gemm_proj": "attn.proj + residual (+ LN partial sums)",
        "attention": "softmax(q k^T / 8) v, 3072 heads of 197 x 64", "gemm_patch": "patchify conv as a GEMM (bf16 images: pixels gathered by the operand loader) + cls/pos",
        "im2col": "image -> patch rows", "layernorm": "ln_finalize: (sum, sumsq) partials -> (mean, rstd)",
        "final_ln": "final LayerNorm of the CLS rows -> fp32 out", "assemble": "cls row + stats of the token buffer",
        "cls_tail": "last block after its attention, CLS rows only: proj + residual, fc1 + GELU, fc2 + residual at M = batch (launch-latency bound)"}

line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
print(f"# Per-kernel roofline, {d['config']['workload']}")
print(f"\n`{d['metric']}`: **{d['value']} {d['unit']}**, {d['ms_per_step']} ms/step, n_gpus {d['n_gpus']}; "
      f"dominant kernel `{d['roofline']['kernel']}` {d['roofline']['achieved']} TFLOP/s = {d['roofline']['frac']} of {d['roofline']['peak']}.")
print("\nTimes: HIP events recorded by libvdr on the stream of each launch, averaged over the bench's K steps (second, fully bracketed pass); work: algorithmic FLOPs /")
print("bytes of DESIGN.md §4.  MFMA-bound classes are priced against 2.5 PFLOP/s dense bf16, HBM-bound ones against 8 TB/s.\n")
print("| kernel class | what | launches/step | ms/step | share | achieved | bound | fraction of peak |")
print("|---|---|---|---|---|---|---|---|")
tot = sum(v["ms_per_step"] for v in d["kernels"].values())
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    if k in HBM_CLASSES:
        ach, frac, bound = f"{v['GB/s_algorithmic']:.0f} GB/s", v["GB/s_algorithmic"] / PEAK_GBS, "HBM"
    else:
        ach, frac, bound = f"{v.get('TFLOP/s', 0.0):.0f} TFLOP/s", v.get("TFLOP/s", 0.0) / PEAK_TF, "MFMA"
    print(f"| `{k}` | {NOTE.get(k, '')} | {v['launches_per_step']} | {v['ms_per_step']:.3f} | {100 * v['ms_per_step'] / tot:.1f} % | {ach} | {bound} | {frac:.3f} |")
print(f"| **sum** | | | {tot:.3f} | | {d['TFLOPs_per_s']} TFLOP/s whole forward | MFMA | {d['roofline']['whole_forward_frac']} |")
if d.get("full_last_block"):
    f = d["full_last_block"]
    print(f"\nLast block: {d['config'].get('last_block', '')}.  Every token of the last block computed, as the reference does before it "
          f"keeps x[:, 0] (same run, same K steps): {f.get('value')} images/s, {f.get('ms_per_step')} ms/step; features bitwise equal: {f.get('features_bitwise_equal')}."
          f"  FLOPs per image: {d['roofline'].get('flops_per_image_executed', 0) / 1e9:.2f} G executed, {d['roofline'].get('flops_per_image_every_token', 0) / 1e9:.2f} G with every token.")
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print(f"\nCPU baseline in the same run: {c['value']} {c['unit']} on {c['cores']} cores ({c['kind']}; {c['sample']}).")
