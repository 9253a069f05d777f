mkdir -p gpurun_out/ab2
L=vit-deep-radiomics_amd/vdr/libvdr.so
for r in 1 2; do for n in a b; do
  cp ab/libvdr_$n.so $L
  timeout -k 10 150 python bench.py --no-cpu-baseline --model vit_large14_336 --batch 64 --out dense --steps 10 > gpurun_out/ab2/vitl_${n}_$r.json 2>/dev/null || exit 1
  timeout -k 10 150 python bench.py --no-cpu-baseline --model medsam --batch 16 --steps 10 > gpurun_out/ab2/sam_${n}_$r.json 2>/dev/null || exit 1
  timeout -k 10 150 python bench.py --no-cpu-baseline --model dinov2_giant14_224 --batch 32 --steps 10 > gpurun_out/ab2/vitg_${n}_$r.json 2>/dev/null || exit 1
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab2/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"], d["kernels"]["attention"]["ms_per_step"])
PY
