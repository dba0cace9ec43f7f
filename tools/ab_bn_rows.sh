# same-box A/B of the fp32 BatchNorm apply passes: row-geometry kernels (default) vs the per-item kernels (DG_OPT_BN_ITEMS=1), alternating
for v in 0 1 0 1; do
  echo "== DG_OPT_BN_ITEMS=$v"
  DG_OPT_BN_ITEMS=$v python bench.py --steps 21 --warmup 6 --no_extra --no_cpu_baseline --mfma_dtype f32x3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512px f32x3', d['value'], d['ms_per_step'])"
  DG_OPT_BN_ITEMS=$v python bench.py --steps 21 --warmup 6 --no_extra --no_cpu_baseline --mfma_dtype f32 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512px f32  ', d['value'], d['ms_per_step'])"
  DG_OPT_BN_ITEMS=$v python bench.py --image_size 64 --batch_size 64 --steps 60 --warmup 12 --no_extra --no_cpu_baseline --mfma_dtype f32 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('64px  f32  ', d['value'], d['ms_per_step'])"
  DG_OPT_BN_ITEMS=$v python bench.py --image_size 64 --batch_size 64 --steps 60 --warmup 12 --no_extra --no_cpu_baseline --mfma_dtype f32x3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('64px  f32x3', d['value'], d['ms_per_step'])"
done
