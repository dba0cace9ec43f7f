# same-box A/B: conv1's activation backward inside the input-gradient / weight-gradient kernels (default) vs the stand-alone act_bwd pass (DG_FUSE_C3_DGRAD_ACT=0)
for v in 1 0 1 0; do
  echo "== DG_FUSE_C3_DGRAD_ACT=$v"
  DG_FUSE_C3_DGRAD_ACT=$v python bench.py --steps 21 --warmup 6 --no_extra --no_cpu_baseline --mfma_dtype f32x3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512px f32x3', d['value'], d['ms_per_step'])"
  DG_FUSE_C3_DGRAD_ACT=$v python bench.py --steps 21 --warmup 6 --no_extra --no_cpu_baseline --mfma_dtype bf16 --act_dtype bf16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512px bf16 ', d['value'], d['ms_per_step'])"
  DG_FUSE_C3_DGRAD_ACT=$v python bench.py --steps 21 --warmup 6 --no_extra --no_cpu_baseline --mfma_dtype f32 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512px f32  ', d['value'], d['ms_per_step'])"
done
