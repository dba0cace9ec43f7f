export DG_LIB=$PWD/discogan_modernized_amd/libdiscogan_hip_experiments.so
for cap in 0 -512 -1000000 0 -512; do
  echo "== DG_OPT_UNDERSTORY=$cap"
  DG_OPT_UNDERSTORY=$cap python bench.py --steps 21 --warmup 6 --no_extra --no_cpu_baseline --mfma_dtype f32x3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32x3', d['value'], d['ms_per_step'])"
  DG_OPT_UNDERSTORY=$cap python bench.py --steps 21 --warmup 6 --no_extra --no_cpu_baseline --mfma_dtype bf16 --act_dtype bf16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 ', d['value'], d['ms_per_step'])"
done
