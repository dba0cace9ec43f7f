set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dp_gpu.py -x -q 2>&1 | tail -5
for ov in off graph off graph; do
python3 bench.py --image_size 64 --batch_size 64 --no_extra --no_cpu_baseline --no_roofline --steps 60 --warmup 9 --mfma_dtype f32 --comm capi --overlap $ov 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bs64 f32 1-rank capi overlap=$ov', d['value'], d['ms_per_step'], d['config']['allreduce_overlap'])"
done
