# on the GPU box: same-box A/B of the grouped-launch schedule at 64 px (bash tools/ab_group_64.sh)
set -e
cd $GRAFT_REPO_ROOT
run() {  # label, flags...
 lab=$1; shift
 python3 bench.py --image_size 64 --no_extra --no_cpu_baseline --no_roofline --steps 60 --warmup 9 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lab', d['value'], d['ms_per_step'])"
}
for m in f32 f32x3; do
 run "bs64 $m ungrouped 2 streams" --batch_size 64 --mfma_dtype $m --group_launch off
 run "bs64 $m grouped 2 streams" --batch_size 64 --mfma_dtype $m --group_launch on
 run "bs64 $m grouped 1 stream" --batch_size 64 --mfma_dtype $m --group_launch on --single_stream
 run "bs128 $m ungrouped 2 streams" --batch_size 128 --mfma_dtype $m --group_launch off
 run "bs128 $m grouped 2 streams" --batch_size 128 --mfma_dtype $m --group_launch on
 run "bs256 $m ungrouped 2 streams" --batch_size 256 --mfma_dtype $m --group_launch off --steps 30
 run "bs256 $m grouped 2 streams" --batch_size 256 --mfma_dtype $m --group_launch on --steps 30
 run "bs256 $m grouped single plans" --batch_size 256 --mfma_dtype $m --group_launch on --group_plan single --steps 30
done
