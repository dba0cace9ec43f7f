# on the GPU box: same-box A/B at 512 px / batch 32, one rank: the discriminators' Adam behind the D-step graph on the main stream
# (--overlap off) against on the communication stream under the next iteration's first graph (--overlap graph)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for ov in off graph; do
 for m in "f32x3" "bf16 --act_dtype bf16"; do
 python3 bench.py --no_extra --no_cpu_baseline --no_roofline --steps 12 --warmup 6 --mfma_dtype $m --overlap $ov 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('512 $m overlap=$ov', d['value'], d['ms_per_step'], d['config']['allreduce_overlap'])"
 done; done; done
