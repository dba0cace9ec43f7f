# on the GPU box: bash tools/pmc_round.sh [SIZE BATCH ROUND EXTRA TAG]  -> gpurun_out/<ROUND>_pmc_traffic_per_launch_<SIZE>px_bs<BATCH><TAG>.json
# EXTRA: further bench_ops.py flags, e.g. "--bf16 1 --shadow 1 --layers 2,3,4,5,6" with TAG "_bf16_lds_dma"
# (counters in their own runs: --pmc with --kernel-trace only, one counter per pass)
set -e
SIZE=${1:-64}; BATCH=${2:-256}; RND=${3:-r02}; EXTRA=${4:-}; TAG=${5:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pf /tmp/pw
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python3 $R/tools/bench_ops.py --size $SIZE --batch $BATCH --iters 3 $EXTRA > /tmp/pf.log 2>&1
echo "fetch pass done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- python3 $R/tools/bench_ops.py --size $SIZE --batch $BATCH --iters 3 $EXTRA > /tmp/pw.log 2>&1
echo "write pass done"
f=$(find /tmp/pf -name "*counter_collection.csv" | head -1); w=$(find /tmp/pw -name "*counter_collection.csv" | head -1)
test -n "$f" && test -n "$w"
python3 $R/tools/pmc_traffic.py "$f" "$w" $R/gpurun_out/${RND}_pmc_traffic_per_launch_${SIZE}px_bs${BATCH}${TAG}.json $SIZE $BATCH
