# on the GPU box: bash tools/pmc_sq.sh [SIZE BATCH BF16 OUT EXTRA]  -> mean SQ counters per dispatch for every kernel of tools/bench_ops.py
# EXTRA: further bench_ops.py flags, e.g. "--shadow 1 --layers 2,3,4,5,6" (the LDS-DMA kernel on bf16 shadow operands)
set -e
SIZE=${1:-64}; BATCH=${2:-256}; BF=${3:-0}; OUT=${4:-pmc_sq.txt}; EXTRA=${5:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/psq
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/psq -- python3 $R/tools/bench_ops.py --size $SIZE --batch $BATCH --bf16 $BF --iters 3 $EXTRA > /tmp/psq.log 2>&1 || { tail -5 /tmp/psq.log; exit 1; }
f=$(find /tmp/psq -name "*counter_collection.csv" | head -1)
test -n "$f"
python3 $R/tools/pmc_summary.py "$f" > $R/gpurun_out/$OUT
