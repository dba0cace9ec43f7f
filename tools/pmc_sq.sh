# on the GPU box: bash tools/pmc_sq.sh  -> prints mean SQ counters per dispatch for the igemm kernels
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/psq
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/psq -- python3 $R/tools/bench_ops.py --size 64 --batch 256 > /tmp/psq.log 2>&1 || { tail -5 /tmp/psq.log; exit 1; }
f=$(find /tmp/psq -name "*counter_collection.csv" | head -1)
test -n "$f"
python3 $R/tools/pmc_summary.py "$f" | grep -A8 "igemm_kernel" < /dev/null || python3 $R/tools/pmc_summary.py "$f" > $R/gpurun_out/pmc_sq.txt
python3 $R/tools/pmc_summary.py "$f" > $R/gpurun_out/pmc_sq.txt
