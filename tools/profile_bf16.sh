# on the GPU box: bash tools/profile_bf16.sh [ROUND]  -> gpurun_out/<ROUND>_*bf16* (copy what should be judged into profiles/)
# the bf16 configuration (BASELINE configs[4] arithmetic: bf16 MFMA, bf16-stored feature maps, fp32 BatchNorm statistics), single stream, eager
set -e
RND=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pb
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -- python3 $R/bench.py --mfma_dtype bf16 --act_dtype bf16 --no_extra --no_cpu_baseline --no_graph --single_stream > $R/gpurun_out/${RND}_bench_under_rocprof_512px_bs32_bf16_single_stream.json 2>/tmp/eb.log
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${RND}_rocprofv3_kernel_stats__512px_bs32_bf16_single_stream_eager.csv
echo "bf16 profile done"
