# on the GPU box: bash tools/profile_x3.sh [ROUND]  -> gpurun_out/<ROUND>_*f32x3* (copy what should be judged into profiles/)
# the f32x3 configuration (fp32-accurate conv products from three bf16 planes per operand; plane kernel igemm_dma_x3.hip), single stream, eager
set -e
RND=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/px
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px -- python3 $R/bench.py --mfma_dtype f32x3 --no_extra --no_cpu_baseline --no_graph --single_stream > $R/gpurun_out/${RND}_bench_under_rocprof_512px_bs32_f32x3_single_stream.json 2>/tmp/ex.log
cp $(find /tmp/px -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${RND}_rocprofv3_kernel_stats__512px_bs32_f32x3_single_stream_eager.csv
echo "f32x3 profile done"
