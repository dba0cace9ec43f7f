# on the GPU box: bash tools/profile_round.sh [ROUND]  -> gpurun_out/<ROUND>_* (copy what should be judged into profiles/)
set -e
RND=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
# headline command (512 px / batch 32, 2 streams, hipGraph), headline only
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --no_extra --no_cpu_baseline > $R/gpurun_out/${RND}_bench_under_rocprof_512px_bs32_default.json 2>/tmp/e1.log
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${RND}_rocprofv3_kernel_stats__512px_bs32_default_2streams_graph.csv
echo "p1 done"
# single stream, eager: isolated kernel durations (what roofline.achieved is measured on)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 $R/bench.py --no_extra --no_cpu_baseline --no_graph --single_stream > $R/gpurun_out/${RND}_bench_under_rocprof_512px_bs32_single_stream.json 2>/tmp/e2.log
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${RND}_rocprofv3_kernel_stats__512px_bs32_single_stream_eager.csv
echo "p2 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $R/bench.py --image_size 64 --no_extra --no_cpu_baseline --no_graph --single_stream > $R/gpurun_out/${RND}_bench_under_rocprof_64px_bs256_single_stream.json 2>/tmp/e3.log
cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${RND}_rocprofv3_kernel_stats__64px_bs256_single_stream_eager.csv
echo "p3 done"
