set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --no_512 --no_cpu_baseline > $R/gpurun_out/r01_bench_under_rocprof_default.json 2>/tmp/e1.log
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01_rocprofv3_kernel_stats__bench_default_2streams_graph.csv
echo "p1 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 $R/bench.py --no_512 --no_cpu_baseline --no_graph --single_stream > $R/gpurun_out/r01_bench_under_rocprof_single_stream.json 2>/tmp/e2.log
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01_rocprofv3_kernel_stats__bench_single_stream_eager.csv
echo "p2 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $R/bench.py --no_512 --no_cpu_baseline --no_graph --single_stream --image_size 512 --batch_size 32 --steps 6 --warmup 3 --no_roofline > $R/gpurun_out/r01_bench_under_rocprof_512.json 2>/tmp/e3.log
cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01_rocprofv3_kernel_stats__512px_bs32_single_stream.csv
echo "p3 done"
cd $R && timeout -k 10 400 python bench.py > gpurun_out/r01_bench_default_64px_bs256.json 2>gpurun_out/bench_default.err
echo "bench done"
