# on the GPU box: bash tools/profile_r04.sh  -> gpurun_out/r04_* (copy what should be judged into profiles/)
# Round 4: rocprofv3 --kernel-trace --stats of the bench command at the headline workload (512 px / batch 32: default dispatch and
# single-stream eager for f32x3; single-stream eager for exact f32 and bf16), then the PMC traffic passes (FETCH_SIZE / WRITE_SIZE in
# their own runs) over tools/bench_ops.py at the metric's other configuration, 64 px / batch 64, for exact f32 and f32x3.
# (64 px / batch 64 kernel stats: tools/profile_r04_64.sh.)
set -e
RND=r04
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() {   # tag, bench flags...
  tag=$1; shift
  rm -rf /tmp/pp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/bench.py --no_extra --no_cpu_baseline "$@" > $R/gpurun_out/${RND}_bench_under_rocprof_512px_bs32_${tag}.json 2>/tmp/e_${tag}.log
  cp $(find /tmp/pp -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${RND}_rocprofv3_kernel_stats__512px_bs32_${tag}.csv
  echo "profile ${tag} done"
}
prof f32x3_default_2streams_graph
prof f32x3_single_stream_eager --no_graph --single_stream
prof f32_single_stream_eager --mfma_dtype f32 --no_graph --single_stream
prof bf16_single_stream_eager --mfma_dtype bf16 --act_dtype bf16 --no_graph --single_stream
cd $R
[ -n "$NO_PMC" ] && { echo "r04 kernel-stat profiles done (NO_PMC)"; exit 0; }
bash tools/pmc_round.sh 64 64 r04 "" ""
bash tools/pmc_round.sh 64 64 r04 "--bf16 2" _f32x3
echo "r04 profiles done"
