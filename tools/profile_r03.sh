# on the GPU box: bash tools/profile_r03.sh  -> gpurun_out/r03_* (copy what should be judged into profiles/)
# Round 3: the headline arithmetic is f32x3.  rocprofv3 --kernel-trace --stats of the bench command (2 streams + hipGraph), of the
# single-stream eager form (isolated kernel durations: what roofline.achieved is measured on) for f32x3, exact f32 and bf16, then the
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE in their own runs) over tools/bench_ops.py for the three arithmetics, then the SQ
# counters of the exact-f32 64-column input-grad tile at batch 8 (a launch short enough that SQ_VALU_MFMA_BUSY_CYCLES stays < 2^31).
set -e
RND=r03
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
bash tools/pmc_round.sh 512 32 r03 "--bf16 2 --x3planes 2 --x3cm 1" _f32x3
bash tools/pmc_round.sh 512 32 r03 "--bf16 1 --shadow 1" _bf16
bash tools/pmc_sq.sh 512 8 0 r03_pmc_sq_counters_bench_ops_512px_bs8_fp32_layers12.txt "--layers 1,2"
bash tools/pmc_sq.sh 512 32 2 r03_pmc_sq_counters_bench_ops_512px_bs32_f32x3_narrow_layers.txt "--x3planes 2 --x3cm 1 --layers 1,2"
echo "r03 profiles done"
