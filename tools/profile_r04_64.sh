# on the GPU box: bash tools/profile_r04_64.sh [TAG]  -> gpurun_out/r04_*64px_bs64* (copy what should be judged into profiles/)
# Round 4: the metric's FIRST configuration, 64 px / batch 64 (BASELINE configs[2]'s per-GPU shape, the reference CLI's default).
# rocprofv3 --kernel-trace --stats of the bench command at that shape: default dispatch (hipGraph) and single-stream eager (isolated
# kernel durations), exact f32 and f32x3.
set -e
RND=r04
TAG=${1:-}      # EXTRA (environment): further bench flags for every run, e.g. EXTRA='--group_launch off' with TAG _ungrouped
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() {   # tag, bench flags...
  tag=$1; shift
  rm -rf /tmp/pp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/bench.py --image_size 64 --batch_size 64 --no_extra --no_cpu_baseline $EXTRA "$@" > $R/gpurun_out/${RND}_bench_under_rocprof_64px_bs64_${tag}${TAG}.json 2>/tmp/e_${tag}.log || { tail -20 /tmp/e_${tag}.log; exit 1; }
  cp $(find /tmp/pp -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${RND}_rocprofv3_kernel_stats__64px_bs64_${tag}${TAG}.csv
  echo "profile ${tag} done"
}
prof f32_default_graph --mfma_dtype f32
prof f32_single_stream_eager --mfma_dtype f32 --no_graph --single_stream
prof f32x3_default_graph --mfma_dtype f32x3
prof f32x3_single_stream_eager --mfma_dtype f32x3 --no_graph --single_stream
echo "r04 64px profiles done"
