# usage (on the GPU box): bash tools/trace_run.sh [bench flags]  -> prints the timeline analysis of an eager run
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tr
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $R/bench.py --no_512 --no_cpu_baseline --no_roofline --no_graph --steps 12 --warmup 6 "$@" > /tmp/tr.json 2>/tmp/tr.err
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
test -n "$f"
cut -c1-160 /tmp/tr.json
python3 $R/tools/timeline.py "$f" 0.5
