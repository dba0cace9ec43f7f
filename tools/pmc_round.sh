# on the GPU box: bash tools/pmc_round.sh  -> gpurun_out/r01_pmc_traffic_per_launch_64px_bs256.json
# (counters in their own runs: --pmc with --kernel-trace only, one counter per pass)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pf /tmp/pw
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python3 $R/tools/bench_ops.py --size 64 --batch 256 > /tmp/pf.log 2>&1
echo "fetch pass done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- python3 $R/tools/bench_ops.py --size 64 --batch 256 > /tmp/pw.log 2>&1
echo "write pass done"
f=$(find /tmp/pf -name "*counter_collection.csv" | head -1); w=$(find /tmp/pw -name "*counter_collection.csv" | head -1)
test -n "$f" && test -n "$w"
python3 $R/tools/pmc_traffic.py "$f" "$w" $R/gpurun_out/r01_pmc_traffic_per_launch_64px_bs256.json
