set -e
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r02
bash tools/profile_bf16.sh r02
bash tools/pmc_sq.sh 512 32 1 r02_pmc_sq_counters_bench_ops_512px_bs32_bf16_lds_dma.txt "--shadow 1 --layers 2,3,4,5,6"
