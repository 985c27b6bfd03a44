cd $GRAFT_REPO_ROOT
NO_PMC=1 tools/vote_variants.sh gpurun_out/r04_mock product mock_u64 product mock_u64
NO_PMC=1 BENCH_ARGS="--config c4 --cells 32 --steps 2 --warmup 1" tools/vote_variants.sh gpurun_out/r04_mock_c4 product mock_u64
