cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_c4; mkdir -p $OUT
PPF_DEBUG_ACC32=1 timeout -k 10 300 python bench.py --config c4 --steps 1 --warmup 1 --cells auto --no-cpu-baseline --no-other-configs > $OUT/c4_dbg.json 2> $OUT/c4_dbg.err; echo "rc=$?"
grep "acc32" $OUT/c4_dbg.err | head -80
