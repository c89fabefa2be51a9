cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
EDT_BENCH_FORCE_SHARDED=1 EDT_SHARD_CHUNKS=4 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_shard -o p -- python bench.py --steps 20 --warmup 3 > gpurun_out/prof_shard.log 2>&1
echo rc=$?
grep -a "^{" gpurun_out/prof_shard.log | cut -c1-400
head -8 gpurun_out/prof_shard/p_kernel_stats.csv | cut -c1-110,200-330
