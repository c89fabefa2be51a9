for st in 1 2; do
echo "streams $st"
EDT_SHARD_STREAMS=$st MASTER_ADDR=127.0.0.1 MASTER_PORT=2952$st RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python tools/shard_probe.py 2>&1 | grep "^ext"
EDT_SHARD_STREAMS=$st PROBE_EXT=1024,1024,128 MASTER_ADDR=127.0.0.1 MASTER_PORT=2953$st RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python tools/shard_probe.py 2>&1 | grep "^ext"
done
EDT_SHARD_CHUNKS=4 EDT_BENCH_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 20 --warmup 3 2>/dev/null | grep -a "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench chunks 4', d['ms_per_step'], d['config']['output_verified'], d['roofline']['kernel_ms'])"
