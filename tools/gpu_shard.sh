#!/bin/bash
# GPU session for the Z-sharded path: virtual-rank tests + 1-rank RCCL dry runs of bench.py's N > 1 leg
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/pytest_gpu.log
for ch in 1 4; do
EDT_SHARD_CHUNKS=$ch EDT_BENCH_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2951$ch RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 20 --warmup 3 2>/dev/null | grep -a "^{" > gpurun_out/bench_shard_c$ch.json
python -c "
import json; d=json.load(open('gpurun_out/bench_shard_c$ch.json')); print('chunks $ch', d['ms_per_step'], d['config']['output_verified'], d['roofline'])"
done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_cfg2.json; python -c "
import json; d=json.load(open('gpurun_out/bench_cfg2.json')); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
