#!/bin/bash
# GPU session for the Z-sharded path: virtual-rank tests + 1-rank RCCL dry run of bench.py's N > 1 leg
mkdir -p gpurun_out
python -m pytest tests/test_gpu_paths.py -m gpu -x -q --durations=12 2>&1 | tail -25 | tee gpurun_out/pytest_paths.log
EDT_BENCH_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_shard1.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/bench_shard1.log | cut -c1-600
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > gpurun_out/bench_cfg2.json; python -c "
import json; d=json.load(open('gpurun_out/bench_cfg2.json')); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
