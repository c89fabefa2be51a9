#!/bin/bash
# quick GPU session: parity tests + bench lines (cfg2, cfg3) + per-pass times
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err
python bench.py --steps 20 --warmup 3 --config cfg3 --no-cpu-baseline > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err
python bench.py --steps 20 --warmup 3 --config cfg3m --no-cpu-baseline > gpurun_out/bench_cfg3m.json 2> gpurun_out/bench_cfg3m.err
python tools/membench.py > gpurun_out/membench.log 2>&1
tail -3 gpurun_out/bench_cfg2.err
python - <<'PY'
import json
for c in ("cfg2","cfg3","cfg3m"):
    try:
        d=json.load(open(f"gpurun_out/bench_{c}.json")); print(c, d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["output_verified"])
    except Exception as e: print(c, "ERR", e)
PY
tail -4 gpurun_out/membench.log
