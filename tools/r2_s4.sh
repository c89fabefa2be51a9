#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/s5_pytest_auto.log
EDT_HIP_DEBUG_MODE=0x4000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/s5_pytest_win.log
EDT_HIP_DEBUG_MODE=0xC000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/s5_pytest_win64.log
b() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --config ${CFG} > gpurun_out/s5_${CFG}_$tag.json 2> gpurun_out/s5_${CFG}_$tag.err; }
for CFG in cfg2 cfg3 cfg3m cfg5; do
  b auto X=1
  b hull EDT_HIP_DEBUG_MODE=0x2000
  b force EDT_HIP_DEBUG_MODE=0x4000
  b lim64 EDT_HIP_WINDOW_LIMIT=64
  b lim256 EDT_HIP_WINDOW_LIMIT=256
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s5_cfg*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["output_verified"])
    except Exception as e: print(f, "ERR", e)
PY
