#!/bin/bash
# round-2 session 2: tile choice (flatness) A/B + PMC of the windowed path on cfg3
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/s2_pytest_auto.log
b() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --config ${CFG} > gpurun_out/s2_${CFG}_$tag.json 2> gpurun_out/s2_${CFG}_$tag.err; }
for CFG in cfg2 cfg3 cfg3m cfg5; do
  b auto X=1
  b hull EDT_HIP_DEBUG_MODE=0x2000
  b fd4 EDT_HIP_WINDOW_FLATDIV=4
  b fd32 EDT_HIP_WINDOW_FLATDIV=32
  b fdoff EDT_HIP_WINDOW_FLATDIV=65536
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/s2_cfg*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["output_verified"])
    except Exception as e: print(f, "ERR", e)
PY
PMC_PASSES=3 bash tools/pmc.sh s2c3 --config cfg3 2>&1 | grep -A30 "k_column_pass_wave" | head -70
