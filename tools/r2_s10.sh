#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_gpu_voxel_graph.py -m gpu -x -q 2>&1 | tail -3
EDT_HIP_DEBUG_MODE=0x4000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -x -q 2>&1 | tail -3
EDT_HIP_DEBUG_MODE=0xC000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -x -q 2>&1 | tail -3
python tools/vg_probe.py 2>&1 | tail -1
python bench.py --no-cpu-baseline > gpurun_out/s10_bench.json 2> gpurun_out/s10_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/s10_bench.json"))
print(d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
for s in d.get("secondary", []): print(s["config"], s.get("ms_per_step"), s.get("kernel_ms"), s.get("whole_job_frac"), s.get("output_verified"))
PY
python tools/fuzz_gpu.py 400 21 2>&1 | tail -1
EDT_HIP_DEBUG_MODE=0x4000 python tools/fuzz_gpu.py 300 22 2>&1 | tail -1
EDT_HIP_DEBUG_MODE=0xC000 python tools/fuzz_gpu.py 200 23 2>&1 | tail -1
