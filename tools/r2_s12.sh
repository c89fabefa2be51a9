#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -x -q 2>&1 | tail -2
EDT_HIP_DEBUG_MODE=0x4000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -x -q 2>&1 | tail -2
python bench.py --no-cpu-baseline > gpurun_out/s12_bench.json 2> gpurun_out/s12_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/s12_bench.json"))
print(d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
for s in d.get("secondary", []): print(s["config"], s.get("ms_per_step"), s.get("kernel_ms"), s.get("whole_job_frac"), s.get("output_verified"))
PY
EDT_HIP_DEBUG_MODE=0x100000 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fp32 form', d['ms_per_step'], d['roofline']['kernel_ms'])"
python tools/fuzz_gpu.py 500 41 2>&1 | tail -1
