#!/bin/bash
# what the driver runs at round end (full GPU tier, smoke, bench) + fuzz in every mode
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/final_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["whole_job_frac"], d["config"]["output_verified"], d["cpu_baseline"]["value"])
for s in d.get("secondary", []): print(s["config"], s.get("ms_per_step"), s.get("whole_job_frac"), s.get("output_verified"))
PY
python tools/fuzz_gpu.py 1500 51 2>&1 | tail -1
EDT_HIP_DEBUG_MODE=0x4000 python tools/fuzz_gpu.py 800 52 2>&1 | tail -1
EDT_HIP_DEBUG_MODE=0xC000 python tools/fuzz_gpu.py 500 53 2>&1 | tail -1
EDT_HIP_DEBUG_MODE=0x100000 python tools/fuzz_gpu.py 500 54 2>&1 | tail -1
FUZZ_MAX_AXIS=2100 python tools/fuzz_gpu.py 500 55 2>&1 | tail -1
