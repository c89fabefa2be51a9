#!/bin/bash
# a larger randomised parity run of the final build, every tile-choice / pass-1 form
python tools/fuzz_gpu.py 5000 101 2>&1 | tail -1
EDT_HIP_DEBUG_MODE=0x4000 python tools/fuzz_gpu.py 2500 102 2>&1 | tail -1
EDT_HIP_DEBUG_MODE=0xC000 python tools/fuzz_gpu.py 2000 103 2>&1 | tail -1
EDT_HIP_DEBUG_MODE=0x100000 python tools/fuzz_gpu.py 2000 104 2>&1 | tail -1
EDT_HIP_DEBUG_MODE=0x2000 python tools/fuzz_gpu.py 1500 105 2>&1 | tail -1
FUZZ_MAX_AXIS=2100 python tools/fuzz_gpu.py 2000 106 2>&1 | tail -1
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['steps'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['whole_job_frac'], d['config']['output_verified'])
for s in d.get('secondary', []): print(s['config'], s.get('ms_per_step'), s.get('whole_job_frac'), s.get('output_verified'))"
