#!/bin/bash
# Round-5 closing session of the final build (through gpurun; the GPU budget left was 8 minutes): fuzz of the integer kernel's
# shapes first -- the session stops at the first mismatch --, then the whole GPU tier, smoke(), cfg3 family timings, the default
# bench line.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
{
  echo "# tools/fuzz_gpu.py on the final build (GPU vs oracle, bit for bit)"
  echo "general, 300 cases:";                                python tools/fuzz_gpu.py 300 101 2>&1 | grep "MISMATCH\|cases"
  echo "integer kernel's shapes (FUZZ_Q16=1), 150 cases:";   FUZZ_Q16=1 python tools/fuzz_gpu.py 150 102 2>&1 | grep "MISMATCH\|cases"
  echo "the same, 120 more:";                                FUZZ_Q16=1 python tools/fuzz_gpu.py 120 103 2>&1 | grep "MISMATCH\|cases"
  echo "the same, tiles beyond 16 bits as two wide passes (0x40000000), 100 cases:"; FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x40000000 python tools/fuzz_gpu.py 100 104 2>&1 | grep "MISMATCH\|cases"
} > gpurun_out/r05b_fuzz.txt 2>&1
cat gpurun_out/r05b_fuzz.txt
if grep -q "MISMATCH\|Traceback\|Error" gpurun_out/r05b_fuzz.txt; then echo "STOP: fuzz mismatch"; exit 1; fi
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r05b_gpu_tier.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r05b_gpu_tier.txt
for c in cfg3 cfg3L cfg3M; do ./tools/gpu_session.sh ab $c $c; done 2>&1 | tee gpurun_out/r05b_cfg3.txt
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/final_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["whole_job_frac"], d["config"]["output_verified"], d["config"].get("verified_by"), d["cpu_baseline"]["value"])
print(d.get("timing"))
for s in d.get("secondary", []):
    print(s["config"], s.get("ms_per_step", s.get("gpu_seconds_total")), s.get("kernel_ms"), s.get("whole_job_frac"), s.get("output_verified"), s.get("error"))
PY
