#!/bin/bash
# Round-6 closing session of the FINAL build (through gpurun): the widest fuzz first -- every form-selection bit, the voxel graph,
# the sharded phases and the 2-/3-process driver, all on THIS build (VERDICT r5 "What's weak" 1: round 5's widest run was one kernel
# change behind) -- then the whole GPU tier, smoke(), the default bench line.
#   -> profiles/r06_fuzz_final.txt, r06_gpu_tier.txt, r06_bench_line.json
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
f() { echo "$1:"; shift; env "$@" 2>&1 | grep "MISMATCH\|cases\|Traceback\|Error" ; }
{
  echo "# tools/fuzz_*.py on the final build of round 6 (GPU vs oracle, bit for bit)"
  f "general, 1500 cases" python tools/fuzz_gpu.py 1500 6101
  f "general, axes up to 2100, 300 cases" FUZZ_MAX_AXIS=2100 python tools/fuzz_gpu.py 300 6102
  f "integer kernel's shapes (FUZZ_Q16=1), 600 cases" FUZZ_Q16=1 python tools/fuzz_gpu.py 600 6103
  f "the same shapes, volumes of +inf (FUZZ_INF=1: sparse structure, no border), 400 cases" FUZZ_Q16=1 FUZZ_INF=1 python tools/fuzz_gpu.py 400 6115
  f "the same, without the short cut for tiles of nothing but +inf (0x80), 150 cases" FUZZ_Q16=1 FUZZ_INF=1 EDT_HIP_DEBUG_MODE=0x80 python tools/fuzz_gpu.py 150 6116
  f "the same shapes, volumes of slabs and boxes (FUZZ_FLAT=1: tiles without structure), a random pitch of the index buffer per case (FUZZ_PAD=1), 500 cases" FUZZ_Q16=1 FUZZ_FLAT=1 FUZZ_PAD=1 python tools/fuzz_gpu.py 500 6117
  f "the same, without the short cuts for whole tiles (0x80), 150 cases" FUZZ_Q16=1 FUZZ_FLAT=1 FUZZ_PAD=1 EDT_HIP_DEBUG_MODE=0x80 python tools/fuzz_gpu.py 150 6118
  f "the same, fp32 between passes Y and Z (0x10000000), 100 cases" FUZZ_Q16=1 FUZZ_FLAT=1 EDT_HIP_DEBUG_MODE=0x10000000 python tools/fuzz_gpu.py 100 6119
  f "integer kernel's shapes, a random pitch per case (FUZZ_PAD=1), 400 cases" FUZZ_Q16=1 FUZZ_PAD=1 python tools/fuzz_gpu.py 400 6120
  f "volumes of +inf, a random pitch per case, 200 cases" FUZZ_Q16=1 FUZZ_INF=1 FUZZ_PAD=1 python tools/fuzz_gpu.py 200 6121
  f "the same, tiles beyond 16 bits as two wide passes (0x40000000), 200 cases" FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x40000000 python tools/fuzz_gpu.py 200 6104
  f "the same, no wide form (0x20000000), 200 cases" FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x20000000 python tools/fuzz_gpu.py 200 6105
  f "the same, fp32 between passes Y and Z (0x10000000), 200 cases" FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x10000000 python tools/fuzz_gpu.py 200 6106
  f "the same shapes on the fp32 kernels (0x8000000), 100 cases" FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x8000000 python tools/fuzz_gpu.py 100 6107
  f "every tile windowed (0x4000), 200 cases" EDT_HIP_DEBUG_MODE=0x4000 python tools/fuzz_gpu.py 200 6108
  f "hulls only (0x2000), 200 cases" EDT_HIP_DEBUG_MODE=0x2000 python tools/fuzz_gpu.py 200 6109
  f "fp32 form of pass X (0x100000), 200 cases" EDT_HIP_DEBUG_MODE=0x100000 python tools/fuzz_gpu.py 200 6110
  f "voxel-graph transform (FUZZ_VG=1), 600 cases" FUZZ_VG=1 python tools/fuzz_gpu.py 600 6111
  f "the two sharded phases as virtual ranks, 16-bit / fp32 records (tools/fuzz_shard.py), 400 cases" python tools/fuzz_shard.py 400 6112
  echo "the whole sharded driver, W processes sharing the GPU over gloo (tools/fuzz_driver.py):"
  python tools/fuzz_driver.py 2 300 6113 2>&1 | grep "MISMATCH\|cases\|Traceback"
  python tools/fuzz_driver.py 3 200 6114 2>&1 | grep "MISMATCH\|cases\|Traceback"
} > gpurun_out/r06_fuzz_final.txt 2>&1
cat gpurun_out/r06_fuzz_final.txt
if grep -q "MISMATCH\|Traceback\|Error" gpurun_out/r06_fuzz_final.txt; then echo "STOP: fuzz mismatch"; exit 1; fi
(time python -m pytest tests -m gpu -x -q 2>&1 | tail -8) 2>&1 | tee gpurun_out/r06_gpu_tier.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r06_gpu_tier.txt
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err
tail -2 gpurun_out/r06_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_line.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["config"]["output_verified"], d["config"].get("verified_by"), d["cpu_baseline"]["value"])
print(d.get("timing"))
for s in d.get("secondary", []):
    print(s["config"], s.get("ms_per_step", s.get("gpu_seconds_total")), s.get("kernel_ms"), s.get("whole_job_frac"), s.get("output_verified"), s.get("error"))
PY
