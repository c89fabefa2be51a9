#!/bin/bash
# Round-4 closing session (through gpurun): the whole GPU tier, smoke(), fuzz (general + the integer kernel's shapes, plane on /
# off / no integer kernel), the round's profile, the default bench line.  Results under gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r04_gpu_tier.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r04_gpu_tier.txt
{
  echo "# tools/fuzz_gpu.py on the final build (GPU vs oracle, bit for bit)"
  echo "general, 600 cases:";                                  python tools/fuzz_gpu.py 600 81 2>&1 | tail -1
  echo "general, axes up to 2100, 200 cases:";                 FUZZ_MAX_AXIS=2100 python tools/fuzz_gpu.py 200 82 2>&1 | tail -1
  echo "integer kernel's shapes (FUZZ_Q16=1), 250 cases:";     FUZZ_Q16=1 python tools/fuzz_gpu.py 250 83 2>&1 | tail -1
  echo "the same, fp32 between passes Y and Z (0x10000000), 120 cases:"; FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x10000000 python tools/fuzz_gpu.py 120 84 2>&1 | tail -1
  echo "the same shapes on the fp32 kernels (0x8000000), 80 cases:";     FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x8000000 python tools/fuzz_gpu.py 80 85 2>&1 | tail -1
  echo "every tile windowed (0x4000), 150 cases:";             EDT_HIP_DEBUG_MODE=0x4000 python tools/fuzz_gpu.py 150 86 2>&1 | tail -1
  echo "hulls only (0x2000), 150 cases:";                      EDT_HIP_DEBUG_MODE=0x2000 python tools/fuzz_gpu.py 150 87 2>&1 | tail -1
  echo "voxel-graph transform (FUZZ_VG=1), 300 cases:";        FUZZ_VG=1 python tools/fuzz_gpu.py 300 88 2>&1 | tail -1
  echo "the same, fp32 form of its pass X (0x100000), 150 cases:";       FUZZ_VG=1 EDT_HIP_DEBUG_MODE=0x100000 python tools/fuzz_gpu.py 150 89 2>&1 | tail -1
  echo "the same on the fp32 column kernels (0x8000000), 150 cases:";    FUZZ_VG=1 EDT_HIP_DEBUG_MODE=0x8000000 python tools/fuzz_gpu.py 150 90 2>&1 | tail -1
  echo "the two sharded phases as virtual ranks, 16-bit / fp32 records (tools/fuzz_shard.py), 400 cases:"; python tools/fuzz_shard.py 400 11 2>&1 | tail -1
  echo "the whole sharded driver, W processes sharing the GPU over gloo (tools/fuzz_driver.py):"
  for a in "2 500 21" "3 400 22" "4 300 23"; do python tools/fuzz_driver.py $a 2>&1 | grep "^world\|MISMATCH" | tail -3; done
} > gpurun_out/r04_fuzz.txt 2>&1
cat gpurun_out/r04_fuzz.txt
# (SKIP_PROFILE=1: a closing session after a change that left the profiled kernels alone)
if [ -z "$SKIP_PROFILE" ]; then ./tools/profile_r04.sh > gpurun_out/r04_profile.log 2>&1; tail -2 gpurun_out/r04_profile.log; fi
python tools/rank_shape_probe.py 2>&1 | tail -3 | tee gpurun_out/r04_rank_shape_probe.txt
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/final_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["whole_job_frac"], d["config"]["output_verified"], d["cpu_baseline"]["value"])
for s in d.get("secondary", []):
    print(s["config"], s.get("ms_per_step", s.get("gpu_seconds_total")), s.get("kernel_ms"), s.get("whole_job_frac"), s.get("output_verified"), s.get("error"))
PY
