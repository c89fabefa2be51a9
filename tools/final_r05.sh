#!/bin/bash
# Round-5 closing session (through gpurun): the whole GPU tier, smoke(), fuzz, the rank-shape probe, the default bench line.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r05_gpu_tier.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r05_gpu_tier.txt
{
  echo "# tools/fuzz_gpu.py on the final build (GPU vs oracle, bit for bit)"
  echo "general, 400 cases:";                                  python tools/fuzz_gpu.py 400 81 2>&1 | tail -1
  echo "general, axes up to 2100, 150 cases:";                 FUZZ_MAX_AXIS=2100 python tools/fuzz_gpu.py 150 82 2>&1 | tail -1
  echo "integer kernel's shapes (FUZZ_Q16=1), 250 cases:";     FUZZ_Q16=1 python tools/fuzz_gpu.py 250 83 2>&1 | tail -1
  echo "the same, tiles beyond 16 bits as two wide passes (0x40000000), 120 cases:"; FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x40000000 python tools/fuzz_gpu.py 120 96 2>&1 | tail -1
  echo "the same, no wide form (0x20000000), 120 cases:";      FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x20000000 python tools/fuzz_gpu.py 120 97 2>&1 | tail -1
  echo "the same, fp32 between passes Y and Z (0x10000000), 100 cases:"; FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x10000000 python tools/fuzz_gpu.py 100 84 2>&1 | tail -1
  echo "the same shapes on the fp32 kernels (0x8000000), 60 cases:";     FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x8000000 python tools/fuzz_gpu.py 60 85 2>&1 | tail -1
  echo "every tile windowed (0x4000), 100 cases:";             EDT_HIP_DEBUG_MODE=0x4000 python tools/fuzz_gpu.py 100 86 2>&1 | tail -1
  echo "hulls only (0x2000), 100 cases:";                      EDT_HIP_DEBUG_MODE=0x2000 python tools/fuzz_gpu.py 100 87 2>&1 | tail -1
  echo "voxel-graph transform (FUZZ_VG=1), 200 cases:";        FUZZ_VG=1 python tools/fuzz_gpu.py 200 88 2>&1 | tail -1
  echo "the two sharded phases as virtual ranks, 16-bit / fp32 records (tools/fuzz_shard.py), 300 cases:"; python tools/fuzz_shard.py 300 11 2>&1 | tail -1
  echo "the whole sharded driver, W processes sharing the GPU over gloo (tools/fuzz_driver.py):"
  for a in "2 300 21" "3 200 22"; do python tools/fuzz_driver.py $a 2>&1 | grep "^world\|MISMATCH" | tail -3; done
} > gpurun_out/r05_fuzz.txt 2>&1
cat gpurun_out/r05_fuzz.txt
python tools/rank_shape_probe.py 2>&1 | tail -3 | tee gpurun_out/r05_rank_shape_probe.txt
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/final_bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["whole_job_frac"], d["config"]["output_verified"], d["config"].get("verified_by"), d["cpu_baseline"]["value"])
print(d.get("timing"))
for s in d.get("secondary", []):
    print(s["config"], s.get("ms_per_step", s.get("gpu_seconds_total")), s.get("kernel_ms"), s.get("whole_job_frac"), s.get("output_verified"), s.get("error"))
PY
