#!/bin/bash
# Round 6: an EXTENDED fuzz of the final build (beyond the closing session's 6 600 cases): other seeds, four times the cases ->
# profiles/r06_fuzz_extended.txt.  Stops at nothing: every mismatch is printed with its triage (FUZZ_DUMP=1).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
export FUZZ_DUMP=1
f() { echo "$1:"; shift; env "$@" 2>&1 | grep -A9 "MISMATCH\|cases\|Traceback\|Error" ; }
{
  echo "# tools/fuzz_*.py on the final build of round 6, extended run (GPU vs oracle, bit for bit)"
  f "general, 6000 cases" python tools/fuzz_gpu.py 6000 8101
  f "general, axes up to 2100, 1000 cases" FUZZ_MAX_AXIS=2100 python tools/fuzz_gpu.py 1000 8102
  f "integer kernel's shapes (FUZZ_Q16=1), 2500 cases" FUZZ_Q16=1 python tools/fuzz_gpu.py 2500 8103
  f "the same shapes, volumes of +inf (FUZZ_INF=1), 2000 cases" FUZZ_Q16=1 FUZZ_INF=1 python tools/fuzz_gpu.py 2000 8104
  f "the same, 0x40000000, 600 cases" FUZZ_Q16=1 FUZZ_INF=1 EDT_HIP_DEBUG_MODE=0x40000000 python tools/fuzz_gpu.py 600 8105
  f "the same, 0x10000000, 600 cases" FUZZ_Q16=1 FUZZ_INF=1 EDT_HIP_DEBUG_MODE=0x10000000 python tools/fuzz_gpu.py 600 8106
  f "volumes of slabs and boxes (FUZZ_FLAT=1), a random pitch of the index buffer per case (FUZZ_PAD=1), 2000 cases" FUZZ_Q16=1 FUZZ_FLAT=1 FUZZ_PAD=1 python tools/fuzz_gpu.py 2000 8112
  f "the same, 0x80 (no short cuts for whole tiles), 500 cases" FUZZ_Q16=1 FUZZ_FLAT=1 FUZZ_PAD=1 EDT_HIP_DEBUG_MODE=0x80 python tools/fuzz_gpu.py 500 8113
  f "integer kernel's shapes, a random pitch per case, 1500 cases" FUZZ_Q16=1 FUZZ_PAD=1 python tools/fuzz_gpu.py 1500 8114
  f "general shapes with slabs and boxes, axes up to 1400, 1000 cases" FUZZ_MAX_AXIS=1400 FUZZ_FLAT=1 FUZZ_PAD=1 python tools/fuzz_gpu.py 1000 8115
  f "integer kernel's shapes, 0x400 (foreground planes kept), 600 cases" FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x400 python tools/fuzz_gpu.py 600 8107
  f "voxel-graph transform (FUZZ_VG=1), 3000 cases" FUZZ_VG=1 python tools/fuzz_gpu.py 3000 8108
  f "the two sharded phases as virtual ranks (tools/fuzz_shard.py), 1500 cases" python tools/fuzz_shard.py 1500 8109
  echo "the whole sharded driver (tools/fuzz_driver.py):"
  python tools/fuzz_driver.py 2 1000 8110 2>&1 | grep "MISMATCH\|cases\|Traceback"
  python tools/fuzz_driver.py 3 600 8111 2>&1 | grep "MISMATCH\|cases\|Traceback"
} > gpurun_out/r06_fuzz_extended.txt 2>&1
cat gpurun_out/r06_fuzz_extended.txt
