#!/bin/bash
# round 5, GPU session E: persistent workgroups of the integer column kernel (EDT_Q16_PERSIST = workgroups per CU)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_q16.py -m gpu -x -q 2>&1 | tail -2
EDT_Q16_PERSIST=4 python -m pytest tests/test_gpu_q16.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
b() {  # b <tag> <cfg> [env...]
  local tag=$1 cfg=$2; shift 2
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/r05e_${tag}.json 2> gpurun_out/r05e_${tag}.err
  python - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r05e_{t}.json"))
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(t, "ERR", e, open(f"gpurun_out/r05e_{t}.err").read()[-800:])
PY
}
for c in cfg2 cfg3 cfg3L; do
  b ${c}_p0 $c
  b ${c}_p4 $c EDT_Q16_PERSIST=4
  b ${c}_p8 $c EDT_Q16_PERSIST=8
  b ${c}_p16 $c EDT_Q16_PERSIST=16
done
