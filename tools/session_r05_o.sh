#!/bin/bash
# round 5, GPU session O: the bit-plane transposer under pass Y (side stream) -- parity and A/B (EDT_HIP_NO_OVERLAP=1)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_gpu_q16.py tests/test_gpu_extras.py -m gpu -x -q 2>&1 | tail -2
b() {  # b <tag> <cfg> [env...]
  local tag=$1 cfg=$2; shift 2
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/r05o_${tag}.json 2> gpurun_out/r05o_${tag}.err
  python - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r05o_{t}.json"))
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(t, "ERR", e, open(f"gpurun_out/r05o_{t}.err").read()[-800:])
PY
}
for c in cfg2 cfg3 cfg3L; do
  b ${c} $c
  b ${c}_serial $c EDT_HIP_NO_OVERLAP=1
done
b cfg2_b cfg2
python tools/fuzz_gpu.py 200 98 2>&1 | tail -1
FUZZ_Q16=1 python tools/fuzz_gpu.py 100 99 2>&1 | tail -1
