#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_q16.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
b() {  # b <tag> <cfg> [env...]
  local tag=$1 cfg=$2; shift 2
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/r05g_${tag}.json 2> gpurun_out/r05g_${tag}.err
  python - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r05g_{t}.json"))
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(t, "ERR", e, open(f"gpurun_out/r05g_{t}.err").read()[-800:])
PY
}
b cfg2 cfg2
b cfg2_fullwide cfg2 EDT_HIP_DEBUG_MODE=0x40000000
b cfg2_nowide cfg2 EDT_HIP_DEBUG_MODE=0x20000000
b cfg1 cfg1
b cfg2_again cfg2
FUZZ_Q16=1 python tools/fuzz_gpu.py 100 95 2>&1 | tail -1
