#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_paths.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
b() {  # b <tag> <cfg> [env...]
  local tag=$1 cfg=$2; shift 2
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/r05n_${tag}.json 2> gpurun_out/r05n_${tag}.err
  python - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r05n_{t}.json"))
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(t, "ERR", e, open(f"gpurun_out/r05n_{t}.err").read()[-800:])
PY
}
b cfg3f cfg3f
b cfg3f_q16off cfg3 EDT_HIP_DEBUG_MODE=0x8000000
b cfg2 cfg2
b cfg3f_2 cfg3f
