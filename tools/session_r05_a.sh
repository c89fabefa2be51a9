#!/bin/bash
# round 5, GPU session A: the wide form of the integer kernel (tests), the headline with / without it, pass-X mapping probes
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_q16.py tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r05a_pytest.txt
b() {  # b <tag> <cfg> [env...]
  local tag=$1 cfg=$2; shift 2
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/r05a_${tag}.json 2> gpurun_out/r05a_${tag}.err
  python - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r05a_{t}.json"))
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(t, "ERR", e, open(f"gpurun_out/r05a_{t}.err").read()[-800:])
PY
}
b cfg2 cfg2
b cfg2_nowide cfg2 EDT_HIP_DEBUG_MODE=0x20000000
b cfg1 cfg1
b cfg3 cfg3
b cfg3L cfg3L
b cfg2_again cfg2
./tools/rowmap_probe > gpurun_out/r05_rowmap_probe.txt 2>&1; cat gpurun_out/r05_rowmap_probe.txt
./tools/gpu_session.sh prof r05a cfg2
