#!/bin/bash
# round 5, GPU session C: the whole GPU tier on the build with the wide form + the border short cuts, times of every configuration
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r05c_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
b() {  # b <tag> <cfg> [env...]
  local tag=$1 cfg=$2; shift 2
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/r05c_${tag}.json 2> gpurun_out/r05c_${tag}.err
  python - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r05c_{t}.json"))
    print(t, d["ms_per_step"], d["timing"]["batch_ms"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(t, "ERR", e, open(f"gpurun_out/r05c_{t}.err").read()[-800:])
PY
}
b cfg2 cfg2
b cfg1 cfg1
b cfg3 cfg3
b cfg3m cfg3m
b cfg3L cfg3L
b cfg3La cfg3La
b cfg3M cfg3M
b cfg2_again cfg2
FUZZ_Q16=1 python tools/fuzz_gpu.py 120 91 2>&1 | tail -1
python tools/fuzz_gpu.py 150 92 2>&1 | tail -1
