#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_q16.py tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -x -q 2>&1 | tail -2
b() {  # b <tag> <cfg> [env...]
  local tag=$1 cfg=$2; shift 2
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/r05m_${tag}.json 2> gpurun_out/r05m_${tag}.err
  python - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r05m_{t}.json"))
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(t, "ERR", e, open(f"gpurun_out/r05m_{t}.err").read()[-800:])
PY
}
for c in cfg2 cfg3 cfg3f cfg3m cfg3L cfg3M; do b $c $c; done
b cfg2_alt cfg2 EDT_BENCH_ALTERNATE=1
python bench.py --steps 10 --warmup 2 --size 1024 --no-cpu-baseline --no-secondary --config cfg4 > gpurun_out/r05m_cfg4.json 2> gpurun_out/r05m_cfg4.err
python -c "
import json; d = json.load(open('gpurun_out/r05m_cfg4.json')); print('cfg4', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['whole_job_frac'])"
