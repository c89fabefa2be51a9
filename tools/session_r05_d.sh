#!/bin/bash
# round 5, GPU session D: the granule-wise far windows (tests + times on the large-cell segmentations)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_q16.py tests/test_gpu_parity.py tests/test_gpu_paths.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r05d_pytest.txt
b() {  # b <tag> <cfg> [env...]
  local tag=$1 cfg=$2; shift 2
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/r05d_${tag}.json 2> gpurun_out/r05d_${tag}.err
  python - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r05d_{t}.json"))
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(t, "ERR", e, open(f"gpurun_out/r05d_{t}.err").read()[-800:])
PY
}
b cfg3L cfg3L
b cfg3La cfg3La
b cfg3M cfg3M
b cfg3Ma cfg3Ma
b cfg3 cfg3
b cfg3m cfg3m
b cfg2 cfg2
python bench.py --steps 10 --warmup 2 --size 1024 --no-cpu-baseline --no-secondary --config cfg4 > gpurun_out/r05d_cfg4.json 2> gpurun_out/r05d_cfg4.err
python -c "
import json; d = json.load(open('gpurun_out/r05d_cfg4.json')); print('cfg4', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['whole_job_frac'])"
FUZZ_Q16=1 python tools/fuzz_gpu.py 150 93 2>&1 | tail -1
PMC_PASSES=3 ./tools/gpu_session.sh pmc r05d cfg3L
