#!/bin/bash
# round 5, GPU session J: plain vs non-temporal fill loads of the column kernels (lib/nt = the round's build with -DEDT_Q16_NT_FILL=1)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_q16.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
b() {  # b <tag> <cfg> [env...]
  local tag=$1 cfg=$2; shift 2
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/r05j_${tag}.json 2> gpurun_out/r05j_${tag}.err
  python - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r05j_{t}.json"))
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(t, "ERR", e, open(f"gpurun_out/r05j_{t}.err").read()[-800:])
PY
}
NT=$PWD/euclidean-distance-transform-3d_amd/lib/nt/libedt_hip.so
for c in cfg2 cfg3 cfg3f cfg3L cfg3m; do
  b ${c}_plain $c
  b ${c}_nt $c EDT_HIP_LIB=$NT
done
b cfg2_plain2 cfg2
python bench.py --steps 10 --warmup 2 --size 1024 --no-cpu-baseline --no-secondary --config cfg4 > gpurun_out/r05j_cfg4.json 2> gpurun_out/r05j_cfg4.err
python -c "
import json; d = json.load(open('gpurun_out/r05j_cfg4.json')); print('cfg4 plain', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['whole_job_frac'])"
EDT_HIP_LIB=$NT python bench.py --steps 10 --warmup 2 --size 1024 --no-cpu-baseline --no-secondary --config cfg4 > gpurun_out/r05j_cfg4nt.json 2> gpurun_out/r05j_cfg4nt.err
python -c "
import json; d = json.load(open('gpurun_out/r05j_cfg4nt.json')); print('cfg4 nt', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['whole_job_frac'])"
