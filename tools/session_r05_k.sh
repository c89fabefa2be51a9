#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
b() {  # b <tag> <cfg> [env...]
  local tag=$1 cfg=$2; shift 2
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/r05k_${tag}.json 2> gpurun_out/r05k_${tag}.err
  python - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r05k_{t}.json"))
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(t, "ERR", e, open(f"gpurun_out/r05k_{t}.err").read()[-800:])
PY
}
NT=$PWD/euclidean-distance-transform-3d_amd/lib/nt/libedt_hip.so
b cfg2 cfg2
b cfg2_nt cfg2 EDT_HIP_LIB=$NT
b cfg3 cfg3
b cfg3M cfg3M
b cfg2_b cfg2
for v in "" "EDT_HIP_LIB=$NT"; do
  env $v python bench.py --steps 10 --warmup 2 --size 1024 --no-cpu-baseline --no-secondary --config cfg4 > gpurun_out/r05k_cfg4.json 2> gpurun_out/r05k_cfg4.err
  python -c "
import json; d = json.load(open('gpurun_out/r05k_cfg4.json')); print('cfg4 [$v]', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['whole_job_frac'])"
done
