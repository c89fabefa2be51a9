#!/bin/bash
# round 5, GPU session L: store policies (non-temporal result stores of the integer kernel; of pass X's indices), on the same
# volume again and again and on two volumes taken in turn (EDT_BENCH_ALTERNATE=1)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
b() {  # b <tag> <cfg> [env...]
  local tag=$1 cfg=$2; shift 2
  env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/r05l_${tag}.json 2> gpurun_out/r05l_${tag}.err
  python - $tag <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r05l_{t}.json"))
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"], "frac32B", d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
except Exception as e:
    print(t, "ERR", e, open(f"gpurun_out/r05l_{t}.err").read()[-800:])
PY
}
L=$PWD/euclidean-distance-transform-3d_amd/lib
for c in cfg2 cfg3; do
  b ${c} $c
  b ${c}_znt $c EDT_HIP_LIB=$L/znt/libedt_hip.so
  b ${c}_xnt $c EDT_HIP_LIB=$L/xnt/libedt_hip.so
  b ${c}_alt $c EDT_BENCH_ALTERNATE=1
  b ${c}_alt_znt $c EDT_BENCH_ALTERNATE=1 EDT_HIP_LIB=$L/znt/libedt_hip.so
  b ${c}_alt_xnt $c EDT_BENCH_ALTERNATE=1 EDT_HIP_LIB=$L/xnt/libedt_hip.so
done
