#!/bin/bash
# pass X ring kernel (edt_rowring.hip): parity, then cfg2 / cfg3 / cfg4-slab step and per-pass times for ring depths 2, 3, 4
# against the register-pipelined kernel (EDT_HIP_DEBUG_MODE=0x400000)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_rowring.py -m gpu -x -q 2>&1 | tail -5
one() {  # one <label> <cfg> [size]
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --config $2 ${3:+--size $3} > gpurun_out/ring_$1.json 2> gpurun_out/ring_$1.err
  python - $1 <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/ring_{t}.json"))
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["output_verified"])
except Exception as e:
    print(t, "ERR", e, open(f"gpurun_out/ring_{t}.err").read()[-400:])
PY
}
EDT_HIP_DEBUG_MODE=0x400000 one regs_cfg2 cfg2
for d in 2 3 4; do EDT_ROW_RING=$d one ring${d}_cfg2 cfg2; done
EDT_HIP_DEBUG_MODE=0x400000 one regs_cfg3 cfg3
EDT_ROW_RING=3 one ring3_cfg3 cfg3
EDT_HIP_DEBUG_MODE=0x400000 one regs_cfg4 cfg4 1024
for d in 2 3; do EDT_ROW_RING=$d one ring${d}_cfg4 cfg4 1024; done
