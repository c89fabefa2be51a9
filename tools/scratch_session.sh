cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FUZZ_DUMP=1 FUZZ_Q16=1 FUZZ_INF=1 python tools/fuzz_gpu.py 500 7001 2>&1 | grep -A9 "MISMATCH\|cases"
FUZZ_DUMP=1 FUZZ_Q16=1 FUZZ_INF=1 EDT_HIP_DEBUG_MODE=0x40000000 python tools/fuzz_gpu.py 200 7002 2>&1 | grep -A9 "MISMATCH\|cases"
FUZZ_DUMP=1 FUZZ_Q16=1 FUZZ_INF=1 EDT_HIP_DEBUG_MODE=0x10000000 python tools/fuzz_gpu.py 200 7003 2>&1 | grep -A9 "MISMATCH\|cases"
