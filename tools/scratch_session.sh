cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FUZZ_DUMP=1
FUZZ_Q16=1 FUZZ_FLAT=1 FUZZ_PAD=1 python tools/fuzz_gpu.py 400 7001 2>&1 | grep -A8 "MISMATCH\|cases\|Traceback\|Error"
FUZZ_Q16=1 FUZZ_FLAT=1 FUZZ_PAD=1 EDT_HIP_DEBUG_MODE=0x80 python tools/fuzz_gpu.py 100 7002 2>&1 | grep -A8 "MISMATCH\|cases\|Traceback\|Error"
FUZZ_Q16=1 FUZZ_PAD=1 python tools/fuzz_gpu.py 300 7003 2>&1 | grep -A8 "MISMATCH\|cases\|Traceback\|Error"
FUZZ_Q16=1 FUZZ_INF=1 FUZZ_PAD=1 python tools/fuzz_gpu.py 150 7004 2>&1 | grep -A8 "MISMATCH\|cases\|Traceback\|Error"
FUZZ_PAD=1 python tools/fuzz_gpu.py 300 7005 2>&1 | grep -A8 "MISMATCH\|cases\|Traceback\|Error"
