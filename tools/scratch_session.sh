cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_q16.py tests/test_gpu_parity.py tests/test_gpu_paths.py -m gpu -x -q 2>&1 | tail -3
FUZZ_Q16=1 FUZZ_FLAT=1 FUZZ_PAD=1 python tools/fuzz_gpu.py 200 7301 2>&1 | grep -A8 "MISMATCH\|cases\|Traceback\|Error"
FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x10000000 python tools/fuzz_gpu.py 200 7302 2>&1 | grep -A8 "MISMATCH\|cases\|Traceback\|Error"
FUZZ_Q16=1 EDT_HIP_DEBUG_MODE=0x100000 python tools/fuzz_gpu.py 200 7303 2>&1 | grep -A8 "MISMATCH\|cases\|Traceback\|Error"
FUZZ_Q16=1 python tools/fuzz_gpu.py 300 7304 2>&1 | grep -A8 "MISMATCH\|cases\|Traceback\|Error"
for i in 1 2 3; do
for c in cfg2 cfg3; do
./tools/gpu_session.sh ab new_${c}_$i $c
./tools/gpu_session.sh ab old_${c}_$i $c EDT_HIP_LIB=euclidean-distance-transform-3d_amd/lib/prev/libedt_hip.so
done
done
