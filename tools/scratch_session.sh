cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_q16.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
FUZZ_Q16=1 FUZZ_FLAT=1 FUZZ_PAD=1 python tools/fuzz_gpu.py 300 7201 2>&1 | grep -A8 "MISMATCH\|cases\|Traceback\|Error"
FUZZ_MAX_AXIS=1400 FUZZ_FLAT=1 python tools/fuzz_gpu.py 300 7202 2>&1 | grep -A8 "MISMATCH\|cases\|Traceback\|Error"
for i in 1 2; do
for c in cfg2 cfg1 cfg3; do
./tools/gpu_session.sh ab new_${c}_$i $c
./tools/gpu_session.sh ab old_${c}_$i $c EDT_HIP_LIB=euclidean-distance-transform-3d_amd/lib/prev/libedt_hip.so
done
done
