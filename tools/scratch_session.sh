cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_q16.py tests/test_gpu_paths.py tests/test_gpu_parity.py tests/test_gpu_extras.py -m gpu -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "sweep or cfg3 or sdf" 2>&1 | tail -3
FUZZ_Q16=1 python tools/fuzz_gpu.py 200 911 2>&1 | grep "MISMATCH\|cases"
python tools/fuzz_gpu.py 400 912 2>&1 | grep "MISMATCH\|cases"
python tools/fuzz_shard.py 100 913 2>&1 | grep "MISMATCH\|cases"
for c in onesF sw256 cfg2 cfg3; do
./tools/gpu_session.sh ab ${c}_a $c
./tools/gpu_session.sh ab ${c}_b $c EDT_HIP_DEBUG_MODE=0x80
done
