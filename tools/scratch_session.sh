cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_q16.py tests/test_gpu_paths.py tests/test_gpu_index_form.py tests/test_gpu_extras.py -m gpu -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "cfg3 or cfg4 or sweep" 2>&1 | tail -3
FUZZ_Q16=1 python tools/fuzz_gpu.py 200 921 2>&1 | grep "MISMATCH\|cases"
python tools/fuzz_gpu.py 400 922 2>&1 | grep "MISMATCH\|cases"
for i in 1 2; do for c in cfg2 cfg3; do
./tools/gpu_session.sh ab ${c}_a$i $c
./tools/gpu_session.sh ab ${c}_b$i $c EDT_HIP_DEBUG_MODE=0x400
done; done
