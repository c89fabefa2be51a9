cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_q16.py tests/test_gpu_paths.py tests/test_gpu_parity.py tests/test_gpu_index_form.py -m gpu -x -q 2>&1 | tail -3
FUZZ_Q16=1 python tools/fuzz_gpu.py 250 901 2>&1 | grep "MISMATCH\|cases"
python tools/fuzz_gpu.py 300 902 2>&1 | grep "MISMATCH\|cases"
python tools/fuzz_shard.py 100 903 2>&1 | grep "MISMATCH\|cases"
for c in cfg2 cfg3 cfg3m cfg3M cfg3Ma cfg3L cfg3La sw26 sphere250; do
./tools/gpu_session.sh ab ${c}_small $c
./tools/gpu_session.sh ab ${c}_nosmall $c EDT_HIP_DEBUG_MODE=0x80
done
