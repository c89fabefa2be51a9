cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_q16.py -m gpu -x -q -k "without_structure or nothing_but_inf" 2>&1 | tail -3
python -m pytest tests/test_gpu_extras.py tests/test_gpu_reference_verbatim.py tests/test_gpu_voxel_graph.py tests/test_gpu_multiproc.py tests/test_gpu_multi_device.py -m gpu -x -q 2>&1 | tail -3
python tools/fuzz_shard.py 150 9201 2>&1 | grep "MISMATCH\|cases"
python tools/fuzz_driver.py 2 100 9202 2>&1 | grep "MISMATCH\|cases"
