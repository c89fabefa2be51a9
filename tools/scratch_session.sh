cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_q16.py tests/test_gpu_paths.py tests/test_gpu_extras.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
for c in cfg2 cfg3 cfg3m; do
./tools/gpu_session.sh ab new_${c}_$i $c
./tools/gpu_session.sh ab old_${c}_$i $c EDT_HIP_LIB=euclidean-distance-transform-3d_amd/lib/prev/libedt_hip.so
done
done
./tools/gpu_session.sh bench cfg5 cfg3f
EDT_HIP_LIB=euclidean-distance-transform-3d_amd/lib/prev/libedt_hip.so ./tools/gpu_session.sh bench cfg5
