cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_extras.py tests/test_gpu_parity.py tests/test_gpu_reference_suite.py -m gpu -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "sdf" 2>&1 | tail -3
python -m pytest tests/test_gpu_reference_verbatim.py -m gpu -x -q 2>&1 | tail -3
for m in 0 0x400; do
EDT_HIP_DEBUG_MODE=$m python bench.py --steps 40 --secondary cfg5_sdf > gpurun_out/sdf_$m.json 2> gpurun_out/sdf_$m.err
python - $m <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/sdf_{sys.argv[1]}.json"))
for s in d.get("secondary", []): print(sys.argv[1], s["config"], s.get("ms_per_step"), s.get("two_transform_ms"), s.get("output_verified"))
PY
done
