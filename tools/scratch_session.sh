cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_voxel_graph.py -m gpu -x -q 2>&1 | tail -3
FUZZ_VG=1 python tools/fuzz_gpu.py 300 801 2>&1 | grep "MISMATCH\|cases"
for v in 1 0 1 0; do
EDT_HIP_VG_SIDE_STREAM=$v EDT_BENCH_VERIFY=0 python bench.py --steps 40 --secondary cfg5 --no-cpu-baseline > gpurun_out/vg_$v.json 2> gpurun_out/vg_$v.err
python - $v <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/vg_{sys.argv[1]}.json"))
for s in d.get("secondary", []): print("side", sys.argv[1], s["config"], s.get("ms_per_step"))
PY
done
