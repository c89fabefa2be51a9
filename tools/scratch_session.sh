cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_voxel_graph.py -m gpu -x -q 2>&1 | tail -3
FUZZ_VG=1 python tools/fuzz_gpu.py 300 7101 2>&1 | grep "MISMATCH\|cases\|Traceback\|Error"
for i in 1 2; do
python bench.py --steps 20 --secondary cfg5 --no-cpu-baseline > gpurun_out/vg_new_$i.json 2>gpurun_out/vg.err
EDT_HIP_LIB=euclidean-distance-transform-3d_amd/lib/prev/libedt_hip.so python bench.py --steps 20 --secondary cfg5 --no-cpu-baseline > gpurun_out/vg_old_$i.json 2>gpurun_out/vg.err
done
python - <<'PY'
import json
for n in ("new_1","old_1","new_2","old_2"):
    d=json.load(open(f"gpurun_out/vg_{n}.json"))
    for s in d["secondary"]:
        print(n, s["config"], s.get("ms_per_step"), s.get("kernel_ms"), s.get("output_verified"), s.get("error"))
PY
