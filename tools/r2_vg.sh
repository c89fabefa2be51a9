#!/bin/bash
# voxel-graph session: parity of the native form, cfg5 timing, per-kernel breakdown
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_voxel_graph.py -x -q 2>&1 | tail -5
timeout 120 python tools/vg_probe.py 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_vg4 -o p -- python $GRAFT_REPO_ROOT/tools/vg_probe.py > $GRAFT_REPO_ROOT/gpurun_out/prof_vg4.log 2>&1
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_vg4/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:80], r["Calls"], round(float(r["AverageNs"]) / 1e6, 4), r["MinNs"], r["MaxNs"])
PY
