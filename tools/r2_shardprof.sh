#!/bin/bash
# rocprofv3 kernel stats of the N > 1 bench leg as a 1-rank RCCL dry run (4 z-chunks): the XF + SC column kernel
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
EDT_SHARD_CHUNKS=4 EDT_BENCH_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02_shard -o p -- \
  python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/prof_r02_shard.log 2>&1
grep -a "^{" $GRAFT_REPO_ROOT/gpurun_out/prof_r02_shard.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('output_verified'), d.get('scaling'))"
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_r02_shard/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"]) / 1e6, 4))
PY
