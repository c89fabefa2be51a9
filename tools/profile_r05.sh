#!/bin/bash
# Round-5 profile (run through gpurun from the repo root): rocprofv3 kernel-trace stats + PMC passes (each in its own run)
# for the headline (cfg2), configs[2] (cfg3), the large-cell segmentation (cfg3L), configs[3] on one GPU (cfg4, 1024^3),
# configs[4] (cfg5: the voxel-graph transform, tools/vg_probe.py), and the sharded leg as a 1-rank RCCL
# dry run; profiles/r05_traffic.json from the FETCH_SIZE / WRITE_SIZE passes.  tools/collect_profiles.py r05 copies the
# summaries into profiles/.
./tools/gpu_session.sh prof r05 cfg2 -- pmc r05 cfg2 -- prof r05 cfg3 -- pmc r05 cfg3 -- prof r05 cfg3L -- pmc r05 cfg3L
BENCH_SIZE=1024 BENCH_STEPS=5 PMC_PASSES=3 ./tools/gpu_session.sh prof r05 cfg4 -- pmc r05 cfg4
./tools/gpu_session.sh cmd r05 cfg5 python tools/vg_probe.py -- pmccmd r05 cfg5 python tools/vg_probe.py
./tools/gpu_session.sh profshard r05
python tools/traffic_from_pmc.py r05
# then, locally: python tools/collect_profiles.py r05
