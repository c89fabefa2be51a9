#!/bin/bash
# Round-4 profile (run through gpurun from the repo root): rocprofv3 kernel-trace stats + PMC passes (each in its own run)
# for the headline (cfg2), configs[2] (cfg3), the large-cell segmentation (cfg3L), configs[3] on one GPU (cfg4, 1024^3),
# configs[4] (cfg5: the voxel-graph transform, tools/vg_probe.py), the 4096-row shape and the sharded leg as a 1-rank RCCL
# dry run; profiles/r04_traffic.json from the FETCH_SIZE / WRITE_SIZE passes.  tools/collect_profiles.py r04 copies the
# summaries into profiles/.
./tools/gpu_session.sh prof r04 cfg2 -- pmc r04 cfg2 -- prof r04 cfg3 -- pmc r04 cfg3 -- prof r04 cfg3L -- pmc r04 cfg3L
BENCH_SIZE=1024 BENCH_STEPS=5 PMC_PASSES=3 ./tools/gpu_session.sh prof r04 cfg4 -- pmc r04 cfg4
./tools/gpu_session.sh cmd r04 cfg5 python tools/vg_probe.py -- pmccmd r04 cfg5 python tools/vg_probe.py
./tools/gpu_session.sh cmd r04 4096x4096x8 python tools/shape_times.py 4096 4096 8 -- pmccmd r04 4096x4096x8 python tools/shape_times.py 4096 4096 8
./tools/gpu_session.sh profshard r04
python tools/traffic_from_pmc.py r04
# then, locally: python tools/collect_profiles.py r04
