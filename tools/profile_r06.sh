#!/bin/bash
# Round-6 profile of the FINAL build (run through gpurun from the repo root): rocprofv3 kernel-trace stats + PMC passes (each in
# its own run) for the headline (cfg2), configs[2] (cfg3), the large-cell segmentations (cfg3L, sw256), configs[3] on one GPU
# (cfg4, 1024^3), configs[4] (cfg5: the voxel-graph transform, tools/vg_probe.py; its sdf leg, tools/sdf_probe.py) and the
# sharded leg as a 1-rank RCCL dry run; profiles/r06_traffic.json from the FETCH_SIZE / WRITE_SIZE passes.
# tools/collect_profiles.py r06 copies the summaries into profiles/.
./tools/gpu_session.sh prof r06 cfg2 -- pmc r06 cfg2 -- prof r06 cfg3 -- pmc r06 cfg3 -- prof r06 cfg3L -- pmc r06 cfg3L
PMC_PASSES=3 ./tools/gpu_session.sh prof r06 sw256 -- pmc r06 sw256
BENCH_SIZE=1024 BENCH_STEPS=5 PMC_PASSES=3 ./tools/gpu_session.sh prof r06 cfg4 -- pmc r06 cfg4
./tools/gpu_session.sh cmd r06 cfg5 python tools/vg_probe.py -- pmccmd r06 cfg5 python tools/vg_probe.py
./tools/gpu_session.sh cmd r06 cfg5_sdf python tools/sdf_probe.py -- pmccmd r06 cfg5_sdf python tools/sdf_probe.py
./tools/gpu_session.sh profshard r06
python tools/traffic_from_pmc.py r06 cfg2 cfg3 cfg3L sw256 cfg4 cfg5 cfg5_sdf
# then, locally: python tools/collect_profiles.py r06
