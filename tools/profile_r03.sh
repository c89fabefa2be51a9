#!/bin/bash
# Round-3 profile: rocprofv3 kernel-trace stats + PMC passes (each in its own run) for the headline (cfg2), the dense
# multi-label volume (cfg3) and the large-cell segmentation (cfg3L); profiles/r03_traffic.json from the PMC passes
# (FETCH_SIZE x 2.0, WRITE_SIZE x 1.0: profiles/r02_counter_calibration.txt).  Copy gpurun_out/{prof,pmc}_r03* summaries
# into profiles/ afterwards (tools/collect_profiles.py r03).
./tools/gpu_session.sh prof r03 cfg2 -- pmc r03 cfg2 -- prof r03 cfg3 -- pmc r03 cfg3 -- prof r03 cfg3L -- pmc r03 cfg3L
python tools/traffic_from_pmc.py r03
