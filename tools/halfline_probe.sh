#!/bin/bash
# tools/halfline_probe (times) + its FETCH_SIZE per kernel under rocprofv3 -> gpurun_out/r05_halfline_probe.txt
# (through gpurun from the repo root; build first: hipcc --offload-arch=gfx950 -O3 -o tools/halfline_probe tools/halfline_probe.hip)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
./tools/halfline_probe | tee gpurun_out/r05_halfline_probe.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_halfline -o p -- ./tools/halfline_probe > /dev/null 2>&1
python - <<'PY' | tee -a gpurun_out/r05_halfline_probe.txt
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_halfline/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"FETCH_SIZE per launch (KiB, raw counter): {sum(v)/len(v):12.1f}   x2 = {2*sum(v)/len(v)*1024/1e6:8.1f} MB   {k}")
print("the array is 268.4 MB")
PY
