#!/bin/bash
# round-2 session 6: full GPU suite (incl. full-size parity), bench line with secondaries
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/s6_pytest.log
python bench.py > gpurun_out/s6_bench.json 2> gpurun_out/s6_bench.err; tail -3 gpurun_out/s6_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/s6_bench.json"))
print(d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
for s in d.get("secondary", []): print(s)
print(d.get("cpu_baseline"))
PY
