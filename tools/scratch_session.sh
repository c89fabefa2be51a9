cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -x -q 2>&1 | tail -8) 2>&1 | tee gpurun_out/r06_gpu_tier.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r06_gpu_tier.txt
python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench.err
tail -2 gpurun_out/r06_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_line.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["output_verified"], d["cpu_baseline"]["value"], d["roofline"]["traffic"]["total"])
for s in d.get("secondary", []):
    print(s["config"], s.get("ms_per_step", s.get("gpu_seconds_total")), s.get("whole_job_frac"), s.get("output_verified"), s.get("error"))
PY
