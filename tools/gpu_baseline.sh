#!/bin/bash
# one-shot GPU session: parity tests, bench (cfg2 + cfg3), calibration, rocprof kernel stats
mkdir -p gpurun_out
rocminfo | grep -c gfx950 > gpurun_out/gfx950_agents.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err
python bench.py --steps 20 --warmup 3 --config cfg3 --no-cpu-baseline > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err
python bench.py --steps 20 --warmup 3 --config cfg3m --no-cpu-baseline > gpurun_out/bench_cfg3m.json 2> gpurun_out/bench_cfg3m.err
python tools/membench.py > gpurun_out/membench.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_base -o p -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_base.log 2>&1
cat gpurun_out/bench_cfg2.json gpurun_out/bench_cfg3.json gpurun_out/bench_cfg3m.json gpurun_out/membench.log
