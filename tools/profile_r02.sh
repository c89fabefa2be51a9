#!/bin/bash
# Round-2 profile: rocprofv3 kernel-trace stats + PMC passes (each in its own run) for the headline (cfg2) and the
# dense multi-label volume (cfg3), and the calibration of FETCH_SIZE / WRITE_SIZE on a copy of known size with
# the access widths the kernels use (tools/rowprobe: 4 B/lane buffer loads + stores; float4 copy: 16 B/lane).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in cfg2 cfg3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02_$cfg -o p -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/prof_r02_$cfg.log 2>&1
  echo "stats $cfg rc=$?"
  i=0
  for ctrs in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE" \
              "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
              "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_r02${cfg}_$i -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --config $cfg > gpurun_out/pmc_r02${cfg}_$i.log 2>&1
    echo "pmc $cfg pass $i rc=$?"
  done
  python tools/pmc_summary.py r02$cfg > gpurun_out/pmc_r02${cfg}_summary.txt
done
python tools/traffic_from_pmc.py r02
if [ -z "$SKIP_CAL" ]; then  # (the calibration does not depend on the library: once per round)
i=0
for ctrs in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_r02cal_$i -o p -- tools/rowprobe > gpurun_out/pmc_r02cal_$i.log 2>&1
  echo "cal pass $i rc=$?"
done
python tools/pmc_summary.py r02cal > gpurun_out/pmc_r02cal_summary.txt
cat gpurun_out/pmc_r02cal_summary.txt
fi
python bench.py > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_r02.json"))
print(d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["whole_job_frac"], d["config"]["output_verified"])
for s in d.get("secondary", []): print(s["config"], s.get("ms_per_step"), s.get("kernel_ms"), s.get("whole_job_frac"), s.get("output_verified"))
PY
