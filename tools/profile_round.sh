#!/bin/bash
# Round profile: rocprofv3 kernel-trace stats of the default bench command + PMC traffic passes.
# Everything lands in gpurun_out/; copy the summaries into profiles/ afterwards.
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o p -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
echo "stats rc=$?"
for ctrs in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_${tag}_$i -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
python tools/pmc_summary.py $tag | tee gpurun_out/pmc_${tag}_summary.txt
python bench.py --steps 20 --warmup 3 | tee gpurun_out/bench_$tag.json
