#!/bin/bash
# usage: tools_pmc.sh <tag> [bench args...]  -- PMC passes (each in its own rocprofv3 run, kernel-trace only)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" \
            "WRITE_SIZE GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_${tag}_$i -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pass $i rc=$?"
done
find gpurun_out -name "*counter_collection.csv" | head
