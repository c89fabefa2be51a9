#!/bin/bash
# usage: tools/pmc.sh <tag> [bench args...]  -- PMC passes (each in its own rocprofv3 run, kernel-trace only)
tag=$1; shift
maxpass=${PMC_PASSES:-6}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
            "SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F64" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" \
            "WRITE_SIZE GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  if [ $i -gt $maxpass ]; then break; fi
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_${tag}_$i -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pass $i rc=$?"
done
python tools/pmc_summary.py $tag | tee gpurun_out/pmc_${tag}_summary.txt
