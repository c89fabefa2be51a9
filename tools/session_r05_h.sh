#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
PMC_PASSES=4 ./tools/gpu_session.sh pmc r05hs cfg2 > /dev/null 2>&1
EDT_HIP_DEBUG_MODE=0x40000000 PMC_PASSES=4 ./tools/gpu_session.sh pmc r05hf cfg2 > /dev/null 2>&1
for m in s f; do echo "== $m"; grep -A20 "k_column_pass_q16<true, 1, true" gpurun_out/pmc_r05h${m}cfg2_summary.txt | head -22; done
