cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, numpy as np
sys.path[:0] = ["euclidean-distance-transform-3d_amd", "."]
import edt
from oracle import harness
d = np.load("gpurun_out_in/fuzz_fail_245.npz") if False else None
PY
python -m pytest tests/test_gpu_q16.py -m gpu -x -q -k "refused_tile or nothing_but_inf" 2>&1 | tail -3
FUZZ_Q16=1 python tools/fuzz_gpu.py 600 6103 2>&1 | grep "MISMATCH\|cases"
FUZZ_Q16=1 python tools/fuzz_gpu.py 600 6203 2>&1 | grep "MISMATCH\|cases"
FUZZ_Q16=1 python tools/fuzz_gpu.py 600 6303 2>&1 | grep "MISMATCH\|cases"
