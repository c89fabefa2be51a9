#!/bin/bash
# sharded paths with the index form of pass 1: virtual ranks, one-process multi-GPU, NCCL world-1 dry run, per-rank phases
mkdir -p gpurun_out
python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_multi_device.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
python tools/rank_shape_probe.py 2>&1 | tail -4
EDT_HIP_DEBUG_MODE=0x100000 python tools/rank_shape_probe.py 2>&1 | tail -2
