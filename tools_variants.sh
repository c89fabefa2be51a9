#!/bin/bash
for t in "" _A _B _C; do
  echo "variant '$t'"
  EDT_HIP_LIB=$GRAFT_REPO_ROOT/euclidean-distance-transform-3d_amd/lib/libedt_hip$t.so python tools_membench.py 2>&1 | tail -2
done
