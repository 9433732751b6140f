#!/bin/bash
# (on the GPU box) whole fused encode + top-k call, one process per library, same box: tools/ab_libs.sh lib1 lib2 ...
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for l in "$@"; do
  echo -n "$l: "; CLT_LIB=$R/clt_training_amd/$l timeout -k 10 120 python3 -u $R/tools/ab_w8.py --arms 0 --rounds 4 2>/dev/null | grep median
done
done
