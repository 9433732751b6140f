#!/bin/bash
# (on the GPU box) the forward fan-out, one process per library, same box: tools/ab_fanout.sh lib1 lib2 ... [-- bench_fanout flags]
R=$GRAFT_REPO_ROOT
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; shift
for rep in 1 2; do
for l in "${libs[@]}"; do
  echo -n "$l: "; CLT_LIB=$R/clt_training_amd/$l timeout -k 10 120 python3 -u $R/tools/bench_fanout.py "$@" 2>/dev/null | grep "us"
done
done
