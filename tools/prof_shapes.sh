#!/bin/bash
# (on the GPU box) rocprofv3 kernel stats of the fused encode + top-k call at a given shape: tools/prof_shapes.sh N M d k
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/ks_shape
BN=$1 BM=$2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_shape -o p -- python3 $R/tools/bench_fused.py --N $1 --M $2 --d $3 --k $4 > /tmp/ks_shape.log 2>&1 || { tail -5 /tmp/ks_shape.log; exit 1; }
grep encode_topk /tmp/ks_shape.log
python3 $R/tools/kstats.py /tmp/ks_shape 12
