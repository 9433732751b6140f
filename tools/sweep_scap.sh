#!/bin/bash
# (on the GPU box) the fused encode + top-k call at C2 for several caps of the sample stride S (-DCLT_TUNING library)
R=$GRAFT_REPO_ROOT
export CLT_LIB=$R/clt_training_amd/libcltmi355_tuning.so
for rep in 1 2; do
for s in ${@:-16 24 32 48}; do
  echo -n "S cap $s: "; CLT_FUSED_SCAP=$s timeout -k 10 120 python3 -u $R/tools/ab_w8.py --arms 0 --rounds 4 2>/dev/null | grep -E "median|fallback" | tr '\n' ' '; echo
done
done
