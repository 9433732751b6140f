#!/bin/bash
# kernel stats + SQ wave-cycle split of the fused encode + top-k call, eight-wave filter pass (arm 0) against the four-wave
# one (arm 128), -DCLT_TUNING library.  Usage (on the GPU box): bash tools/prof_w8.sh [arms]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export CLT_LIB=$R/clt_training_amd/libcltmi355_tuning.so
cd /tmp
for arm in ${1:-0 128}; do
  export CLT_GEMM_DEBUG=$arm
  rm -rf /tmp/ks_$arm /tmp/pmc_$arm
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$arm -o p -- python3 $R/tools/prof_encode.py > /dev/null 2>&1 || exit 1
  echo "== arm $arm kernel stats"; python3 $R/tools/kstats.py /tmp/ks_$arm 9
  timeout -s KILL 120 rocprofv3 --kernel-include-regex "gemm_" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_LDS -d /tmp/pmc_$arm -o s --output-format csv -- python3 $R/tools/prof_encode.py > /dev/null 2>&1 || exit 1
  echo "== arm $arm SQ"; python3 $R/tools/pmc_sq_summary.py $(find /tmp/pmc_$arm -name "*counter_collection.csv" | head -1)
done
