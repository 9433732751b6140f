#!/bin/bash
# (on the GPU box) MFMA-pipe occupancy of the encoder kernels: one --pmc pass over tools/bench_fused.py at the C2 layer shape.
# SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES (both summed over the CUs) = share of the CU-busy cycles with the MFMA pipe busy.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pmc_mfma
timeout -s KILL 120 rocprofv3 --kernel-include-regex "gemm_w8|gemm_fast16" --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU -d /tmp/pmc_mfma -o m --output-format csv -- python3 $R/tools/bench_fused.py --N 8192 --M 98304 --d 768 --k 32 > /tmp/pmc_mfma.log 2>&1 || { tail -5 /tmp/pmc_mfma.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
csv.field_size_limit(1 << 30)
f = glob.glob('/tmp/pmc_mfma/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in acc.items():
    med = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
    print(k, len(next(iter(c.values()))), 'launches', {n: int(v) for n, v in med.items()})
    if med.get('SQ_BUSY_CU_CYCLES'):
        print('   MFMA busy / CU busy = %.3f' % (med.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / med['SQ_BUSY_CU_CYCLES']))
PY
