"""Summarise one `rocprofv3 --pmc SQ_* --output-format csv` pass per kernel: counter means per launch and the
wave-cycle split (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        n[k] += 1
for k, v in acc.items():
    wc = v["SQ_WAVE_CYCLES"] or 1.0
    print(k, "launches", n[k], {c: round(x / max(1, n[k])) for c, x in v.items()},
          "wait_any/wave %.3f wait_inst/wave %.3f active/wave %.3f" % (v["SQ_WAIT_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc,
                                                                      v["SQ_ACTIVE_INST_ANY"] / wc))
