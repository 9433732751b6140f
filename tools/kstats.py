"""Print the top rows of a rocprofv3 --stats kernel CSV (found under a directory)."""
import csv
import glob
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
fs = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)
if not fs:
    sys.exit(f"no *kernel_stats.csv under {d}")
for r in list(csv.DictReader(open(fs[0])))[:top]:
    print(f'{r["Name"][:90]:90s} {int(r["Calls"]):6d} {float(r["AverageNs"]) / 1e3:10.1f} us {float(r["Percentage"]):6.2f} %')
