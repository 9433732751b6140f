"""Print the multi-rank keys of a bench.py JSON line: python tools/show_coll.py file"""
import json
import sys

j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
for k in ("value", "n_gpus", "simulated", "rccl", "overlap"):
    print(k, j.get(k))
print(j["config"]["parallelism"])
for k, v in (j.get("collectives") or {}).items():
    print("  ", k, v)
