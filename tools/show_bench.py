"""Print the per-kernel table of bench.py JSON lines: python tools/show_bench.py gpurun_out/x.log ..."""
import json
import sys

for f in sys.argv[1:]:
    for line in open(f):
        if line.startswith("{"):
            j = json.loads(line)
            print(f, j["value"], "tok/s", j["ms_per_step"], "ms host", j["host_enqueue_wall_ms_per_step"], j["config"]["parallelism"])
            tot = 0.0
            for k, v in j["kernels"].items():
                tot += v["ms_per_step"]
                print(f"    {k:28s} {v['launches_per_step']:6.1f} x {v['mean_ms']:8.4f} = {v['ms_per_step']:7.3f} ms  {v.get('achieved')} {v.get('unit')}")
            print(f"    kernels total {tot:.2f} ms; other {j['ms_per_step'] - tot:.2f} ms")
