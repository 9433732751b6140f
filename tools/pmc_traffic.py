"""Per-launch HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE in one run, WRITE_SIZE in another; they do not
fit one pass together with nothing else: MI355X_MICROARCH.md, HBM / rocprofv3 section):

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f --output-format csv -- python3 bench.py --steps 1 --warmup 1 --layers 2 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --layers 2 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv profiles/rNN_pmc_traffic.json

Counters are KiB; hbm_bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): gfx950's FETCH_SIZE tallies 128-byte requests at
64 bytes (the guide's correction).  Medians over the launches of each kernel (the first step's launches of the fused
encoder include cold caches; medians ignore them)."""
import csv
import json
import statistics
import sys

csv.field_size_limit(1 << 30)

# kernel-name substring -> the C-ABI entry point (or stage of it) the launch belongs to
RULES = [
    ("rows_lds_multi_kernel", "Lb1EEE", "clt_decode_bwd_rows_adam_multi"),
    ("rows_lds_multi_kernel", "Lb0EEE", "clt_spt_dense_adam_multi"),
    ("rows_lds_kernel", "Lb1ELb1EEE", "clt_decode_bwd_rows_adam"),  # <T, CH, U, BWD, ADAM>
    ("rows_lds_kernel", "Lb0ELb1EEE", "clt_spt_dense_adam"),
    ("decode_fwd_multi_lds_kernel", "", "clt_decode_fwd_multi"),
    ("decode_fwd_multi_kernel", "", "clt_decode_fwd_multi(register gather)"),
    ("decode_bwd_acts_multi_kernel", "", "clt_decode_bwd_acts_multi"),
    ("decode_fwd_kernel", "", "clt_decode_fwd"),
    ("decode_bwd_acts_kernel", "", "clt_decode_bwd_acts"),
    ("spt_dense_kernel", "", "spt_dense_kernel (register variant)"),
    ("fused_select_stream_kernel", "", "fused_select_stream_kernel"),
    ("fused_select_kernel", "", "fused_select_kernel"),
    ("fused_bucket_kernel", "", "fused_bucket_kernel"),
    ("gemm_w8_filter_kernel", "", "gemm_w8_filter_kernel"),
    ("sample_topk_kernel", "", "clt_topk_rows(sample)"),  # (round 3: the register-resident sample top-k + threshold)
    ("topk_rows_kernel", "", "clt_topk_rows(sample)"),
    ("adam_kernel", "", "clt_adam_step"),
    ("pack_x_kernel", "", "pack_x_kernel"),
    ("gemm_fast", "", "gemm_fast_kernel"),  # gemm_fast_kernel (32x32x16) and gemm_fast16_kernel (16x16x32)
]


JOBS = {}  # tag -> jobs per launch of a batched kernel (grid.y), when the CSV carries the grid


def load(path, counter):
    per = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            for sub, sub2, tag in RULES:
                if sub in name and sub2 in name:
                    if tag == "gemm_fast_kernel":  # split the GEMM launches by what they do: grid / LDS size tell
                        tag = f"gemm_fast_kernel<lds={row['LDS_Block_Size']}>"
                    per.setdefault(tag, []).append(float(row["Counter_Value"]))
                    if "multi" in tag and row.get("Grid_Size_Y") and row.get("Workgroup_Size_Y"):
                        JOBS[tag] = max(1, int(row["Grid_Size_Y"]) // max(1, int(row["Workgroup_Size_Y"])))
                    break
    return per


def main():
    fpath, wpath, out = sys.argv[1:4]
    fetch, write = load(fpath, "FETCH_SIZE"), load(wpath, "WRITE_SIZE")
    kernels = {}
    for tag in sorted(set(fetch) | set(write)):
        fv = [v for v in fetch.get(tag, []) if v > 64]  # drop the gated no-op launches
        wv = write.get(tag, [])
        if not fv:
            continue
        f_med = statistics.median(fv)
        w_med = statistics.median([v for v in wv if v > 0] or [0.0])
        kernels[tag] = {"launches": len(fv), "FETCH_SIZE_KiB": f_med, "WRITE_SIZE_KiB": w_med,
                        "hbm_bytes": int(1024 * (2 * f_med + w_med))}
        if tag in JOBS:  # batched launch: the probe's launches hold JOBS[tag] (source, target) pairs each
            kernels[tag]["jobs_per_launch"] = JOBS[tag]
            kernels[tag]["hbm_bytes_per_job"] = kernels[tag]["hbm_bytes"] // JOBS[tag]
    # the fused encoder call = x pack + sample GEMM + sample top-k + filter GEMM + bucket + merge (its gated fallback
    # launches fetch nothing)
    stages = [k for k in kernels if k.startswith("gemm_fast_kernel") or k in (
        "clt_topk_rows(sample)", "fused_select_kernel", "fused_select_stream_kernel", "fused_bucket_kernel",
        "gemm_w8_filter_kernel", "pack_x_kernel")]
    if stages:
        kernels["clt_encode_topk_fwd"] = {"stages": stages, "hbm_bytes": sum(kernels[k]["hbm_bytes"] for k in stages)}
    doc = {"command": "rocprofv3 --pmc FETCH_SIZE (then, in a second run, --pmc WRITE_SIZE) --output-format csv -- python3 "
                      "bench.py --steps 1 --warmup 1 --layers 2 --no-cpu-baseline; tools/pmc_traffic.py",
           "note": "per-launch medians at BASELINE configs[1] layer shapes (N=8192, d=768, M=98304, k=32, bf16); counters are "
                   "KiB; hbm_bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE): gfx950 FETCH_SIZE tallies 128-B requests at 64 B "
                   "(MI355X_MICROARCH.md, HBM section)",
           "kernels": kernels}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in kernels.items():
        print(f"{k:50s} {v.get('launches', 0):4d} launches  {v['hbm_bytes'] / 1e6:10.1f} MB")


if __name__ == "__main__":
    main()
