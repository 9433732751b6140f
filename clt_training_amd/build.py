"""Build libcltmi355.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libcltmi355.so"
SOURCES = ["common.hip", "decode.hip", "segments.hip", "decode_rows.hip", "topk.hip", "encoder_gemm.hip", "encoder_gemm_fast.hip", "encoder_gemm_fast16.hip", "encoder_gemm_w8.hip", "fvu.hip", "adam.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True, tuning: bool = False) -> Path:
    """tuning=True: -DCLT_TUNING (the kernels' timing / A-B switches read the environment) into libcltmi355_tuning.so,
    for tools/ only (select it with CLT_LIB=...); the shipped library never reads the environment."""
    objdir = HERE / ("build_tuning" if tuning else "build")
    objdir.mkdir(exist_ok=True)
    flags = FLAGS + (["-DCLT_TUNING"] if tuning else [])
    lib = HERE / "libcltmi355_tuning.so" if tuning else LIB
    headers = [*CSRC.glob("*.h"), HERE.parent / "include" / "clt_mi355.h"]

    def compile_one(src):
        obj = objdir / (Path(src).stem + ".o")
        if force or _stale(obj, [CSRC / src, *headers]):
            cmd = [HIPCC, *flags, "-c", str(CSRC / src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, tuning="--tuning" in sys.argv)
