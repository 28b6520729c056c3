"""Build the gfx950 kernel libraries (libmudg_hip*.so), in-tree.

`python -m mudg_amd.build` compiles every .hip under mudg_amd/csrc with hipcc for gfx950 and links one shared
library next to the sources.  hipcc cross-compiles without a GPU, so this runs in the CPU-only build container;
the resulting .so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmudg_hip.so")
OUT_FP16 = os.path.join(HERE, "libmudg_hip_fp16.so")
OUT_X3 = os.path.join(HERE, "libmudg_hip_x3.so")
OUT_X6 = os.path.join(HERE, "libmudg_hip_x6.so")
OUT_X3_DBG = os.path.join(HERE, "libmudg_hip_x3_dbg.so")  # bf16x3 + kernel-variant switches (same use)
OUT_DBG = os.path.join(HERE, "libmudg_hip_dbg.so")      # bf16 + kernel-variant switches (tests / tools only; hip.py loads it under MUDG_DEBUG_VARIANTS=1)
SOURCES = ["capi.hip", "gemm.hip", "pgemm.hip", "wgemm.hip", "attention.hip", "norm.hip", "misc.hip", "post.hip", "train.hip", "attention_bwd.hip", "wgrad.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off", *os.environ.get("MUDG_EXTRA_HIPCC_FLAGS", "").split()]      # extra flags: kernel experiments only


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the gfx950 kernels cannot be built")
    return exe


def _stamp() -> str:
    h = hashlib.sha256()
    names = [n for n in sorted(os.listdir(CSRC)) if n.endswith((".hip", ".h"))] + ["../../include/mudg_hip.h"]
    for name in names:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _build_one(out: str, extra, tag: str, force: bool, verbose: bool) -> str:
    stamp_file = out + ".stamp"
    stamp = _stamp()
    if not force and os.path.exists(out) and os.path.exists(stamp_file):
        with open(stamp_file) as f:
            if f.read().strip() == stamp:
                return out
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build", tag)
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return out


def build(force: bool = False, verbose: bool = True, only=None) -> str:
    """Every operand-type build of the same sources: bf16 (default), fp16 (-DMUDG_OPERAND_FP16), the split-operand
    precision modes bf16x3 / bf16x6 (-DMUDG_PLANES=2 / 3; csrc/common.h), and the bf16 build with the kernel-variant
    switches compiled in (-DMUDG_DEBUG_VARIANTS: the only library in which an environment variable can change which
    kernel runs; the variant tests and the A/B tools load it).  The five libraries are built side by side;
    `only` (or `--only dbg,bf16` on the command line) restricts the set while iterating on a kernel."""
    jobs = {"fp16": (OUT_FP16, ["-DMUDG_OPERAND_FP16"]), "x3": (OUT_X3, ["-DMUDG_PLANES=2"]), "x6": (OUT_X6, ["-DMUDG_PLANES=3"]),
            "dbg": (OUT_DBG, ["-DMUDG_DEBUG_VARIANTS"]), "x3dbg": (OUT_X3_DBG, ["-DMUDG_PLANES=2", "-DMUDG_DEBUG_VARIANTS"]), "bf16": (OUT, [])}
    tags = [t for t in jobs if only is None or t in only]
    with ThreadPoolExecutor(max_workers=len(tags)) as ex:
        list(ex.map(lambda t: _build_one(jobs[t][0], jobs[t][1], t, force, verbose), tags))
    return OUT


if __name__ == "__main__":
    sel = None
    if "--only" in sys.argv:
        sel = sys.argv[sys.argv.index("--only") + 1].split(",")
    build(force="--force" in sys.argv, only=sel)
    print(OUT)
