"""In-tree build of libhgnn_hip.so with hipcc for gfx950 (MI355X / CDNA4).

    python -m hierarchicalgnn_amd.build [--force]

hipcc cross-compiles without a GPU.  The shared object is written next to the
sources (hierarchicalgnn_amd/csrc/libhgnn_hip.so); it is git-ignored but
travels to the GPU box with the working tree.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libhgnn_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + \
        [os.path.join(HERE, "..", "include", "hgnn_hip.h")]


def up_to_date() -> bool:
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(p) <= t for p in _deps())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libhgnn_hip.so")
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        if not force and os.path.exists(obj) and all(
                os.path.getmtime(p) <= os.path.getmtime(obj)
                for p in [src] + [d for d in _deps() if d.endswith(".h")]):
            continue
        cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("build failed: " + " ".join(cmd))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
