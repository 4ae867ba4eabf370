"""Build the HIP extension in-tree: csrc/libmppi_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmppi_hip.so")
SOURCES = ["mppi_capi.hip"]
HEADER = os.path.join(HERE, "..", "include", "mppi_hip.h")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def deps() -> list:
    """Everything the library is compiled from: every .hip / .hpp / .inc under csrc/ (mppi_models.inc holds all the
    model arithmetic and is included twice by mppi_models.hpp) plus the public header."""
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".inc", ".h"))]
    return out + [HEADER]


def source_digest() -> str:
    """sha256 over everything the library is compiled from (deps(), in order, file names included): what profiles/
    pmc_constants.json is keyed on, so that counters measured on another build of the kernels are recognised as stale."""
    import hashlib

    h = hashlib.sha256()
    for path in deps():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in deps())


def build(force: bool = False, verbose: bool = False) -> str:
    if force or stale():
        cmd = [hipcc(), *FLAGS, "-o", LIB, *SOURCES]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
