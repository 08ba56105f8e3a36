"""Compile libgumbi_hip.so for gfx950 in-tree (``gumbi_amd/lib/``) with hipcc.

The .so is git-ignored but travels with the repo snapshot to the GPU box; hipcc cross-compiles
without a GPU, so this also is the "does it build" check on CPU-only machines.
"""

from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "lib" / "libgumbi_hip.so"
SOURCES = ["engine.hip"]
HEADERS = sorted(p.name for p in CSRC.glob("*.hpp")) + ["../../include/gumbi_hip.h"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    newest = max((CSRC / f).resolve().stat().st_mtime for f in SOURCES + HEADERS)
    return newest > LIB.stat().st_mtime


def build_library(force: bool = False, verbose: bool = False) -> Path:
    """The product library.  ``GUMBI_BUILD_TUNING=1`` builds ``libgumbi_hip_tuning.so`` beside it instead, with
    ``-DGMB_TUNING`` (the environment switches of the tuning tools, tools/README.md; load it with GUMBI_HIP_LIB)."""
    tuning = os.environ.get("GUMBI_BUILD_TUNING") == "1"
    target = LIB.with_name("libgumbi_hip_tuning.so") if tuning else LIB
    if not force and not tuning and not needs_build():
        return LIB
    LIB.parent.mkdir(parents=True, exist_ok=True)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall",
           "-Wno-unused-function", *(["-DGMB_TUNING"] if tuning else []), *(os.environ.get("GUMBI_BUILD_DEFINES", "").split() if tuning else []),
           *[str(CSRC / s) for s in SOURCES], "-o", str(target)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return target


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
