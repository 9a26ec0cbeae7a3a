"""Builds libycge_hip.so (host C++ + gfx950 HIP kernels) in-tree with hipcc.

    python -m yetanotherconsolegameengine_amd.build [--force]

Flags that matter (see csrc/ycge_math.h): -ffp-contract=off keeps every fp32 operation
individually rounded (bit-parity with the reference's scalar C#); division and sqrt stay
on hipcc's IEEE-correct default expansions; no -ffast-math anywhere.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB = LIB_DIR / "libycge_hip.so"
SOURCES = ["ycge_host.cpp", "ycge_accel.cpp", "ycge_kernels.hip", "ycge_post.hip", "ycge_bvh_build.hip"]
HEADERS = ["ycge_device.h", "ycge_accel.h", "ycge_math.h", "ycge_rt.hip.h", "ycge_coop.hip.h", "ycge_anyhit.hip.h", "ycge_keysort.h",
           "experiments/ycge_refill.hip.h", "experiments/ycge_atrous_persist_groups.hip.h"]
ARCH = "gfx950"

FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-fno-gpu-rdc", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise FileNotFoundError("hipcc not found (ROCm toolchain required; there is no CPU build of the kernels)")


def source_hash() -> str:
    """sha256 (16 hex digits) over the kernel / host sources and the build flags: names the BUILD a measurement belongs to (the
    rocprofv3 counter summaries under profiles/ carry it; bench.py prints counter-derived figures only for the build it runs)."""
    import hashlib
    h = hashlib.sha256()
    for n in sorted(SOURCES + HEADERS):
        h.update(n.encode()); h.update((CSRC / n).read_bytes())
    h.update((PKG.parent / "include" / "ycge.h").read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def is_stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / n for n in SOURCES + HEADERS] + [PKG.parent / "include" / "ycge.h", Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> Path:
    if not force and not is_stale():
        return LIB
    LIB_DIR.mkdir(exist_ok=True)
    cmd = [hipcc(), *FLAGS, *extra_flags, "-x", "hip", *[str(CSRC / s) for s in SOURCES], "-o", str(LIB)]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libycge_hip.so")
    if verbose and r.stderr:
        print(r.stderr)
    return LIB


# Test variants of the library (lib/var_<name>.so, loaded by path): the same sources with other -D settings.  Built by
# __graft_entry__.build() next to the product so that they travel to the GPU box with it.
VARIANTS = {
    # the work list of the order-free occlusion queries cut to 112 entries (HIGH = 32): wide rounds are cut back and one-item dives happen
    # on ordinary scenes, so the GPU tests see every mode of the list (csrc/ycge_anyhit.hip.h)
    "bfs112": ["-DYCGE_BFS_LIST=112u"],
    # the measured-and-rejected kernel forms of csrc/experiments/ (k_trace_refill, the group hand-over A-trous): out of the product build,
    # kept bit-exact by their parity tests through this one
    "experiments": ["-DYCGE_EXPERIMENTS=1"],
}


def variant_path(name: str) -> Path:
    return LIB_DIR / f"var_{name}.so"


def build_variant(name: str, flags=None, force: bool = False) -> Path:
    flags = list(VARIANTS[name] if flags is None else flags)
    out = variant_path(name)
    deps = [CSRC / n for n in SOURCES + HEADERS] + [PKG.parent / "include" / "ycge.h", Path(__file__)]
    if not force and out.exists() and all(d.stat().st_mtime <= out.stat().st_mtime for d in deps):
        return out
    LIB_DIR.mkdir(exist_ok=True)
    r = subprocess.run([hipcc(), *FLAGS, *flags, "-x", "hip", *[str(CSRC / s) for s in SOURCES], "-o", str(out)], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError(f"hipcc failed building {out.name}")
    return out


if __name__ == "__main__":
    if "--hash" in sys.argv:
        print(source_hash()); sys.exit(0)
    p = build_library(force="--force" in sys.argv, verbose=True)
    print("built", p)
