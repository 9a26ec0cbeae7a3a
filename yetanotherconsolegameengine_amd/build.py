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
SOURCES = ["ycge_host.cpp", "ycge_frame.cpp", "ycge_post_host.cpp", "ycge_resident.cpp", "ycge_accel.cpp", "ycge_kernels.hip", "ycge_post.hip", "ycge_bvh_build.hip"]
HEADERS = ["ycge_ctx.h", "ycge_device.h", "ycge_accel.h", "ycge_math.h", "ycge_rt.hip.h", "ycge_coop.hip.h", "ycge_anyhit.hip.h", "ycge_keysort.h",
           "experiments/ycge_refill.hip.h", "experiments/ycge_atrous_persist_groups.hip.h", "experiments/ycge_taa_in_trace.hip.h"]
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


def stamp_path(lib: Path) -> Path:
    return lib.with_name(lib.name + ".srchash")


def is_stale(lib: Path = LIB, flags=()) -> bool:
    """A library is current when the stamp written beside it names THIS content of the sources, headers and flags (content, not mtime:
    a snapshot copied to another box keeps no useful timestamps)."""
    try:
        return not lib.exists() or stamp_path(lib).read_text().strip() != source_hash() + "".join(" " + f for f in flags)
    except OSError:
        return True


class build_lock:
    """One builder at a time per OUTPUT FILE of a checkout (ranks of a launcher, pytest workers; different libraries build side by side):
    an exclusive flock on lib/.build.<name>.lock."""

    def __init__(self, out: Path = LIB):
        self.name = out.name

    def __enter__(self):
        import fcntl
        LIB_DIR.mkdir(exist_ok=True)
        self.f = open(LIB_DIR / f".build.{self.name}.lock", "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.f, fcntl.LOCK_UN)
        self.f.close()


def compile_to(out: Path, flags=(), verbose: bool = False) -> None:
    """hipcc into a temporary name, then an atomic rename: a process that loads `out` meanwhile sees the old or the new file, never half of one."""
    tmp = out.with_name(out.name + f".tmp{os.getpid()}")
    cmd = [hipcc(), *FLAGS, *flags, "-x", "hip", *[str(CSRC / s) for s in SOURCES], "-o", str(tmp)]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        tmp.unlink(missing_ok=True)
        raise RuntimeError(f"hipcc failed building {out.name}")
    if verbose and r.stderr:
        print(r.stderr)
    os.replace(tmp, out)
    stamp_path(out).write_text(source_hash() + "".join(" " + f for f in flags) + "\n")


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> Path:
    if not force and not is_stale(LIB, extra_flags):
        return LIB
    with build_lock(LIB):
        if force or is_stale(LIB, extra_flags):          # (someone else may have built it while we waited for the lock)
            compile_to(LIB, extra_flags, verbose)
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
    # the counting TWIN of the timed stage kernels of voxel worlds: the same non-counting instances with per-lane counters of what they
    # walk (scene-tree steps by kind, cell steps, cell fetches, queue records).  bench.py --config 5 replays its timed frames through it for
    # `roofline.timed_work`; profiles/vox_stats.py reads the per-phase clocks of k_wf_trace_p from it
    "voxstat": ["-DYCGE_DBG_VOXSTAT=1"],
    # the exception barrier of the C-ABI under allocation failure: the library's own operator new (-Bsymbolic: bound inside the library) throws std::bad_alloc on
    # the n-th call after ycge_debug_fail_allocation(n) - tests/test_gpu_abi_barrier.py walks n through ycge_scene_upload and ycge_create
    "faultinject": ["-DYCGE_FAULT_INJECTION=1", "-Wl,-Bsymbolic"],
}


def variant_path(name: str) -> Path:
    return LIB_DIR / f"var_{name}.so"


def build_variant(name: str, flags=None, force: bool = False) -> Path:
    flags = list(VARIANTS[name] if flags is None else flags)
    out = variant_path(name)
    if not force and not is_stale(out, flags):
        return out
    with build_lock(out):
        if force or is_stale(out, flags):
            compile_to(out, flags)
    return out


def build_all(verbose: bool = False):
    """The product and every test variant, side by side (one hipcc each: ~1 minute of wall time instead of one per library)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as ex:
        futs = [ex.submit(build_library, False, verbose)] + [ex.submit(build_variant, v) for v in VARIANTS]
        return [f.result() for f in futs]


if __name__ == "__main__":
    if "--hash" in sys.argv:
        print(source_hash()); sys.exit(0)
    p = build_library(force="--force" in sys.argv, verbose=True)
    print("built", p)
