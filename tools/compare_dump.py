#!/usr/bin/env python3
"""A dump of the REFERENCE (tools/ReferenceDump) against this repository's ORACLE, on any machine with g++ and numpy - no GPU, no torch.

    python tools/compare_dump.py <dump dir> [<dump dir> ...]        # each holds scene.ysc, meta.json and the files of Program.cs
    python tools/compare_dump.py                                    # every dump under tests/golden/reference/ and $YCGE_REFERENCE_GOLDENS

Builds oracle/liborc_oracle.so (g++ -O2 -ffp-contract=off) if needed, rebuilds each scene from its scene.ysc, renders the dump's frames with
the oracle and prints, per dump, the table of tools/ReferenceDump/README.md: builders bit-exact; rays bit-exact (or <= 1e-6 RMS where the
hosts' sinf / cosf / tanf differ); G-buffer bit-exact wherever the primary ray is; radiance, TAA history and SDR chexels within 1e-4 RMS
(north_star).  Exit status 0 = every dump within its bars: THAT is what turns "parity unpinned"  into a measurement;
commit the dumps that are small under tests/golden/reference/<name>/ and tests/test_reference_goldens.py holds the HIP path to them too.
The checking code is tests/reference_dump.py - the same functions the test suite uses.
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "tests", ROOT / "tools"):
    sys.path.insert(0, str(p))

import numpy as np  # noqa: E402

import oracle_binding as ob  # noqa: E402
import reference_dump as rd  # noqa: E402
import scene_file  # noqa: E402


def compare(d: Path, threads: int = 8) -> int:
    dump = rd.Dump(d)
    back = scene_file.LoadedScene(d / "scene.ysc")
    if (back.fb_width, back.fb_height) != (dump.fbW, dump.fbH):
        print(f"{d}: scene.ysc ({back.fb_width}x{back.fb_height}) is not the file the dump ({dump.fbW}x{dump.fbH}) was made from")
        return 1
    print(f"== {d.name}: console {dump.fbW}x{dump.fbH}, trace grid {dump.hiW}x{dump.hiH}, {dump.frames} frame(s), {dump.meta.get('n_meshes', 0)} mesh(es); dumped by: {dump.meta.get('runtime', '?')}")
    o = ob.OracleRenderer(None, back.fb_width, back.fb_height, back.super_sample, back.pose, flat=back)
    bad = rd.compare_accel(dump, o, dump.meta.get("n_meshes", 0))
    print(f"   builders (scene BVH + {dump.meta.get('n_meshes', 0)} mesh BVH): {'bit-exact' if not bad else '; '.join(bad)}")
    print("   frame | rays bit-exact  rays RMS | G-buffer mismatches where rays agree (albedo normal depth sky) | radiance RMS (bit-exact) | history RMS (bit-exact) | SDR RMS (bit-exact)")
    for k in range(1, dump.frames + 1):
        sdr = o.render(stages=2, threads=threads, want_sdr=True)
        rep, b = rd.compare_frame(dump, k, o, sdr)
        bad += b
        g = [rep[n + "_mismatch_where_rays_agree"] for n in ("g_albedo.f32", "g_normal.f32", "g_depth.f32", "sky.u8")]
        print(f"   {k:5d} | {100 * rep['rays_bit_exact']:8.4f} %  {rep['rays_rms']:.2e} | {g[0]:6d} {g[1]:6d} {g[2]:6d} {g[3]:6d} | "
              f"{rep['current_hdr.f32_rms']:.2e} ({100 * rep['current_hdr.f32_bit_exact']:.2f} %) | {rep['taa_history.f32_rms']:.2e} ({100 * rep['taa_history.f32_bit_exact']:.2f} %) | "
              f"{rep['sdr_rms']:.2e} ({100 * rep['sdr_bit_exact']:.2f} %)")
    o.close()
    for line in bad:
        print("   OUTSIDE THE BAR:", line)
    print(f"   -> {'PINNED: the oracle reproduces the reference within the bars' if not bad else 'NOT pinned'}")
    return 1 if bad else 0


def main(argv):
    ob.build_oracle()
    dirs = [Path(a) for a in argv] or rd.dump_dirs()
    if not dirs:
        print("no dump found: run tools/ReferenceDump/run_all.sh on a machine with the .NET 8 SDK first (tools/ReferenceDump/README.md)")
        return 2
    missing = [d for d in dirs if not (d / "meta.json").exists() or not (d / "scene.ysc").exists()]
    if missing:
        print("not a dump (needs scene.ysc + meta.json):", ", ".join(str(d) for d in missing))
        return 2
    return max(compare(d) for d in dirs)


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
