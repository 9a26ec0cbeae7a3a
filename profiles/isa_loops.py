"""Instruction counts of the loop bodies of a kernel, from the compiler's own assembly (hipcc -S with the product's flags): for every loop
(the blocks the compiler marks "in Loop: Header=..."), VALU / SALU / vector memory / LDS / branch instructions in its body, the loops
nested in it included.  A wavefront that runs a loop body under partial masks issues every instruction of every block some lane enters.
    python profiles/isa_loops.py 'k_trace<false, true>' [min_instructions]"""
import collections, re, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from yetanotherconsolegameengine_amd.build import FLAGS  # noqa: E402
want = sys.argv[1] if len(sys.argv) > 1 else "k_trace<false, true>"
floor = int(sys.argv[2]) if len(sys.argv) > 2 else 40
src = ROOT / "yetanotherconsolegameengine_amd" / "csrc" / "ycge_kernels.hip"
with tempfile.TemporaryDirectory() as td:
    subprocess.run(["/opt/rocm/bin/hipcc"] + [f for f in FLAGS if f not in ("-shared", "-fPIC")] + ["--cuda-device-only", "-S", "-x", "hip", str(src), "-o", f"{td}/k.s"],
                   check=True, capture_output=True)
    text = open(f"{td}/k.s").read().splitlines()
names = {}
for ln, l in enumerate(text):
    m = re.match(r"^(_Z\w+):", l)
    if m: names[m.group(1)] = ln
dem = subprocess.run(["c++filt"] + list(names), capture_output=True, text=True).stdout.splitlines()
pick = [n for n, d in zip(names, dem) if want in d.replace("ycge::", "")]
assert pick, f"no kernel matches {want!r}"
start = names[pick[0]]
end = next(i for i in range(start, len(text)) if text[i].strip().startswith("s_endpgm"))
loops = collections.OrderedDict()          # header -> counts; a block counts for every loop it is nested in
cur_loops = []
block_re = re.compile(r"^\.LBB\d+_\d+:\s*;\s*(.*)$")
first_line = {}
for i in range(start, end):
    l = text[i]
    m = block_re.match(l)
    if m:
        cur_loops = re.findall(r"Header=(BB\d+_\d+) Depth=(\d+)", m.group(1))
        own = re.search(r"=>This Loop Header: Depth=(\d+)", m.group(1)) or re.search(r"Loop Header: Depth=(\d+)", m.group(1))
        if own:
            cur_loops = cur_loops + [(l.split(":")[0].lstrip("."), own.group(1))]
        continue
    if l.startswith(".LBB") or l.startswith("; %bb"):
        continue
    ins = l.strip().split()
    if not ins or ins[0].startswith(";") or ins[0].startswith("."):
        continue
    op = ins[0]
    kind = ("vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "lds" if op.startswith("ds_") else "valu" if op.startswith("v_") else
            "branch" if op.startswith(("s_cbranch", "s_branch")) else "wait" if op.startswith(("s_waitcnt", "s_nop")) else "salu" if op.startswith("s_") else "other")
    for h, d in cur_loops:
        c = loops.setdefault((h, int(d)), collections.Counter())
        c[kind] += 1; c["all"] += 1
        first_line.setdefault((h, int(d)), i - start)
        if op in ("ds_bpermute_b32", "v_readlane_b32", "v_mfma"): c["x_" + op] += 1
print(f"{want}: {end - start} lines of assembly; loops of >= {floor} instructions (nested loops included in their parents)")
for (h, d), c in loops.items():
    if c["all"] < floor: continue
    print(f"  {'  ' * (d - 1)}{h:10s} depth {d}  at +{first_line[(h, d)]:5d}: {c['all']:5d} instructions = {c['valu']:4d} VALU {c['salu']:4d} SALU {c['vmem']:3d} vector memory {c['lds']:3d} LDS "
          f"{c['branch']:3d} branches {c['wait']:3d} waits/nops" + (f"  ({c['x_ds_bpermute_b32']} ds_bpermute)" if c["x_ds_bpermute_b32"] else ""))
