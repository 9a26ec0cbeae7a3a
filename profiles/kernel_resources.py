"""Prints VGPR / scratch / LDS / occupancy of every kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from yetanotherconsolegameengine_amd.build import FLAGS  # noqa: E402  (the product's own flags: the numbers are the shipped binary's)

args = [a for a in sys.argv[1:] if not a.endswith(".hip")]
names = [a for a in sys.argv[1:] if a.endswith(".hip")] or ["ycge_kernels.hip"]
src = ROOT / "yetanotherconsolegameengine_amd" / "csrc" / names[0]
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + [f for f in FLAGS if f not in ("-shared", "-fPIC")] +
                       ["--cuda-device-only", "-c", "-x", "hip", str(src), "-o", f"{td}/k.o", "-Rpass-analysis=kernel-resource-usage"] + args,
                       capture_output=True, text=True)
cur = {}
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if not rows:
    sys.exit("no kernels found (did the compile fail?)\n" + r.stderr[-2000:])
dem = subprocess.run(["c++filt"] + [r_["name"] for r_ in rows], capture_output=True, text=True).stdout.splitlines()
for r_, d in zip(rows, dem):
    d = d.replace("ycge::", "").split("(")[0].replace("void ", "")
    print(f"{d:46s} VGPR={r_.get('VGPRs','?'):>4s} SGPR={r_.get('TotalSGPRs','?'):>4s} scratch={r_.get('ScratchSize [bytes/lane]','?'):>5s} "
          f"LDS={r_.get('LDS Size [bytes/block]','?'):>6s} occ={r_.get('Occupancy [waves/SIMD]','?')}")
