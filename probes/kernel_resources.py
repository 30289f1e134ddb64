#!/usr/bin/env python3
"""Register / LDS / spill table of every kernel in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python probes/kernel_resources.py iggt_official_amd/csrc/attention_v3.hip [extra hipcc flags]"""
import re
import subprocess
import sys

sys.path.insert(0, __file__.rsplit("/probes/", 1)[0])
from iggt_official_amd import build_ext  # noqa: E402

src = sys.argv[1]
cmd = [build_ext.HIPCC] + build_ext.FLAGS + sys.argv[2:] + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    print(f"{name[:90]:90s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>4s} spill {r.get('VGPRs Spill', r.get('VGPR Spill','?')):>4s} "
          f"scratch {r.get('ScratchSize [bytes/lane]','?'):>5s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} lds {r.get('LDS Size [bytes/block]','?')}")
