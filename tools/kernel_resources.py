#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in skx_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "sketchy_amd", "csrc", "skx_kernels.hip")
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I",
                      os.path.dirname(src), "-c", src, "-o", "/tmp/skx_kernels_res.o", "-Rpass-analysis=kernel-resource-usage"],
                     capture_output=True, text=True).stderr
rows, cur = [], {}
for ln in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur:
            rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if cur:
    rows.append(cur)
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(.*", "", n).replace("void skx::", "")
    if pat and not re.search(pat, n):
        continue
    print(f"{n:60s} sgpr {r.get('TotalSGPRs', '?'):>4} vgpr {r.get('VGPRs', '?'):>4} scratch {r.get('ScratchSize [bytes/lane]', '?'):>4} "
          f"occ {r.get('Occupancy [waves/SIMD]', '?'):>2} lds {r.get('LDS Size [bytes/block]', '?'):>6}")
