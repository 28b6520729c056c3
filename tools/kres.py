#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of one .hip source: `python tools/kres.py mudg_amd/csrc/gemm.hip [-DMUDG_PLANES=2 ...]`
(hipcc -Rpass-analysis=kernel-resource-usage, condensed to one line per kernel)."""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *extra, "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for name, r in rows.items():
    print(f"{name:58s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>4s} SGPR {r.get('TotalSGPRs','?'):>4s} "
          f"scratch {r.get('ScratchSize [bytes/lane]','?'):>5s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} LDS {r.get('LDS Size [bytes/block]','?')}")
