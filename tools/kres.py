"""Register / LDS / spill report per kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage, the library's flags).

    python tools/kres.py gemm.hip [name-substring]

Run in the build container (no GPU needed); the numbers quoted in DESIGN.md for VGPR budgets and spills come from here."""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from painter_amd import build as B

src = os.path.join(B.CSRC, sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
extra = B.EXTRA.get(os.path.basename(src), [])
r = subprocess.run([B._hipcc()] + B.FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
cur, rows = None, {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for name, d in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if pat and pat not in dem:
        continue
    print("%-4s vgpr %-4s agpr %-4s sgpr spill v/s %s/%s  scratch %-5s occ %s lds %-6s  %s" % (
        d.get("VGPRs", "?"), d.get("AGPRs", "?"), d.get("TotalSGPRs", "?"), d.get("VGPRs Spill", d.get("VGPR Spill", "?")), d.get("SGPRs Spill", d.get("SGPR Spill", "?")),
        d.get("ScratchSize [bytes/lane]", "?"), d.get("Occupancy [waves/SIMD]", "?"), d.get("LDS Size [bytes/block]", "?"), dem[:150]))
