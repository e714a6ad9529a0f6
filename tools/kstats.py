#!/usr/bin/env python3
"""tools/kstats.py FILE.hip -- per-kernel VGPR/SGPR/LDS/scratch/occupancy from hipcc remarks."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics",
       "-I", os.path.join(root, "include"), "-I", os.path.join(root, "gpt4roi_amd", "csrc"),
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
r = subprocess.run(cmd, capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur)
        cur = cur.split("(")[0][-70:]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][\w ]*?)\s*(?:\[[^\]]*\])?: (\S+)", line)
    if m and cur:
        rows[cur][m.group(1)] = m.group(2)
if r.returncode != 0:
    print(r.stderr[-3000:])
for k, v in rows.items():
    print(f"{k:72s} vgpr={v.get('VGPRs','?'):>4} agpr={v.get('AGPRs','?'):>4} sgpr={v.get('TotalSGPRs','?'):>4} "
          f"scratch={v.get('ScratchSize','?'):>5} occ={v.get('Occupancy','?'):>3} lds={v.get('LDS Size','?')}")
