#!/usr/bin/env python3
"""Table of the compiler's resource remarks (make -C raytracer_challenge_amd/csrc usage 2> file) per kernel."""
import re
import sys

rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    t = re.search(r"(render_weighted_kernel|render_kernel)ILi(\d+)ELb(\d)ELb(\d)ELi(\d)E", r["name"])
    if not t:
        continue
    print(f"{t.group(1):24s} MAXD={t.group(2):2s} STATS={t.group(3)} RAYS={t.group(4)} TABLES={t.group(5)}  "
          f"vgpr={r.get('VGPRs')} vspill={r.get('VGPRs Spill')} sspill={r.get('SGPRs Spill')} scratch={r.get('ScratchSize')} occ={r.get('Occupancy')}")
