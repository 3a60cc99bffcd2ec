#!/usr/bin/env python
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: VGPRs, scratch, occupancy, LDS per kernel."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
seen = set()
for b in blocks:
    name = b.split("\n")[0].strip()
    if name in seen:
        continue
    seen.add(name)

    def g(key):
        m = re.search(key + r": (\d+)", b)
        return int(m.group(1)) if m else -1

    dem = subprocess.run(["c++filt", name.split()[0]], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void nq::", "")
    if pat and not re.search(pat, dem):
        continue
    print("%-46s vgpr=%4d agpr=%3d scratch=%4d occ=%d lds=%6d sgpr=%3d" % (
        dem[:46], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
        g(r"LDS Size \[bytes/block\]"), g("SGPRs")))
