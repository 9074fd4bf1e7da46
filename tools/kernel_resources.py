#!/usr/bin/env python3
"""tools/kernel_resources.py <resource-usage.txt> [filter] -- table of VGPRs / scratch / occupancy / LDS per kernel from
`hipcc -Rpass-analysis=kernel-resource-usage` remarks (no GPU needed: the register budget of a kernel variant is known before
it ever runs)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    d = re.sub(r"\(.*", "", d).replace("hipbfv::", "").replace("void ", "")
    if flt and not re.search(flt, d):
        continue
    print(f"{d[:84]:84s} vgpr={g('VGPRs'):4d} agpr={g('AGPRs'):3d} scratch={g('ScratchSize .bytes/lane.'):4d} occ={g('Occupancy .waves/SIMD.')} lds={g('LDS Size .bytes/block.'):6d}")
