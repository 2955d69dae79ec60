#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS report of one HIP source: compiles it for gfx950 with -Rpass-analysis=kernel-resource-usage and
prints one line per kernel instantiation (demangled).  usage: tools/kernel_resources.py csrc/gemm.hip [filter-regex] [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-pass-failed",
                    "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra, capture_output=True, text=True)
blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")
for b, d in zip(blocks, dem):
    g = lambda k: int(re.search(re.escape(k) + r": (\d+)", b).group(1))
    m = re.search(r"(\w+<.*>)\(", d)
    nm = (m.group(1) if m else d).replace("gtav::(anonymous namespace)::", "")
    if flt and not re.search(flt, nm):
        continue
    print(f"{nm:70s} VGPR {g('VGPRs'):4d} AGPR {g('AGPRs'):4d} SGPR {g('TotalSGPRs'):4d} scratch {g('ScratchSize [bytes/lane]'):5d} "
          f"spill s/v {g('SGPRs Spill')}/{g('VGPRs Spill')} occ {g('Occupancy [waves/SIMD]')} LDS {g('LDS Size [bytes/block]')}")
