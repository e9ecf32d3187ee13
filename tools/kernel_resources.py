#!/usr/bin/env python3
"""Prints VGPR / SGPR / scratch / occupancy per kernel as reported by hipcc for gfx950."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-fno-slp-vectorize", "-Iinclude", "-c", "raytrace_amd/csrc/trace_kernels.hip",
       "-o", "/tmp/tk_res.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True).stderr
cur = {}


def flush():
    if cur:
        n = cur.get("name", "")
        m = re.search(r"ClosestHitKernelILi(\d)ELb([01])ELi([012])ELb([01])ELb([01])E", n)
        if m:
            n = "ClosestHit<width=%s quant=%s mode=%s count=%s split=%s>" % m.groups()
        print("%-62s vgpr=%s sgpr=%s scratch=%s waves/simd=%s" % (n, cur.get("v"), cur.get("s"), cur.get("sc"), cur.get("o")))


for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        flush()
        cur = {"name": t.split(":", 1)[1].strip()}
    elif t.startswith("VGPRs:"):
        cur["v"] = t.split()[-1]
    elif t.startswith("TotalSGPRs:"):
        cur["s"] = t.split()[-1]
    elif t.startswith("ScratchSize"):
        cur["sc"] = t.split()[-1]
    elif t.startswith("Occupancy"):
        cur["o"] = t.split()[-1]
flush()
