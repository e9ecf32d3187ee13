#!/usr/bin/env python3
"""Folds tools/profile_gpu.sh outputs (gpurun_out/prof_<tag>/) into profiles/<round>/: the text summary per tag,
one row of pmc_table.json per tag (keyed as bench.py looks it up) and, from the gather microbenchmark's log,
l2_gather_peak.json.  Every row carries the git commit and a hash of the kernel sources (trace_kernels.hip +
trace_math.h) of the tree it was collected in -- run this right after the profile, before editing the kernels:
bench.py uses a row's counters only while the tree still hashes the same.
usage: tools/collect_profiles.py r02 tag=key [tag=key ...] [gather=LOGFILE]"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
table_path = os.path.join(dst, "pmc_table.json")
table = json.load(open(table_path)) if os.path.exists(table_path) else {}


def kernel_source_hash():
    """sha256 over the kernel sources, the same way bench.py computes it."""
    h = hashlib.sha256()
    for name in ("trace_kernels.hip", "trace_math.h"):
        h.update(open(os.path.join(ROOT, "raytrace_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


git = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "raytrace_amd/csrc"], capture_output=True, text=True).stdout.strip())
code = {"git": git + ("+uncommitted" if dirty else ""), "kernel_source_sha256": kernel_source_hash()}
for arg in sys.argv[2:]:
    tag, key = arg.split("=", 1)
    if tag == "gather":
        best = 0.0
        for line in open(key):
            m = re.search(r"table\s+2048 KB share 1 active 64: .* ([0-9.]+) lane-loads/ns/CU", line)
            if m:
                best = float(m.group(1))
        json.dump({"l2_requests_per_ns_chip": round(best * 256, 1), "per_cu": best,
                   "how": "tools/ubench/gather_rate: 64 lanes x 16-byte loads at random records of a 2 MiB table (L2-resident, "
                          "beyond the 32 KiB L1), 32 waves per CU, every CU; one request per lane-load",
                   "as_128B_lines_TBs": round(best * 256 * 128 / 1e3, 1)}, open(os.path.join(dst, "l2_gather_peak.json"), "w"), indent=1)
        shutil.copy(key, os.path.join(dst, "gather_rate.txt"))
        continue
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    table[key] = json.load(open(os.path.join(src, "pmc_entry.json")))
    table[key]["code"] = code
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, tag + "_summary.txt"))
    for f in os.listdir(os.path.join(src, "trace")) if os.path.isdir(os.path.join(src, "trace")) else []:
        pass
    print(tag, "->", key, "kernel_us", table[key].get("kernel_us_profiled"))
json.dump(table, open(table_path, "w"), indent=1, sort_keys=True)
