#!/usr/bin/env python3
"""Reads bench.py JSON lines on stdin and prints one short line each (tag = argv[1])."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d["roofline"]
    g = r["algorithmic"]
    print(f"{tag:40s} {d['value']:9.1f} Mrays/s  step {d['ms_per_step']:.4f} ms  kernel {r['kernel_ms']:.4f} ms  {r['bound']} frac {r['frac']:.3f}  "
          f"nodes/ray {g['node_records_per_ray']:.2f} tris/ray {g['tri_records_per_ray']:.2f}  B/ray {g['bytes_per_ray']:.0f}", flush=True)
