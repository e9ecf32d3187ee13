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
    print(f"{tag:40s} {d['value']:9.1f} Mrays/s  kernel {r['kernel_ms']:.4f} ms  frac {r['frac']:.3f}  "
          f"nodes/ray {r['node_records_per_ray']:.2f} tris/ray {r['tri_records_per_ray']:.2f}  B/ray {r['bytes_per_ray']:.0f}", flush=True)
