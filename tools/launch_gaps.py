#!/usr/bin/env python3
"""Mean duration of the trace kernel and mean idle gap between consecutive trace launches, from a rocprofv3 kernel trace:
   tools/launch_gaps.py <kernel_trace.csv>"""
import csv
import re
import sys

pat = re.compile(r"ClosestHitKernel<\d, (?:true|false), \d, false")   # the uninstrumented trace kernels
rows = [r for r in csv.DictReader(open(sys.argv[1])) if pat.search(r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
gaps = [g for g in gaps if 0 <= g < 100]          # launches separated by other phases of the run are not neighbours
print(f"{len(rows)} launches: mean {sum(dur) / len(dur):.2f} us, median {sorted(dur)[len(dur) // 2]:.2f} us; "
      f"gap to the next launch mean {sum(gaps) / max(1, len(gaps)):.2f} us, median {sorted(gaps)[len(gaps) // 2]:.2f} us over {len(gaps)} pairs")
