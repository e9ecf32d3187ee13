#!/usr/bin/env python3
"""Condenses a tools/profile_gpu.sh output directory into a short text summary:
per-kernel time statistics from the kernel trace and per-launch means of every PMC counter for
the trace kernel that took most of the run (the batch kernel, or the lane-refilling one)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KERNEL = "ClosestHitKernel"


def rows(pattern):
    for p in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(p, newline="") as f:
            yield from csv.DictReader(f)


print("== kernel trace (per kernel: launches, mean/min/max duration in us) ==")
dur = defaultdict(list)
for r in rows("trace/**/*kernel_trace.csv"):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
KERNEL = max(("ClosestHitKernel", "RefillHitKernel"), key=lambda name: sum(sum(v) for k, v in dur.items() if name in k))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{len(v):6d}  mean {sum(v)/len(v):10.2f}  min {min(v):10.2f}  max {max(v):10.2f}  total {sum(v)/1e3:9.2f} ms  {k[:110]}")
for r in rows("trace/**/*kernel_trace.csv"):
    if KERNEL in r["Kernel_Name"]:
        keys = [k for k in r if any(s in k for s in ("VGPR", "SGPR", "LDS", "Scratch", "Workgroup_Size", "Grid_Size"))]
        print("launch shape:", {k: r[k] for k in keys})
        break

print("\n== PMC counters, mean per launch of", KERNEL, "==")
acc = defaultdict(list)
for r in rows("pmc_*/**/*counter_collection.csv"):
    if KERNEL in r.get("Kernel_Name", ""):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:36s} {sum(v)/len(v):18.1f}   (n={len(v)})")
m = {k: sum(v) / len(v) for k, v in acc.items()}
kd = [v for k, v in dur.items() if KERNEL in k]
if kd:
    m["kernel_us_profiled"] = sum(sum(v) for v in kd) / sum(len(v) for v in kd)
import json as _json
with open(os.path.join(out, "pmc_entry.json"), "w") as f:   # every counter's per-launch mean: one row of profiles/rNN/pmc_table.json
    _json.dump({k: round(v, 2) for k, v in m.items()}, f, indent=1, sort_keys=True)
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    rd, wr = m["FETCH_SIZE"] * 1024, m["WRITE_SIZE"] * 1024
    print(f"\nHBM-side bytes per launch: read {rd/1e6:.1f} MB as counted, {2*rd/1e6:.1f} MB with the gfx950 x2 "
          f"wide-read correction; written {wr/1e6:.1f} MB")
    import json
    with open(os.path.join(out, "hbm_traffic_entry.json"), "w") as f:
        json.dump({"gb_per_launch": round((2 * rd + wr) / 1e9, 4), "fetch_size_kb": m["FETCH_SIZE"], "write_size_kb": m["WRITE_SIZE"],
                   "note": "2 x FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE, mean per launch"}, f)
if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
    print(f"L2 hit rate: {m['TCC_HIT_sum']/(m['TCC_HIT_sum']+m['TCC_MISS_sum']):.4f}")
if "SQ_ACTIVE_INST_VALU" in m and "SQ_BUSY_CYCLES" in m:
    print(f"VALU active / SQ busy cycles: {m['SQ_ACTIVE_INST_VALU']/m['SQ_BUSY_CYCLES']:.4f}")
if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
    print(f"VALU lane utilisation (thread cycles / (active cycles x 64)): {m['SQ_THREAD_CYCLES_VALU']/(m['SQ_ACTIVE_INST_VALU']*64):.4f}")

if "SQ_ACTIVE_INST_VALU" in m and "SQ_BUSY_CYCLES" in m:
    # SQ counters are summed over the 32 shader engines and count in units of four cycles; a shader engine has
    # 8 CUs x 4 SIMDs, so 8 quad-cycles of VALU issue are on offer per SE cycle
    print(f"VALU issue utilisation while busy (active quad-cycles / (busy cycles x 8)): {m['SQ_ACTIVE_INST_VALU']/(m['SQ_BUSY_CYCLES']*8):.3f}")
if "SQ_WAVE_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
    print(f"mean resident waves per CU while busy (wave quad-cycles x 4 / busy cycles / 8 CUs): {m['SQ_WAVE_CYCLES']*4/m['SQ_BUSY_CYCLES']/8:.1f}")
if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m:
    print(f"VALU / SALU / VMEM-read / LDS instructions per wave: {m['SQ_INSTS_VALU']/m['SQ_WAVES']:.0f} / {m.get('SQ_INSTS_SALU',0)/m['SQ_WAVES']:.0f} / {m.get('SQ_INSTS_VMEM_RD',0)/m['SQ_WAVES']:.0f} / {m.get('SQ_INSTS_LDS',0)/m['SQ_WAVES']:.0f}")
