#!/usr/bin/env python3
"""Writes profiles/<tag>_trsv_dispatch_sequence.txt from the kernel trace tools/gpu_profile.sh
left under gpurun_out/prof_stats/: the TRSV dispatches of the bench command IN ORDER of
submission, per workload - is a long dispatch the first one, every k-th one, or at random?
usage: trsv_dispatch_sequence.py r05"""
import csv
import os
import sys

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
trace = os.path.join(REPO, "gpurun_out", "prof_stats", "gemv_kernel_trace.csv")
rows = [r for r in csv.DictReader(open(trace)) if "trsv_wide" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups, current = [], None
for r in rows:
    plain = ", float, false, accblas::dev::plain_view" in r["Kernel_Name"]
    key = (r["Grid_Size_X"], "plain fp32" if plain else "Acc<fp64, fp32>")
    if key != current:
        groups.append((key, []))
        current = key
    groups[-1][1].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
lines = [
    "TRSV dispatches of `python3 bench.py --steps 50 --warmup 5` under rocprofv3 --kernel-trace (the driver's command;",
    f"profiles/{tag}_gemv_kernel_stats_by_grid.csv has the same launches grouped by kernel and grid), durations in us IN ORDER",
    "of submission per workload: 5 warm-ups + 20 timed + 1 residual check each. The long dispatch of round 4's trace (1.61 ms at",
    "n = 65536, 0.41 ms at 24000) is the FIRST solve on a freshly written matrix - every kernel of the trace has such a first",
    "dispatch (GEMV ~650 against 613-619 us, dot ~175 against 156-160) -; n = 24000 needs ~8 solves to settle. bench.py times solves 6..25:"]
for (grid, kind), seq in groups:
    steady = seq[5:]
    size = "65536 lower" if int(grid) > 500000 else "24000 upper"
    lines.append(f"n = {size}, {kind} (grid {grid}): " + " ".join(f"{x:.0f}" for x in seq))
    lines.append(f"   all {len(seq)}: min {min(seq):.1f} max {max(seq):.1f} (max/min {max(seq) / min(seq):.3f}); "
                 f"dispatches 6..{len(seq)}: min {min(steady):.1f} mean {sum(steady) / len(steady):.1f} "
                 f"max {max(steady):.1f} (max/min {max(steady) / min(steady):.3f}, "
                 f"mean/min {sum(steady) / len(steady) / min(steady):.3f})")
out = os.path.join(REPO, "profiles", f"{tag}_trsv_dispatch_sequence.txt")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
