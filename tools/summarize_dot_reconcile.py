#!/usr/bin/env python3
"""profiles/ summary for tools/dot_trace_reconcile.py: event times of the plain
run, event times of the SAME script under rocprofv3 --kernel-trace, and the
trace's own per-dispatch durations (kernel_trace.csv: End - Start of every dot
dispatch, in dispatch order: 5 warm-up, 50 single, 160 batched, 50 spaced).
Usage: summarize_dot_reconcile.py plain.json traced.json kernel_trace.csv"""
import csv
import json
import sys


def last_json(path):
    with open(path) as f:
        lines = [l for l in f if l.startswith("{")]
    return json.loads(lines[-1])


plain, traced = last_json(sys.argv[1]), last_json(sys.argv[2])
rows = []
with open(sys.argv[3]) as f:
    for row in csv.DictReader(f):
        name = row.get("Kernel_Name", "")
        if "kernel::dot<" in name or "kernel3dot" in name:
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
rows.sort()
dur = [(e - s) / 1e3 for s, e in rows]
groups = {"single": dur[5:55], "batch": dur[55:215], "spaced": dur[215:265]}
gaps = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(55, 214)
        if (i - 55) % 16 != 15]
print("dot n = 2^28 fp16/fp32 (1 GiB per launch), microseconds per launch")
print(f"{'':28s}{'single launches':>22s}{'16 back-to-back':>22s}{'idle GPU first':>22s}")
for label, data in (("HIP events, plain run", plain), ("HIP events under rocprofv3", traced)):
    print(f"{label:28s}" + "".join(
        f"{data[k]['min_us']:9.1f} /{data[k]['mean_us']:8.1f}   " for k in
        ("single", "batch", "spaced")) + " (min / mean)")
print(f"{'rocprofv3 dispatch duration':28s}" + "".join(
    f"{min(v):9.1f} /{sum(v) / len(v):8.1f}   " for v in
    (groups["single"], groups["batch"], groups["spaced"])) + " (min / mean)")
if gaps:
    print(f"gap between back-to-back dispatches in the trace: min {min(gaps):.2f} "
          f"mean {sum(gaps) / len(gaps):.2f} us ({len(gaps)} gaps)")
print(f"dot dispatches in the trace: {len(dur)} (expected {plain['launches']})")
