#!/usr/bin/env python3
"""Condenses gpurun_out/pmc_<TAG>_<i>_<COUNTER>/ (tools/gpu_profile_trsv.sh) into
profiles/<round>_<TAG>_pmc.json: bytes per launch and the ratio to the
algorithmic bytes. FETCH_SIZE is doubled (gfx950 correction for wide streaming
reads, MI355X_MICROARCH.md §HBM); 8-byte sc1 polls are uncalibrated, so ratios
between variants are the robust part. usage: summarize_pmc.py r02 TAG needle"""
import csv
import glob
import json
import os
import re
import sys

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
OUT = os.path.join(REPO, "gpurun_out")


def mean_counter(directory, counter, needle):
    vals = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            # `needle` may list alternatives with '|' (kernels with _Float16
            # operands stay mangled in the rocprofv3 output)
            if r["Counter_Name"] == counter and any(
                    part in r["Kernel_Name"] for part in needle.split("|")):
                vals.append(float(r["Counter_Value"]))
    # the first launch of a case is the warm-up; all launches do the same work
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    rnd, tag, needle = sys.argv[1], sys.argv[2], sys.argv[3]
    cases = []
    i = 0
    while True:
        i += 1
        log = os.path.join(OUT, f"pmc_{tag}_{i}_FETCH_SIZE.log")
        if not os.path.exists(log):
            break
        lines = [l for l in open(log).read().splitlines() if "algorithmic_bytes=" in l]
        line = lines[-1] if lines else ""
        fetch, nf = mean_counter(os.path.join(OUT, f"pmc_{tag}_{i}_FETCH_SIZE"), "FETCH_SIZE", needle)
        write, nw = mean_counter(os.path.join(OUT, f"pmc_{tag}_{i}_WRITE_SIZE"), "WRITE_SIZE", needle)
        m = re.search(r"algorithmic_bytes=(\d+)", line)
        alg = float(m.group(1)) if m else None
        entry = {"case": line, "FETCH_SIZE_KiB_mean": fetch, "fetch_dispatches": nf,
                 "WRITE_SIZE_KiB_mean": write, "write_dispatches": nw}
        if fetch is not None and write is not None and alg:
            rd, wr = 2.0 * fetch * 1024.0, write * 1024.0
            entry.update({"read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                          "algorithmic_bytes_per_launch": alg,
                          "traffic_over_algorithmic": (rd + wr) / alg})
        cases.append(entry)
    out = {"correction": "read bytes = 2 * FETCH_SIZE * 1024 (gfx950 wide-stream correction; 8-byte "
                         "sc1 granule polls are uncalibrated), write bytes = WRITE_SIZE * 1024; "
                         "separate --pmc passes", "kernel_needle": needle, "cases": cases}
    path = os.path.join(REPO, "profiles", f"{rnd}_{tag}_pmc.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
