#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs that tools/gpu_profile.sh left under gpurun_out/
into the tracked summaries under profiles/ (round tag as argv[1], e.g. r01)."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
OUT = os.path.join(REPO, "gpurun_out")
PROF = os.path.join(REPO, "profiles")


def counter_mean(directory, counter, needle):
    path = os.path.join(OUT, directory, "gemv_counter_collection.csv")
    values = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
              if r["Counter_Name"] == counter and needle in r["Kernel_Name"]
              and "reduced_row_major" in r["Kernel_Name"]]
    return sum(values) / len(values), len(values)


def kernel_source_hash():
    """SHA-256 of the GEMV kernel source the counters were taken with."""
    import hashlib
    path = os.path.join(REPO, "accessor-blas_amd", "csrc", "gemv_kernels.hpp")
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(PROF, exist_ok=True)
    shutil.copy(os.path.join(OUT, "prof_stats", "gemv_kernel_stats.csv"),
                os.path.join(PROF, f"{tag}_gemv_kernel_stats.csv"))
    shutil.copy(os.path.join(OUT, "bench.json"), os.path.join(PROF, f"{tag}_bench.json"))
    # the line the SAME command printed under the kernel trace: its HIP-event
    # figures belong beside the trace's own durations (profiles/<tag>_summary.md)
    traced_log = os.path.join(OUT, "prof_stats.log")
    if os.path.exists(traced_log):
        lines = [line for line in open(traced_log) if line.startswith("{")]
        if lines:
            with open(os.path.join(PROF, f"{tag}_bench_under_rocprofv3.json"), "w") as f:
                f.write(lines[-1])
    fetch_kb, n_fetch = counter_mean("prof_fetch", "FETCH_SIZE", "gemv_vec")
    write_kb, n_write = counter_mean("prof_write", "WRITE_SIZE", "gemv_vec")
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
    # FETCH_SIZE reports exactly half of the bytes of a 16-B/lane coalesced
    # streaming read -> double it. WRITE_SIZE is taken as is.
    read_bytes = 2.0 * fetch_kb * 1024.0
    write_bytes = write_kb * 1024.0
    bench = json.load(open(os.path.join(OUT, "bench.json")))
    algorithmic = bench["roofline"]["algorithmic_bytes_per_launch"]
    summary = {
        "kernel": "accblas::dev::kernel::gemv_vec (Acc<fp64, fp32>, 32768x32768)",
        "FETCH_SIZE_KiB_mean": fetch_kb, "fetch_dispatches": n_fetch,
        "WRITE_SIZE_KiB_mean": write_kb, "write_dispatches": n_write,
        "correction": "read bytes = 2 * FETCH_SIZE * 1024 (gfx950 wide-stream "
                      "correction), write bytes = WRITE_SIZE * 1024; separate --pmc passes",
        "gemv_vec_bytes_per_launch": read_bytes + write_bytes,
        "algorithmic_bytes_per_launch": algorithmic,
        "traffic_over_algorithmic": (read_bytes + write_bytes) / algorithmic,
    }
    with open(os.path.join(PROF, f"{tag}_gemv_pmc.json"), "w") as f:
        json.dump(summary, f, indent=1)
    # the counters belong to ONE version of the kernel: profiles/traffic.json ties
    # them to the SHA-256 of its sources, bench.py drops the figure when they changed
    import subprocess
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "update_traffic.py"),
                    f"gemv_32768_fp64_fp32=profiles/{tag}_gemv_pmc.json:"], check=True,
                   stdout=subprocess.DEVNULL)
    # One kernel name can cover launches of different shapes (the headline GEMV
    # and the row block one GPU of eight would own run the same instantiation):
    # the trace grouped by kernel AND grid, so that every entry of the bench's
    # `other_configs` can be looked up on its own
    trace = os.path.join(OUT, "prof_stats", "gemv_kernel_trace.csv")
    if os.path.exists(trace):
        groups = defaultdict(list)
        for r in csv.DictReader(open(trace)):
            name = r["Kernel_Name"]
            if "accblas" not in name:
                continue
            key = (name, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))
            groups[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        with open(os.path.join(PROF, f"{tag}_gemv_kernel_stats_by_grid.csv"), "w") as f:
            out = csv.writer(f)
            out.writerow(["Name", "Grid_Size_X", "Workgroup_Size_X", "Calls", "AverageNs",
                          "MinNs", "MaxNs"])
            for (name, grid, block), ns in sorted(groups.items(),
                                                  key=lambda kv: -sum(kv[1])):
                out.writerow([name, grid, block, len(ns), f"{sum(ns) / len(ns):.1f}", min(ns),
                              max(ns)])
    for name in ("dot", "trsv"):
        src = os.path.join(OUT, f"prof_{name}", f"{name}_kernel_stats.csv")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(PROF, f"{tag}_{name}_kernel_stats.csv"))
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
