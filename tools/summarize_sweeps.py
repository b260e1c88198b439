#!/usr/bin/env python3
"""Turns the CSV the three benchmark CLIs print (reference format: ms per
variant, or relative error with --error) into what the reference publishes as
plots (`/root/reference/README.md:61-76`; the plot scripts themselves are not in
the reference): GFLOP/s and achieved GB/s per variant, and the error curves.

    python tools/summarize_sweeps.py gpurun_out profiles r01

writes <out>/<tag>_sweep_summary.md (plateau = median over the last 20 % of the
sizes, the convention BASELINE.md uses for the digitised A100/V100 curves) and,
if matplotlib is importable, <out>/<tag>_{gemv,dot,trsv}_{flops,error}.svg.
Conventions: GEMV 2n^2 flop, s(n^2+3n) bytes; dot 2n flop,
2sn bytes; TRSV n^2 flop, s(n(n-1)/2+2n) bytes.
"""
import csv
import os
import statistics
import sys

STORAGE_BYTES = {"fp64": 8, "fp32": 4, "fp16": 2}


def storage_of(name):
    """Bytes per stored element of a column such as 'GEMV Acc<fp64, fp32>'."""
    if "Acc<" in name:
        return STORAGE_BYTES[name.split(",")[1].strip(" >")]
    return STORAGE_BYTES[name.split()[-1]]


def read(path):
    rows = []
    with open(path) as f:
        reader = csv.reader(f, delimiter=";")
        header = next(reader)
        for row in reader:
            if not row or row[0].startswith("---"):
                break
            rows.append([float(v) for v in row])
    return header, rows


def flops_bytes(op, n, s):
    if op == "gemv":
        return 2.0 * n * n, s * (n * n + 3.0 * n)
    if op == "dot":
        return 2.0 * n, 2.0 * s * n
    return n * n, s * (n * (n - 1) / 2.0 + 2.0 * n)


def plateau(values):
    tail = values[-max(1, len(values) // 5):]
    return statistics.median(tail)


def main():
    src, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    lines = [f"# Sweep summary ({tag}): reference-format CLI output on MI355X", "",
             "Plateau = median over the last 20 % of the sweep. Times are min-of-10 device-event",
             "times as printed by the CLIs (`benchmark_function`). Source CSVs: `gpurun_out/sweep_*.csv`.", ""]
    curves = {}
    for op in ("gemv", "dot", "trsv"):
        path = os.path.join(src, f"sweep_{op}.csv")
        if not os.path.exists(path):
            continue
        header, rows = read(path)
        names = [h for h in header[1:] if not h.startswith(("Error", "GFLOP", "GB/s", "%HBM"))]
        sizes = [r[0] for r in rows]
        lines += [f"## {op.upper()} timing sweep: n = {int(sizes[0])} .. {int(sizes[-1])}, "
                  f"{len(sizes)} sizes", "",
                  "| variant | plateau ms | GFLOP/s | GB/s | % of 8 TB/s |", "|---|---|---|---|---|"]
        curves[op] = (sizes, {})
        for i, name in enumerate(names):
            s = storage_of(name)
            gflops, gbs = [], []
            for r in rows:
                fl, by = flops_bytes(op, r[0], s)
                gflops.append(fl / (r[1 + i] * 1e-3) / 1e9)
                gbs.append(by / (r[1 + i] * 1e-3) / 1e9)
            curves[op][1][name] = gflops
            lines.append(f"| {name} | {plateau([r[1 + i] for r in rows]):.4f} | "
                         f"{plateau(gflops):.1f} | {plateau(gbs):.1f} | "
                         f"{plateau(gbs) / 80.0:.1f} |")
        lines.append("")
        if op == "dot":
            errs = [h for h in header if h.startswith("Error")]
            first = header.index(errs[0])
            lines += ["| variant | median rel. error vs DOT fp64 | max |", "|---|---|---|"]
            for j, name in enumerate(errs):
                col = [r[first + j] for r in rows]
                lines.append(f"| {name} | {statistics.median(col):.3e} | {max(col):.3e} |")
            lines.append("")
    for op in ("gemv", "trsv"):
        path = os.path.join(src, f"sweep_{op}_error.csv")
        if not os.path.exists(path):
            continue
        header, rows = read(path)
        lines += [f"## {op.upper()} --error sweep (sum|ref-res| / sum|ref| vs the fp64 kernel)", "",
                  "| variant | first size | last size | max |", "|---|---|---|---|"]
        for i, name in enumerate(header[1:]):
            col = [r[1 + i] for r in rows]
            lines.append(f"| {name} | {col[0]:.3e} | {col[-1]:.3e} | {max(col):.3e} |")
        lines.append("")
        curves[op + "_error"] = ([r[0] for r in rows],
                                 {name: [r[1 + i] for r in rows]
                                  for i, name in enumerate(header[1:])})
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"{tag}_sweep_summary.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        return
    for key, (sizes, series) in curves.items():
        fig, ax = plt.subplots(figsize=(8, 4.5))
        for name, values in series.items():
            if key.endswith("_error"):
                if max(values) > 0:
                    ax.semilogy(sizes, [max(v, 1e-17) for v in values], label=name)
            else:
                ax.plot(sizes, values, label=name)
        ax.set_xlabel("n")
        ax.set_ylabel("relative error" if key.endswith("_error") else "GFLOP/s")
        ax.set_title(f"{key} on MI355X")
        ax.legend(fontsize=7)
        ax.grid(True, alpha=0.3)
        fig.tight_layout()
        suffix = "error" if key.endswith("_error") else "flops"
        fig.savefig(os.path.join(out, f"{tag}_{key.split('_')[0]}_{suffix}.svg"))
        plt.close(fig)


if __name__ == "__main__":
    main()
