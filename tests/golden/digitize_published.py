#!/usr/bin/env python3
"""Digitises the error curves the reference publishes as plots
(`/root/reference/plots/{a100,v100}_gemv_error_u-1,1.svg`,
`{a100,v100}_dot_error_median_u-1,1.svg`, embedded at `README.md:61-76`) into
tests/golden/published_bands.json: for every curve the band [min, max] of the
relative error over the plotted sizes, the value at the largest size and the
median. The plots are the only results the reference holds for this path; the
numbers are DATA read off the matplotlib SVG paths (axis ticks -> linear / log
map), accuracy about +-1 %.

Run in the build container (needs /root/reference): python3 tests/golden/digitize_published.py
"""
import json
import math
import os
import re
import statistics

PLOTS = "/root/reference/plots"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "published_bands.json")
FILES = {
    "gemv_error": {"a100": "a100_gemv_error_u-1,1.svg", "v100": "v100_gemv_error_u-1,1.svg"},
    "dot_error_median": {"a100": "a100_dot_error_median_u-1,1.svg",
                         "v100": "v100_dot_error_median_u-1,1.svg"},
}


def groups(svg, prefix):
    """(id, body) of every <g id="prefix_N"> ... element, in document order."""
    out = []
    for m in re.finditer(r'<g id="(%s_\d+)">' % prefix, svg):
        start = m.end()
        depth, pos = 1, start
        while depth:
            nxt = re.compile(r"<g[ >]|</g>").search(svg, pos)
            depth += 1 if nxt.group().startswith("<g") else -1
            pos = nxt.end()
        out.append((m.group(1), svg[start:pos]))
    return out


def ticks(axis_body, log):
    """[(pixel, value)] of the labelled ticks of one axis."""
    out = []
    for _, body in groups(axis_body, "xtick" if not log else "ytick"):
        use = re.search(r'<use [^>]*x="([-\d.]+)"[^>]*y="([-\d.]+)"', body)
        label = re.search(r"<!-- (.*?) -->", body)
        if not use or not label:
            continue
        text = label.group(1)
        if log:
            m = re.search(r"10\^\{(-?\d+)\}", text)
            if not m:
                continue
            out.append((float(use.group(2)), float(m.group(1))))
        else:
            try:
                out.append((float(use.group(1)), float(text.replace("$\\mathdefault{", "")
                                                       .replace("}$", ""))))
            except ValueError:
                # dot plots label the x axis as multiples of 1e8 ("1", "2", ...)
                continue
    return out


def fit(pairs):
    (p0, v0), (p1, v1) = pairs[0], pairs[-1]
    slope = (v1 - v0) / (p1 - p0)
    return lambda p: v0 + (p - p0) * slope


def digitise(path):
    svg = open(path).read()
    axes = dict(groups(svg, "matplotlib.axis"))
    x_of = fit(ticks(axes["matplotlib.axis_1"], log=False))
    y_exp = fit(ticks(axes["matplotlib.axis_2"], log=True))
    legend = svg[svg.index('<g id="legend_1">'):]
    # legend entries: a sample line (stroke colour), its marker, then the label
    labels = {}
    colour = None
    for m in re.finditer(r'<path [^>]*?style="fill:none;stroke:(#[0-9a-f]{6});|<!-- (.*?) -->',
                         legend, re.S):
        if m.group(1):
            colour = m.group(1)
        elif colour is not None:
            labels.setdefault(colour, m.group(2).replace("&lt;", "<").replace("&gt;", ">"))
            colour = None
    curves = {}
    body_end = svg.index('<g id="legend_1">')
    for _, body in groups(svg[:body_end], "line2d"):
        colour = re.search(r"stroke:(#[0-9a-f]{6})", body)
        d = re.search(r'<path [^>]*d="([^"]+)"', body, re.S)
        if not colour or not d or colour.group(1) not in labels:
            continue
        points = [(float(a), float(b)) for a, b in
                  re.findall(r"[ML] ([-\d.]+) ([-\d.]+)", d.group(1))]
        if len(points) < 20:
            continue
        xs = [x_of(p[0]) for p in points]
        ys = [10.0 ** y_exp(p[1]) for p in points]
        curves[labels[colour.group(1)]] = {
            "points": len(points), "x_first": xs[0], "x_last": xs[-1],
            "min": min(ys), "max": max(ys), "median": statistics.median(ys),
            "at_largest_size": ys[-1],
        }
    return curves


def main():
    out = {"source": "digitised from /root/reference/plots/*.svg (README.md:61-76) by "
                     "tests/golden/digitize_published.py; relative errors, +-1 %",
           "metric": {"gemv_error": "sum|ref - res| / sum|ref| vs the fp64 GEMV kernel, "
                                    "n = 100 .. 24500 step 100 (gemv_benchmark.cu:101-124,211-232)",
                      "dot_error_median": "median over 10 re-randomisations of |ref - res| / "
                                          "|ref| vs the fp64 dot kernel (dot_benchmark.cu:175-201,"
                                          "236-283)"}}
    for what, per_gpu in FILES.items():
        out[what] = {gpu: digitise(os.path.join(PLOTS, name)) for gpu, name in per_gpu.items()}
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)
    for what in FILES:
        for gpu, curves in out[what].items():
            for label, c in curves.items():
                print(f"{what:18s} {gpu} {label:32s} n={c['points']:4d} x=[{c['x_first']:.3g},"
                      f"{c['x_last']:.3g}] min={c['min']:.3e} max={c['max']:.3e} "
                      f"median={c['median']:.3e} last={c['at_largest_size']:.3e}")


if __name__ == "__main__":
    main()
