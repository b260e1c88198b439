#!/usr/bin/env python3
"""Offline reading of the stamp dump of tools/tune_trsv_wide.hip (argv[4] there):
the components of the chain's hop over ALL row blocks - visibility of x_{b-1}
and x_{b-2}, the on-CU steps between "x_{b-2} observed" and "x_b published" -
as medians / quartiles. usage: analyze_trsv_stamps.py stamps.csv [...]"""
import csv
import sys

import numpy as np


def load(path):
    rows = list(csv.DictReader(open(path)))
    nb = max(int(r["block"]) for r in rows) + 1
    s = np.full((nb, 8, 20), -1.0)
    for r in rows:
        for k in range(20):
            s[int(r["block"]), int(r["wave"]), k] = float(r[f"s{k}"])
    return s


def stats(x):
    return (f"med {np.median(x):5.2f}  p25 {np.percentile(x, 25):5.2f}  p75 "
            f"{np.percentile(x, 75):5.2f}  p90 {np.percentile(x, 90):5.2f}")


for name in sys.argv[1:]:
    s = load(name)
    nb = s.shape[0]
    pub = s[:, :, 9].max(axis=1)
    b = np.arange(8, nb)
    seen, handed = s[b, 0, 6], s[b, :, 7].max(axis=1)
    c_ready, all_far = s[b, :, 5].max(axis=1), s[b, :, 4].max(axis=1)
    reduced, far_last = s[b, :, 3].max(axis=1), s[b, :, 2].max(axis=1)
    observed, moved = s[b, 0, 11], s[b, :, 8].max(axis=1)
    hop = pub[b] - pub[b - 1]
    print(f"{name}: {nb} row blocks, last publication at {pub[-1]:.1f} us")
    for label, x in [
            ("hop P(b) - P(b-1)", hop),
            ("x_{b-1}: seen - P(b-1)", seen - pub[b - 1]),
            ("x_{b-2}: observed - P(b-2)", observed - pub[b - 2]),
            ("observed -> last far panel consumed", far_last - observed),
            ("-> far sums reduced", reduced - far_last),
            ("-> all waves' sums in LDS", all_far - reduced),
            ("-> c ready (slowest wave)", c_ready - all_far),
            ("max(seen, c ready) -> handed over", handed - np.maximum(seen, c_ready)),
            ("-> mat-vec done", moved - handed),
            ("-> published", pub[b] - moved),
            ("x_{b-2} observed -> x_b published", pub[b] - observed)]:
        print(f"  {label:38s} {stats(x)}")
    print(f"  blocks whose c is ready after x_(b-1) is seen: {np.mean(c_ready > seen):.2f}")
    half = len(b) // 2
    print(f"  hop, first half of the blocks: {stats(hop[:half])}")
    print(f"  hop, second half:              {stats(hop[half:])}")
