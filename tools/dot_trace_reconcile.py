#!/usr/bin/env python3
"""Reconciles the dot's HIP-event time with its rocprofv3 kernel-trace time
(VERDICT r02 weak #5): ONE process measures the 1 GiB fp16 dot (n = 2^28,
fp32 arithmetic) three ways -
  single   50 launches, each bracketed by its own event pair (min, mean);
  batch    10 batches of 16 back-to-back launches between one event pair,
           per-launch time (min, mean over batches) - what bench.py reports;
  spaced   50 single launches with a device synchronisation and a 2 ms host
           sleep in front of each (what an idle GPU gives the first kernel).
Run it plainly and under `rocprofv3 --kernel-trace --stats` in the same gpurun
call; the trace's per-dispatch durations of the same launches are compared with
these numbers by tools/summarize_dot_reconcile.py. Prints one JSON line."""
import json
import os
import sys
import time

import torch

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(REPO, "accessor-blas_amd", "python"))
import accblas  # noqa: E402

F16, F32 = accblas.F16, accblas.F32
n = 1 << 28
dev = torch.device("cuda:0")
x = torch.rand(n, dtype=torch.float32, device=dev).mul_(2).sub_(1).to(torch.float16)
y = torch.rand(n, dtype=torch.float32, device=dev).mul_(2).sub_(1).to(torch.float16)
res = torch.zeros(1, dtype=torch.float32, device=dev)
h = accblas.Handle()
stream = torch.cuda.current_stream()


def launch():
    accblas.acc_dot(h, F32, F16, F32, n, x, 1, y, 1, res, stream)


def timed(count):
    a, b = accblas.Event(), accblas.Event()
    a.record(stream)
    for _ in range(count):
        launch()
    b.record(stream)
    b.synchronize()
    return a.elapsed_ms(b) * 1e3 / count


for _ in range(5):
    launch()
torch.cuda.synchronize()
single = [timed(1) for _ in range(50)]
batch = [timed(16) for _ in range(10)]
spaced = []
for _ in range(50):
    torch.cuda.synchronize()
    time.sleep(0.002)
    spaced.append(timed(1))
torch.cuda.synchronize()
out = {"n": n, "bytes": 4 * n, "launches": 5 + 50 + 160 + 50}
for name, values in (("single", single), ("batch", batch), ("spaced", spaced)):
    out[name] = {"min_us": min(values), "mean_us": sum(values) / len(values),
                 "max_us": max(values)}
print(json.dumps(out))
h.close()
