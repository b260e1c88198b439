"""GPU tests of the three drop-in benchmark executables: header and column
layout of the reference's CSV (`/root/reference/cuda/gemv_benchmark.cu:196-250`,
`dot_benchmark.cu:146-235`, `trsv_benchmark.cu:184-235`), error columns with the
magnitudes the reference publishes, exact zeros for Acc<T,T> vs
plain (same bits), and the added flags. Run with -m gpu."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
BIN = os.path.join(REPO, "accessor-blas_amd", "bin")


def run(name, *args, timeout=600):
    out = subprocess.run([os.path.join(BIN, name), *args], capture_output=True, text=True,
                         timeout=timeout)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().split("\n")
    return lines[0].split(";"), [line.split(";") for line in lines[1:]]


def test_gemv_benchmark_error_mode():
    header, rows = run("gemv_benchmark", "--error", "--size=1500", "--start=500",
                       "--step=500")
    assert header == ["Num rows", "Error GEMV fp64", "Error GEMV fp32",
                      "Error GEMV Acc<fp64, fp64>", "Error GEMV Acc<fp64, fp32>",
                      "Error GEMV Acc<fp32, fp32>", "Error ROCBLAS GEMV fp64",
                      "Error ROCBLAS GEMV fp32"]
    assert [r[0] for r in rows] == ["+500", "+1000", "+1500"]  # showpos, as the reference
    for r in rows:
        values = [float(v) for v in r[1:]]
        assert values[0] == 0.0 and values[2] == 0.0        # reference itself, Acc<fp64,fp64>
        assert values[1] == values[4]                        # plain fp32 == Acc<fp32,fp32>
        assert 2.5e-8 <= values[3] <= 5.5e-8                 # published: 3.8-4.2e-8
        assert 6e-8 <= values[1] <= 2e-7                     # published: 8.4e-8 .. 1.5e-7
        assert values[5] <= 1e-14 and values[6] <= 1e-6      # rocBLAS columns
        assert r[1] == "+0.0000000000000000e+00"             # precision(16), scientific


def test_gemv_benchmark_timing_mode_with_rates_and_fp16():
    header, rows = run("gemv_benchmark", "--size=4096", "--only=4096", "--rates")
    assert header[:8] == ["Num rows", "GEMV fp64", "GEMV fp32", "GEMV Acc<fp64, fp64>",
                          "GEMV Acc<fp64, fp32>", "GEMV Acc<fp32, fp32>",
                          "ROCBLAS GEMV fp64", "ROCBLAS GEMV fp32"]
    assert len(header) == 8 + 3 * 7 and header[8] == "GFLOP/s GEMV fp64"
    assert len(rows) == 1 and len(rows[0]) == len(header)
    times = [float(v) for v in rows[0][1:8]]
    assert all(0.0 < t < 5.0 for t in times)  # milliseconds
    header16, rows16 = run("gemv_benchmark", "--error", "--size=1024", "--only=1024",
                           "--st=fp16", "--ar=fp32")
    assert "Error GEMV Acc<fp32, fp16>" in header16
    err = float(rows16[0][header16.index("Error GEMV Acc<fp32, fp16>")])
    assert 1e-5 <= err <= 2e-3  # storage-rounding dominated


def test_dot_benchmark_timing_and_detailed_error():
    header, rows = run("dot_benchmark", "--size=3000000")
    assert header[0] == "Vector Size" and header[1] == "DOT fp64"
    assert header[8] == "Error DOT fp64" and len(header) == 15
    assert [r[0] for r in rows] == ["1000000", "3000000"]
    for r in rows:
        errors = [float(v) for v in r[8:]]
        assert errors[0] == 0.0 and errors[2] == 0.0
        assert errors[3] <= 5e-6 and errors[1] <= 1e-4 and errors[5] <= 1e-12
    header, rows = run("dot_benchmark", "--error", "--size=1000000")
    assert header == ["Vector Size"] + ["Error " + n for n in (
        "DOT fp64", "DOT fp32", "DOT Acc<fp64, fp64>", "DOT Acc<fp64, fp32>",
        "DOT Acc<fp32, fp32>", "ROCBLAS DOT fp64", "ROCBLAS DOT fp32")]
    separator = [i for i, r in enumerate(rows) if r[0].startswith("-----")]
    assert len(separator) == 1
    raw = rows[separator[0] + 1:]
    assert raw[0][:2] == ["Random iter", "Vector Size"] and len(raw) == 1 + 10
    # first random iteration = the reference generator's vectors: config C1's value
    assert abs(float(raw[1][2]) - 154.655753144421740) <= 1e-12 * 154.655753144421740


@pytest.mark.parametrize("extra", [[], ["--uplo=lower"], ["--diag=non-unit"]])
def test_trsv_benchmark_error_mode(extra):
    header, rows = run("trsv_benchmark", "--error", "--size=1300", "--start=650",
                       "--step=650", *extra)
    assert header == ["Num rows"] + ["Error " + n for n in (
        "TRSV fp64", "TRSV fp32", "TRSV Acc<fp64, fp64>", "TRSV Acc<fp64, fp32>",
        "TRSV Acc<fp32, fp32>", "ROCBLAS TRSV fp64", "ROCBLAS TRSV fp32")]
    for r in rows:
        values = [float(v) for v in r[1:]]
        assert values[0] == 0.0 and values[2] == 0.0
        assert values[1] == values[4]
        assert values[3] <= values[1] <= 1e-3   # fp64 arithmetic never worse than fp32
        assert values[5] <= 1e-11


def _published():
    import json
    with open(os.path.join(REPO, "tests", "golden", "published_bands.json")) as f:
        return json.load(f)


def test_gemv_error_sweep_falls_inside_the_published_bands():
    """The one check against data the reference itself holds (README.md:61-76):
    the full `gemv_benchmark --error --size=24500` sweep (245 sizes, the
    reference's default) against the error curves digitised from
    plots/{a100,v100}_gemv_error_u-1,1.svg (tests/golden/published_bands.json).
    Acc<fp64, fp32> is storage-rounding dominated (its published curve is the
    same on A100 and V100), so its band, median and large-n value must agree
    with the published ones; the fp32 kernels differ in summation order, so they
    must stay within the band spanned by the published fp32 curves (hand kernel
    and cuBLAS)."""
    bands = _published()["gemv_error"]
    header, rows = run("gemv_benchmark", "--error", "--size=24500")
    assert len(rows) == 245 and rows[-1][0] == "+24500"
    acc = [float(r[header.index("Error GEMV Acc<fp64, fp32>")]) for r in rows]
    f32 = [float(r[header.index("Error GEMV fp32")]) for r in rows]
    pub_acc = bands["a100"]["GEMV Accessor<fp64, fp32>"]
    # the A100 and V100 curves of this variant coincide (storage rounding only)
    assert abs(bands["v100"]["GEMV Accessor<fp64, fp32>"]["min"] / pub_acc["min"] - 1) < 1e-6
    # The published curve is NOT reproduced point by point (n = 100: 4.07e-8
    # published, 3.65e-8 here; n = 400: 4.22e-8 vs 3.97e-8): the random inputs of
    # the published run are not the ones the committed generator draws today,
    # so what can be compared is the band and its centre. The small sizes
    # scatter most (100 x 100 terms), hence the 5 % slack on the band edges.
    import statistics
    assert pub_acc["min"] * 0.95 <= min(acc) and max(acc) <= pub_acc["max"] * 1.05
    assert abs(statistics.median(acc) / pub_acc["median"] - 1) <= 0.02
    assert abs(acc[-1] / pub_acc["at_largest_size"] - 1) <= 0.02
    lo = min(c["min"] for g in bands.values() for k, c in g.items() if k.endswith("fp32")
             and "Accessor" not in k)
    hi = max(c["max"] for g in bands.values() for k, c in g.items() if k.endswith("fp32")
             and "Accessor" not in k)
    assert lo * 0.5 <= min(f32) and max(f32) <= hi * 1.05
    # rocBLAS fp64 vs the fp64 kernel: same order of magnitude as cuBLAS fp64
    roc = [float(r[header.index("Error ROCBLAS GEMV fp64")]) for r in rows]
    assert max(roc) <= 10 * bands["v100"]["GEMV cuBLAS fp64"]["max"]
    # the accuracy half of the reference's headline claim (README.md:67-68,
    # plots/a100_gemv_error_u-1,1.svg at n = 24500: 1.53e-7 vs 3.98e-8): fp64
    # arithmetic on fp32 storage is several times more accurate than pure fp32
    assert f32[-1] >= 2.0 * acc[-1]
    pub_f32 = bands["a100"]["GEMV fp32"]
    assert 0.5 <= (f32[-1] / acc[-1]) / (pub_f32["at_largest_size"] /
                                         pub_acc["at_largest_size"]) <= 2.0


def test_headline_claim_fp64_arithmetic_at_fp32_storage_speed():
    """The speed half of the reference's headline claim (README.md:61-76,
    plots/a100_gemv_flops.svg: 342.4 vs 345.6 vs 178.5 GFLOP/s at the plateau):
    Acc<fp64, fp32> runs at the speed of the fp32 kernel - the stream of fp32
    storage is the bound, fp64 arithmetic is free - and about twice as fast as
    the fp64 kernel. At the reference's own largest size, n = 24500. A
    regression of the mixed-precision path's speed fails here, not only in the
    bench line."""
    header, rows = run("gemv_benchmark", "--only=24500")
    assert rows[0][0] == "+24500"
    t = {name: float(rows[0][header.index(name)])
         for name in ("GEMV fp64", "GEMV fp32", "GEMV Acc<fp64, fp32>")}
    assert abs(t["GEMV Acc<fp64, fp32>"] / t["GEMV fp32"] - 1.0) <= 0.03, t
    assert t["GEMV Acc<fp64, fp32>"] <= 0.55 * t["GEMV fp64"], t
    # and it streams: 4 * (n^2 + 3 n) bytes at >= 75 % of 8 TB/s
    n = 24500
    assert 4.0 * (n * n + 3 * n) / (t["GEMV Acc<fp64, fp32>"] * 1e-3) >= 0.75 * 8e12, t


def test_dot_error_medians_over_the_whole_published_range():
    """The published dot sweep runs from 1e6 to 5.35e8 elements
    (cuda/dot_benchmark.cu:26-27, plots/*_dot_error_median_u-1,1.svg): twelve
    sizes spread over ALL of it (1e6, 4.9e7, ... 5.29e8; ten re-randomisations
    of both 5.35e8-element vectors on the host, a few minutes) against the
    digitised bands."""
    bands = _published()["dot_error_median"]
    header, rows = run("dot_benchmark", "--error", "--size=535000000", "--start=1000000",
                       "--step=48000000", timeout=1500)
    separator = [i for i, r in enumerate(rows) if r[0].startswith("-----")][0]
    medians = rows[:separator]
    assert [int(r[0]) for r in medians] == [1000000 + 48000000 * i for i in range(12)]
    acc = [float(r[header.index("Error DOT Acc<fp64, fp32>")]) for r in medians]
    f32 = [float(r[header.index("Error DOT fp32")]) for r in medians]
    f64acc = [float(r[header.index("Error DOT Acc<fp64, fp64>")]) for r in medians]
    assert all(v == 0.0 for v in f64acc)  # same bits as the reference column
    pub_acc = bands["a100"]["DOT Accessor<fp64, fp32>"]
    assert pub_acc["min"] * 0.5 <= min(acc) and max(acc) <= pub_acc["max"] * 1.5
    hi32 = max(bands[g]["DOT fp32"]["max"] for g in bands)
    assert max(f32) <= hi32 * 1.5
    import statistics
    assert statistics.median(acc) <= statistics.median(f32)
    assert statistics.median(acc) <= 3 * pub_acc["median"]
    roc64 = [float(r[header.index("Error ROCBLAS DOT fp64")]) for r in medians]
    assert max(roc64) <= 10 * max(bands[g]["DOT cuBLAS fp64"]["max"] for g in bands)


def test_dot_error_medians_fall_inside_the_published_bands():
    """`dot_benchmark --error` medians (10 re-randomisations per size) against
    the digitised plots/{a100,v100}_dot_error_median_u-1,1.svg. The published
    sweep goes to 5.35e8 elements; the test runs its first 11 sizes (1e6 .. 2.1e7)
    and requires the medians to stay inside the band of the WHOLE published
    curve (the error of a dot product does not trend with n, it scatters with
    the cancellation of each draw)."""
    bands = _published()["dot_error_median"]
    header, rows = run("dot_benchmark", "--error", "--size=21000000")
    separator = [i for i, r in enumerate(rows) if r[0].startswith("-----")][0]
    medians = rows[:separator]
    assert len(medians) == 11
    acc = [float(r[header.index("Error DOT Acc<fp64, fp32>")]) for r in medians]
    f32 = [float(r[header.index("Error DOT fp32")]) for r in medians]
    pub_acc = bands["a100"]["DOT Accessor<fp64, fp32>"]
    assert pub_acc["min"] * 0.5 <= min(acc) and max(acc) <= pub_acc["max"] * 1.5
    hi32 = max(bands[g]["DOT fp32"]["max"] for g in bands)
    assert max(f32) <= hi32 * 1.5
    # fp64 arithmetic on fp32 storage is at least as accurate as pure fp32 in
    # the median (published: 4.2e-8 vs 1.0-1.3e-6; this build's fp32 kernel sums
    # 512 x 8 partial sums per workgroup and is itself ~10x more accurate than
    # the reference's long per-thread sums, so the gap is smaller here)
    import statistics
    assert statistics.median(acc) <= statistics.median(f32)
    assert statistics.median(acc) <= 3 * pub_acc["median"]
