#!/usr/bin/env python3
"""Headline benchmark: GEMV through range<reduced_row_major<2, fp64, fp32>>.

Workload (BASELINE.json north_star target, config "GEMV 32768x32768,
storage=fp32 arithmetic=fp64"): res = alpha*A*x + beta*res with alpha=beta=1
(`/root/reference/cuda/gemv_benchmark.cu:25-28`), A row-major n x n fp32 in HBM,
arithmetic in fp64, `acc_gemv<double, float>` (`cuda/gemv_kernels.cuh:168-193`).
A "step" is one such GEMV of the WHOLE n x n problem. Inputs are resident in HBM
before the timed region.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Both forms work for N > 1: started WITHOUT a launcher (no WORLD_SIZE in the
environment) the process becomes a parent that touches neither torch nor the
GPU, starts one child per GPU (`launch_ranks`), relays rank 0's JSON line and
exits non-zero if any rank fails or the time limit expires (all children are
killed then). Under torch.distributed.run every process is a rank.

N > 1 ("scaling": "strong"): the SAME n x n problem is split
into row blocks, rank g owns rows shard_bounds(n, N, g, align=64) of A (n/N
rows), a replica of x and its slice of res; the timed step is every rank's
row-block GEMV, which needs NO data-path collective. One process per GPU,
torch.distributed (RCCL) for the barrier, the max-over-ranks time and the
measurements reported beside the headline under "multi_gpu": the kernel-only
time, the end-to-end step without and WITH the all-gather of the result slices
(`multi_gpu.sharded_gemv(gather=True)`), per-GPU GB/s, and the weak-scaling
number (every rank a full n x n block) as a second field.

Prints ONE JSON line on rank 0 (see the keys below). Conventions: flops = 2*n^2
per GEMV, algorithmic bytes = s*(n^2 + 3n), s = sizeof(storage) (SURVEY.md §8d
D3); HBM peak 8 TB/s.
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import threading
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(REPO, "accessor-blas_amd", "python"), os.path.join(REPO, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--n", type=int, default=32768, help="rows per GPU and columns")
    p.add_argument("--st", default="f32", choices=["f16", "f32", "f64"])
    p.add_argument("--ar", default="f64", choices=["f32", "f64"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-other-configs", action="store_true",
                   help="skip the untimed measurements of the other BASELINE configs")
    return p.parse_args()


def _die_with_parent():
    """preexec_fn of a rank: the kernel sends SIGKILL to this child when the
    process that started it dies, however that happens (SIGKILL included, which
    no handler in the parent can catch). Only prctl: no GPU, no exec."""
    import ctypes
    PR_SET_PDEATHSIG = 1
    ctypes.CDLL(None, use_errno=True).prctl(PR_SET_PDEATHSIG, int(signal.SIGKILL), 0, 0, 0)


def launch_ranks(world, argv, command=None, timeout=None, poll=0.05, attempts=3):
    """Parent side of `python bench.py --gpus N` without a launcher: starts
    `world` children of `command` (default: this script) with RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR = 127.0.0.1 and a free MASTER_PORT, each in its own
    process group. Rank 0's stdout is relayed when every rank has exited with
    0. The first non-zero exit, or the time limit (ACCBLAS_BENCH_LAUNCH_TIMEOUT
    seconds, default 1500), kills every child's process group and is the
    return value (124 for the time limit). SIGTERM / SIGHUP / SIGINT to the
    parent (a harness `timeout`, a cancelled job) do the same and exit with
    128 + signal; should the parent be killed outright, every rank gets
    SIGKILL from the kernel (PR_SET_PDEATHSIG), so no rank outlives it on a
    GPU. Rank 0's stderr is forwarded as it comes. The port is found by bind-and-close, so another process can take it
    before rank 0 binds it: a job whose rank 0 reports the address in use is
    started again on a new port (`attempts` times in all). The parent never
    imports torch or touches the GPU, and no process is replaced by another."""
    if command is None:
        command = [sys.executable, os.path.abspath(__file__)]
    if timeout is None:
        timeout = float(os.environ.get("ACCBLAS_BENCH_LAUNCH_TIMEOUT", "1500"))
    deadline = time.monotonic() + timeout
    status = 1
    for attempt in range(attempts):
        status, text, errors = _launch_once(world, argv, command, deadline, timeout, poll)
        in_use = status not in (0, 124) and \
            ("address already in use" in errors.lower() or "EADDRINUSE" in errors)
        # (rank 0's stderr was forwarded as it came, see `_launch_once`)
        if not in_use or attempt + 1 == attempts:
            break
        print("bench.py: the rendezvous port was taken by another process; "
              "starting the ranks again on a new one", file=sys.stderr)
    if status == 0:
        # stdout carries the result line and nothing else: what libraries of a
        # rank wrote there besides ("[Gloo] Rank 0 is connected to ...") goes
        # to stderr
        lines = text.splitlines()
        result = [line for line in lines if line.startswith("{")]
        for line in lines:
            if not line.startswith("{"):
                print(line, file=sys.stderr)
        if result:
            sys.stdout.write(result[-1] + "\n")
        else:
            print("bench.py: rank 0 printed no result line", file=sys.stderr)
            status = 1
        sys.stdout.flush()
    else:
        # the result line counts only when every rank finished; what rank 0
        # had printed before a failure goes to stderr
        sys.stderr.write(text)
    sys.stderr.flush()
    return status


def _launch_once(world, argv, command, deadline, timeout, poll):
    """One attempt of `launch_ranks`: (status, rank 0's stdout, rank 0's stderr)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    children = []
    outputs = []
    # rank 0's stderr: forwarded line by line as it comes (diagnostics stay
    # live, and survive a parent that is killed outright) AND kept, because the
    # retry on a taken port is decided from it
    kept = []
    reader = [None]

    def forward(pipe):
        for line in iter(pipe.readline, b""):
            text = line.decode(errors="replace")
            kept.append(text)
            sys.stderr.write(text)
            sys.stderr.flush()
        pipe.close()

    def kill_all():
        for child in children:
            if child.poll() is None:
                try:
                    os.killpg(child.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
        for child in children:
            child.wait()

    def on_signal(signum, _frame):
        # (handlers run in the main thread between two bytecodes: the list of
        # children is whatever has been started so far)
        kill_all()
        print(f"bench.py: signal {signum}; the ranks were stopped", file=sys.stderr)
        sys.stderr.flush()
        os._exit(128 + signum)

    caught = (signal.SIGTERM, signal.SIGHUP, signal.SIGINT)
    in_main_thread = threading.current_thread() is threading.main_thread()
    previous = {}
    if in_main_thread:
        # before the first child exists: there is no window in which a signal
        # ends the parent and leaves a rank behind
        for signum in caught:
            previous[signum] = signal.signal(signum, on_signal)
    status = 0
    try:
        for rank in range(world):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank),
                       WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            # (this pool's host driver supports dmabuf IPC only: without it RCCL
            # fails with hipIpcGetMemHandle: invalid argument)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            out = tempfile.TemporaryFile(mode="w+")
            outputs.append(out)
            children.append(subprocess.Popen(
                list(command) + list(argv), env=env, stdout=out,
                stderr=subprocess.PIPE if rank == 0 else None,
                start_new_session=True, preexec_fn=_die_with_parent))
            if rank == 0:
                reader[0] = threading.Thread(target=forward, args=(children[0].stderr,),
                                             daemon=True)
                reader[0].start()
        while True:
            codes = [child.poll() for child in children]
            failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if failed:
                rank, code = failed[0]
                print(f"bench.py: rank {rank} of {world} exited with {code}; "
                      "stopping the other ranks", file=sys.stderr)
                status = code if code > 0 else 128 - code
                if rank != 0:
                    # a rank that found a foreign listener on the port can fail
                    # before rank 0 has said "address already in use": give rank
                    # 0 a moment to exit by itself, so that the retry above sees
                    # its message
                    grace = time.monotonic() + 2.0
                    while children[0].poll() is None and time.monotonic() < grace:
                        time.sleep(poll)
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                print(f"bench.py: the {world} ranks did not finish within {timeout:.0f} s; "
                      "stopping them", file=sys.stderr)
                status = 124
                break
            time.sleep(poll)
    finally:
        kill_all()
        for signum, handler in previous.items():
            signal.signal(signum, handler)
    if reader[0] is not None:
        reader[0].join(timeout=5.0)  # (the pipe ends with rank 0, killed or not)
    text = ""
    if outputs:
        outputs[0].seek(0)
        text = outputs[0].read()
        text = text.decode(errors="replace") if isinstance(text, bytes) else text
    for out in outputs:
        out.close()
    return status, text, "".join(kept)


def cpu_baseline(oracle, ar, A_rows, lda, x, res_rows, threads, cpu_reps=12):
    """Times the oracle's host-accessor GEMV on `A_rows` (a bounded sample of the
    same workload): once on 1 thread (first 1/8 of the rows), once on all
    threads. Returns (gflops_all, gflops_one, seconds_total, res_after)."""
    import numpy as np
    rows = res_rows.size
    cols = x.size

    def run(row_begin, row_end, nthreads, res):
        bounds = np.linspace(row_begin, row_end, nthreads + 1).astype(np.int64)
        workers = [threading.Thread(
            target=oracle.accessor_gemv_rows,
            args=(ar, int(bounds[i]), int(bounds[i + 1]), cols, 1.0, A_rows, lda, x, 1, 1.0,
                  res, 1)) for i in range(nthreads)]
        t0 = time.perf_counter()
        for w in workers:
            w.start()
        for w in workers:
            w.join()
        return time.perf_counter() - t0

    res_all = res_rows.copy()
    t_all = min(run(0, rows, threads, res_all.copy()) for _ in range(cpu_reps))
    res_all = res_rows.copy()
    run(0, rows, threads, res_all)  # keep one clean result for the parity check
    one_rows = max(rows // 8, 1)
    t_one = run(0, one_rows, 1, res_rows.copy())
    return (2.0 * rows * cols / t_all / 1e9, 2.0 * one_rows * cols / t_one / 1e9, res_all)


class HostStagedCollectives:
    """torch.distributed look-alike for the gloo rehearsal: device tensors are
    staged through the host for the collective (gloo has no device path for
    them). The product path uses torch.distributed (RCCL) directly."""

    def __init__(self, dist):
        self._dist = dist
        self.ReduceOp = dist.ReduceOp

    def get_world_size(self, group=None):
        return self._dist.get_world_size(group)

    def get_rank(self, group=None):
        return self._dist.get_rank(group)

    def barrier(self):
        self._dist.barrier()

    def all_reduce(self, tensor, op=None, group=None):
        host = tensor.cpu()
        self._dist.all_reduce(host, op=op, group=group)
        tensor.copy_(host)

    def all_gather_into_tensor(self, output, tensor, group=None):
        host_in = tensor.cpu()
        host_out = output.cpu()
        self._dist.all_gather_into_tensor(host_out, host_in, group=group)
        output.copy_(host_out)

    def destroy_process_group(self):
        self._dist.destroy_process_group()


def event_times_us(accblas, torch, fn, stream, reps=5, setup=None, warmups=1):
    """(min, mean) HIP-event time of fn() in microseconds over `reps` runs after
    `warmups` untimed ones. `setup()` (e.g. restoring an in-place operand) is
    enqueued BEFORE the start event, so the bracket holds fn()'s launches and
    nothing else - nothing is subtracted afterwards."""
    for _ in range(warmups):
        if setup is not None:
            setup()
        fn()
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        if setup is not None:
            setup()
        a, b = accblas.Event(), accblas.Event()
        a.record(stream)
        fn()
        b.record(stream)
        b.synchronize()
        times.append(a.elapsed_ms(b) * 1e3)
    return min(times), sum(times) / len(times)


def event_time_us(accblas, torch, fn, stream, reps=5, setup=None):
    """min-of-reps HIP-event time of fn() in microseconds."""
    return event_times_us(accblas, torch, fn, stream, reps, setup)[0]


def batch_times_us(accblas, fn, stream, launches, batches=10):
    """(min, mean) per-launch time over `batches` event-bracketed batches of
    `launches` back-to-back calls of fn()."""
    times = []
    for _ in range(batches):
        a, b = accblas.Event(), accblas.Event()
        a.record(stream)
        for _ in range(launches):
            fn()
        b.record(stream)
        b.synchronize()
        times.append(a.elapsed_ms(b) * 1e3 / launches)
    return min(times), sum(times) / len(times)


def recomputable(kernel, algorithmic_bytes, us_min, us_mean, protocol):
    """The fields every entry of `other_configs` carries so that a reader can
    redo the arithmetic: frac_of_8TBs = algorithmic_bytes / us_min / 8e6."""
    return {"kernel": kernel, "algorithmic_bytes": algorithmic_bytes,
            "us_min": us_min, "us_mean": us_mean, "protocol": protocol,
            "GBs": algorithmic_bytes / us_min / 1e3,
            "frac_of_8TBs": algorithmic_bytes / us_min / 1e3 / HBM_PEAK_GBS,
            "frac_of_8TBs_mean": algorithmic_bytes / us_mean / 1e3 / HBM_PEAK_GBS}


def recorded_traffic(entry):
    """`traffic_over_algorithmic` of one kernel, replayed from profiles/traffic.json
    (the builder's rocprofv3 --pmc passes: 2 * FETCH_SIZE + WRITE_SIZE, separate
    runs) if the kernel's sources still hash to what the counters were taken on;
    otherwise the reason there is none. Never measured in this run."""
    import hashlib
    path = os.path.join(REPO, "profiles", "traffic.json")
    if not os.path.exists(path):
        return {"traffic_over_algorithmic": None, "traffic_source": "none: no profiles/traffic.json"}
    with open(path) as f:
        recorded = json.load(f).get("kernels", {}).get(entry)
    if recorded is None:
        return {"traffic_over_algorithmic": None,
                "traffic_source": f"none: no counters recorded for {entry}"}
    digest = hashlib.sha256()
    for name in recorded["sources"]:
        with open(os.path.join(REPO, "accessor-blas_amd", "csrc", name), "rb") as f:
            digest.update(f.read())
    if digest.hexdigest() != recorded["sources_sha256"]:
        return {"traffic_over_algorithmic": None,
                "traffic_source": f"none: kernel changed ({', '.join(recorded['sources'])}) since "
                                  f"the counters in {recorded['source']} were taken"}
    return {"traffic_over_algorithmic": recorded["traffic_over_algorithmic"],
            "traffic_bytes_per_launch": recorded.get("bytes_per_launch"),
            "traffic_source": f"replayed from {recorded['source']} (rocprofv3 --pmc passes on this "
                              "version of the kernel; not measured in this run)"}


def other_configs(accblas, torch, dev, stream, dist, world, rank, headline=None):
    """The remaining BASELINE.json configurations, measured outside the timed
    region of the headline metric (HIP events around the kernel launches only).
    Every rank runs this (the sharded dot needs all of them); rank 0 reports.
    Every entry carries `kernel`, `algorithmic_bytes`, `us_min`, `us_mean` and
    `protocol` (`recomputable`). `headline` = (A, lda, x, res, st, ar, rows,
    cols) of the headline problem, still resident: the row block one GPU of
    eight would own is timed on it at N = 1."""
    F16, F32, F64 = accblas.F16, accblas.F32, accblas.F64
    out = {}

    def uniform(count, dtype, scale=1.0, seed=1):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        return torch.rand(count, dtype=dtype, device=dev, generator=g) \
            .mul_(2 * scale).sub_(scale)

    # -- configs[1]: GEMV 16384^2 fp64/fp64, plain pointers vs accessor
    n = 16384
    A, x = uniform(n * n, torch.float64, seed=11), uniform(n, torch.float64, seed=12)
    r0 = uniform(n, torch.float64, seed=13)
    r_plain, r_acc = r0.clone(), r0.clone()
    accblas.gemv(F64, n, n, 1.0, A, n, x, 1, 1.0, r_plain, 1, stream)
    accblas.acc_gemv(F64, F64, n, n, 1.0, A, n, x, 1, 1.0, r_acc, 1, stream)
    torch.cuda.synchronize()
    identical = bool(torch.equal(r_plain, r_acc))
    t_plain, t_plain_mean = event_times_us(accblas, torch, lambda: accblas.gemv(
        F64, n, n, 1.0, A, n, x, 1, 1.0, r_plain, 1, stream), stream, reps=10)
    t_acc, t_acc_mean = event_times_us(accblas, torch, lambda: accblas.acc_gemv(
        F64, F64, n, n, 1.0, A, n, x, 1, 1.0, r_acc, 1, stream), stream, reps=10)
    byts = 8.0 * (n * n + 3 * n)
    out["gemv_16384_fp64_fp64"] = dict(
        recomputable("accblas::dev::kernel::gemv_vec (Acc<fp64, fp64> through the accessor)",
                     byts, t_acc, t_acc_mean,
                     "warm-up, then min / mean of 10 event-timed launches"),
        plain_us=t_plain, plain_us_mean=t_plain_mean, acc_us=t_acc,
        plain_GBs=byts / t_plain / 1e3, acc_GBs=byts / t_acc / 1e3,
        acc_frac_of_8TBs=byts / t_acc / 1e3 / HBM_PEAK_GBS,
        plain_equals_accessor_bitwise=identical,
        **recorded_traffic("gemv_16384_fp64_fp64"))
    del A, x, r0, r_plain, r_acc

    # -- the shape ONE GPU of an 8-GPU run of the headline problem sees: a row
    # block of rows/8 rows of the SAME matrix (4096 x 32768 fp32 = 512 MiB, 67 us
    # of HBM time). Rotating over all 8 row blocks (4 GiB between two uses of a
    # line) keeps the Infinity Cache out of it.
    if world == 1 and headline is not None:
        hA, hlda, hx, hres, hst, har, hrows, hcols = headline
        share = hrows // 8
        blocks = hrows // share
        sbytes = accblas.DTYPE_BYTES[hst]
        turn_g = [0]

        def block_gemv(rotate=True):
            k = turn_g[0] % blocks if rotate else 0
            turn_g[0] += 1
            lo = k * share
            accblas.acc_gemv(har, hst, share, hcols, 1.0, hA[lo * hlda:], hlda, hx, 1, 1.0,
                             hres[lo:], 1, stream)

        t_one, t_one_mean = event_times_us(accblas, torch, block_gemv, stream, reps=3 * blocks)
        t_same = event_time_us(accblas, torch, lambda: block_gemv(False), stream, reps=10)
        block_gemv()
        torch.cuda.synchronize()
        t_b2b, t_b2b_mean = batch_times_us(accblas, block_gemv, stream, 3 * blocks)
        byts = sbytes * (share * hcols + 2.0 * share + hcols)
        out[f"gemv_rowblock_{share}x{hcols}_one_gpu_share_of_8"] = dict(
            recomputable("accblas::dev::kernel::gemv_vec", byts, t_one, t_one_mean,
                         f"row blocks k = 0..{blocks - 1} of the headline matrix in turn "
                         f"(HBM-sourced); us_min / us_mean: {3 * blocks} event-timed single "
                         "launches"),
            rows=share, cols=hcols,
            us_back_to_back=t_b2b, us_back_to_back_mean_of_10_batches=t_b2b_mean,
            frac_of_8TBs_back_to_back=byts / t_b2b / 1e3 / HBM_PEAK_GBS,
            us_same_row_block=t_same,
            **recorded_traffic("gemv_rowblock_4096x32768"))

    # -- configs[3]: dot n = 2^28, storage fp16, arithmetic fp32, chunked over the
    # ranks, one all-reduce of a single fp32 element
    from accblas import multi_gpu
    n_total = 1 << 28
    align = 16 // 2  # chunk starts stay 16-byte aligned for the vector loads
    lo, hi = multi_gpu.shard_bounds(n_total, world, rank, align)
    count = hi - lo
    # A chunk of 2^25 fp16 pairs (the 8-GPU share) is 134 MB: it stays in the
    # 256 MiB Infinity Cache between repetitions. `copies` buffers, swept in
    # turn, keep more than 512 MiB between two uses of the same line, so the
    # rotating time is the HBM one; both are reported.
    copies = max(1, -(-(640 << 20) // (4 * count)))
    x64 = uniform(count, torch.float64, seed=100 + rank)
    y64 = uniform(count, torch.float64, seed=200 + rank)
    xh = torch.empty(copies * count, dtype=torch.float16, device=dev)
    yh = torch.empty(copies * count, dtype=torch.float16, device=dev)
    accblas.convert(F16, xh, F64, x64, count, stream)
    accblas.convert(F16, yh, F64, y64, count, stream)
    for c in range(1, copies):
        xh[c * count:(c + 1) * count].copy_(xh[:count])
        yh[c * count:(c + 1) * count].copy_(yh[:count])
    handle = accblas.Handle()
    partial = torch.zeros(1, dtype=torch.float32, device=dev)
    ref = torch.zeros(1, dtype=torch.float64, device=dev)
    accblas.dot(handle, F64, count, x64, 1, y64, 1, ref, stream)  # fp64 reference
    turn = [0]

    def local_dot(_lo, _hi, rotate=True):
        c = turn[0] % copies if rotate else 0
        turn[0] += 1
        accblas.acc_dot(handle, F32, F16, F32, count, xh[c * count:], 1, yh[c * count:], 1,
                        partial, stream)
        return partial

    t_resident = event_time_us(accblas, torch, lambda: local_dot(lo, hi, rotate=False), stream)
    reps = 4 * copies
    torch.cuda.synchronize()
    # back-to-back launches over the rotating buffers, per-launch average; the
    # minimum over 10 such batches (the reference's min-of-10 protocol)
    t_kernel, t_kernel_mean = batch_times_us(accblas, lambda: local_dot(lo, hi), stream, reps)
    multi_gpu.sharded_dot(local_dot, n_total, dist=dist, align=align)  # warm-up incl. RCCL
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        total = multi_gpu.sharded_dot(local_dot, n_total, dist=dist, align=align)
    torch.cuda.synchronize()
    t_e2e = (time.perf_counter() - t0) / reps * 1e6
    got = float(total.item())
    if dist is not None:
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        t = torch.tensor([t_kernel, t_e2e, t_resident], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_kernel, t_e2e, t_resident = float(t[0]), float(t[1]), float(t[2])
    want = float(ref.item())
    out["dot_2p28_fp16_fp32"] = {
        "kernel": "accblas::dev::kernel::dot (Acc<fp32, fp16>, deterministic hand-off)",
        "algorithmic_bytes": 4.0 * count, "us_min": t_kernel, "us_mean": t_kernel_mean,
        "frac_of_8TBs": 4.0 * count / t_kernel / 1e3 / HBM_PEAK_GBS,
        "frac_of_8TBs_mean": 4.0 * count / t_kernel_mean / 1e3 / HBM_PEAK_GBS,
        **recorded_traffic("dot_2p28_fp16_fp32"),
        "protocol": f"min / mean over 10 batches of {reps} back-to-back launches sweeping "
                    f"{copies} buffer sets in turn (per-launch average)",
        "elements_per_gpu": count, "kernel_us": t_kernel,
        "kernel_GBs_per_gpu": 4.0 * count / t_kernel / 1e3,
        "kernel_frac_of_8TBs": 4.0 * count / t_kernel / 1e3 / HBM_PEAK_GBS,
        "kernel_us_mean_of_10_batches": t_kernel_mean,
        "kernel_us_same_buffers": t_resident,
        "kernel_GBs_per_gpu_same_buffers": 4.0 * count / t_resident / 1e3,
        "rotating_buffer_sets": copies,
        "end_to_end_us_incl_allreduce": t_e2e,
        "aggregate_GBs_end_to_end": 4.0 * n_total / t_e2e / 1e3,
        "collective": "all_reduce(sum) of 1 fp32 element" if world > 1 else "none",
        "world_size_seen_by_backend": dist.get_world_size() if dist is not None else 1,
        "rel_err_vs_fp64": abs(got - want) / abs(want)}
    handle.close()
    del x64, y64, xh, yh

    # -- the shape ONE GPU of an 8-GPU run of configs[3] sees: 2^25 fp16 pairs
    # (128 MiB). On one GPU it is reported beside the full problem: swept over
    # 6 buffer pairs (768 MiB, three times the Infinity Cache) = sourced from
    # HBM, and on one pair re-read (cache-resident).
    if world == 1:
        chunk, pairs = 1 << 25, 6
        g = torch.Generator(device=dev)
        g.manual_seed(300)
        xc = torch.rand(pairs * chunk, dtype=torch.float32, device=dev, generator=g) \
            .mul_(2).sub_(1).to(torch.float16)
        yc = torch.rand(pairs * chunk, dtype=torch.float32, device=dev, generator=g) \
            .mul_(2).sub_(1).to(torch.float16)
        hc = accblas.Handle()
        rc = torch.zeros(1, dtype=torch.float32, device=dev)
        turn_c = [0]

        def chunk_dot(rotate=True):
            c = turn_c[0] % pairs if rotate else 0
            turn_c[0] += 1
            accblas.acc_dot(hc, F32, F16, F32, chunk, xc[c * chunk:], 1, yc[c * chunk:], 1,
                            rc, stream)

        t_same = event_time_us(accblas, torch, lambda: chunk_dot(False), stream, reps=10)
        t_single = event_time_us(accblas, torch, chunk_dot, stream, reps=4 * pairs)
        t_hbm, t_hbm_mean = batch_times_us(accblas, chunk_dot, stream, 4 * pairs)
        out["dot_chunk_2p25_fp16_fp32_one_gpu_share_of_8"] = {
            "kernel": "accblas::dev::kernel::dot (Acc<fp32, fp16>, deterministic hand-off)",
            "algorithmic_bytes": 4.0 * chunk, "us_min": t_hbm, "us_mean": t_hbm_mean,
            "frac_of_8TBs": 4.0 * chunk / t_hbm / 1e3 / HBM_PEAK_GBS,
            "frac_of_8TBs_mean": 4.0 * chunk / t_hbm_mean / 1e3 / HBM_PEAK_GBS,
            **recorded_traffic("dot_chunk_2p25_fp16_fp32"),
            "elements": chunk, "bytes": 4 * chunk,
            "hbm_us_back_to_back": t_hbm,
            "hbm_GBs": 4.0 * chunk / t_hbm / 1e3,
            "hbm_frac_of_8TBs": 4.0 * chunk / t_hbm / 1e3 / HBM_PEAK_GBS,
            "hbm_us_single_launch": t_single,
            "same_buffers_us_single_launch": t_same,
            "rotating_buffer_pairs": pairs,
            "protocol": "min over 10 batches of 24 back-to-back launches sweeping the "
                        "pairs in turn (per-launch average); single launches: min of "
                        "event-timed launches, each on the next pair / on one pair"}
        hc.close()
        del xc, yc

    # -- configs[4]: TRSV n = 65536, lower, unit diagonal, storage fp32,
    # arithmetic fp64, and the reference CLI's default (upper / unit, n = 24000,
    # cuda/trsv_benchmark.cu:26-30); does not shard (replicas only): rank 0 only
    if rank == 0:
        # (the third: the reference's plain `trsv<float>` column on the same problem)
        for n, uplo, uplo_name, ar in ((65536, accblas.LOWER, "lower", F64),
                                       (24000, accblas.UPPER, "upper", F64),
                                       (24000, accblas.UPPER, "upper", F32)):
            L = uniform(n * n, torch.float32, scale=1.0 / n ** 0.5, seed=31)
            b = uniform(n, torch.float32, seed=32)
            xs = b.clone()
            helper = torch.zeros(2, dtype=torch.int32, device=dev)

            def solve():
                if ar == F64:
                    accblas.acc_trsv(F64, F32, uplo, accblas.UNIT, n, L, n, xs, 1, helper, stream)
                else:
                    accblas.trsv(F32, uplo, accblas.UNIT, n, L, n, xs, 1, helper, stream)
            # the reference's protocol (warm-up, then the minimum of 10 runs,
            # cuda/utils.cuh:236-262) with x restored BEFORE the start event:
            # the bracket holds the solve's one launch and nothing else
            # (five warm-ups: the first solves after another workload run 3-8 %
            # slower - profiles/r05_trsv_spread_before.txt, "seq" -, twenty timed)
            t_trsv, t_trsv_mean = event_times_us(accblas, torch, solve, stream, reps=20,
                                                 setup=lambda: xs.copy_(b), warmups=5)
            xs.copy_(b)
            solve()
            torch.cuda.synchronize()
            # residual of 256 sampled rows in fp64: (T x)_i + x_i - b_i
            rows = torch.randint(0, n, (256,), device=dev)
            Lr = L.view(n, n)[rows].double()
            cols = torch.arange(n, device=dev)
            keep = cols[None, :] < rows[:, None] if uplo == accblas.LOWER \
                else cols[None, :] > rows[:, None]
            Lr = torch.where(keep, Lr, torch.zeros_like(Lr))
            resid = Lr @ xs.double() + xs.double()[rows] - b.double()[rows]
            byts = 4.0 * (n * (n - 1) / 2 + 2 * n)
            out[f"trsv_{n}_{uplo_name}_unit_fp32_{'fp64' if ar == F64 else 'plain_fp32'}"] = dict(
                recomputable("accblas::dev::kernel::trsv_wide", byts, t_trsv, t_trsv_mean,
                             "x restored by a device copy enqueued before the start event; "
                             "5 warm-ups, then min / mean of 20 event-timed solves (one launch "
                             "each)"),
                **recorded_traffic(f"trsv_{n}_{uplo_name}_{'fp64' if ar == F64 else 'fp32'}_fp32"),
                us=t_trsv, us_per_128row_hop=t_trsv / (n / 128),
                sampled_residual_rel=float(resid.abs().sum() / b.double()[rows].abs().sum()),
                helper_word=int(helper[0].item()) & 0xffffffff)
            del L, Lr
    torch.cuda.empty_cache()
    return out


def timed_loop(step, steps, warmup, barrier, record=None):
    """`warmup` untimed steps, then exactly `steps` steps bracketed by
    barrier() on both sides; `record(which)` (0 = start, 1 = stop) marks the
    stream inside the bracket. Returns the wall-clock seconds of the bracket."""
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    if record is not None:
        record(0)
    for _ in range(steps):
        step()
    if record is not None:
        record(1)
    barrier()
    return time.perf_counter() - t0


GEMV_ROW_ALIGN = 64  # row blocks start at multiples of 64 rows (same bits as alone)


def strong_gemv(local_gemv, rows, dist, steps, warmup, barrier, record=None,
                gather_into=None):
    """The N > 1 step of this benchmark: the fixed
    rows x cols problem split into contiguous row blocks, rank g computing
    rows shard_bounds(rows, N, g, align=64) with `local_gemv(lo, hi)` (which
    returns its slice of the result). Without `gather_into` there is no
    collective on the data path; with it every step ends in ONE all-gather of
    the slices into that tensor. Returns (lo, hi, seconds of the timed loop).
    Used by main() on the HIP kernels and by tests/test_distributed.py on the
    oracle (gloo)."""
    from accblas import multi_gpu
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    lo, hi = multi_gpu.shard_bounds(rows, world, rank, GEMV_ROW_ALIGN)

    def step():
        multi_gpu.sharded_gemv(local_gemv, rows, dist=dist, gather=gather_into is not None,
                               res_full=gather_into, align=GEMV_ROW_ALIGN)

    return lo, hi, timed_loop(step, steps, warmup, barrier, record)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become the parent of one child per GPU (before torch,
        # accblas or anything else that could touch the GPU is imported)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.gpus > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (as in launch_ranks)
    import numpy as np
    import torch

    import accblas

    accblas.load()  # fails loudly if the HIP library is missing; no fallback

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    # one process per GPU; ACCBLAS_BENCH_SHARE_GPU=1 (rehearsal on a 1-GPU box
    # only) lets several ranks share device 0 and use gloo, because RCCL needs
    # one device per rank
    share_gpu = os.environ.get("ACCBLAS_BENCH_SHARE_GPU") == "1"
    device_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    dist = None
    backend = "none"
    if world > 1:
        import torch.distributed as torch_dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            torch_dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist = HostStagedCollectives(torch_dist)
            backend = "gloo (host-staged rehearsal, ranks share one GPU)"
        else:
            torch_dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                          device_id=dev)
            dist = torch_dist
            backend = "nccl (RCCL)"

    codes = {"f16": accblas.F16, "f32": accblas.F32, "f64": accblas.F64}
    st, ar = codes[args.st], codes[args.ar]
    st_bytes = accblas.DTYPE_BYTES[st]
    n = args.n
    rows, cols, lda = n, n, n

    # ---- synthetic inputs, generated on the device (U(-1,1) fp64 master, then
    # the reference's static_cast to the storage type, gemv_memory.cuh:92-97).
    # Every rank draws the SAME n x n problem (same seeds) and keeps its rows.
    stream = torch.cuda.current_stream()

    def uniform(count, seed):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        return torch.rand(count, dtype=torch.float64, device=dev, generator=g) \
            .mul_(2.0).sub_(1.0)

    def to_storage(master):
        out = torch.empty(master.numel(), dtype=accblas.torch_dtype(st), device=dev)
        accblas.convert(st, out, accblas.F64, master, master.numel(), stream)
        return out

    x64 = uniform(cols, 7)
    res64 = uniform(rows, 43)
    A64 = uniform(rows * lda, 42)
    A_st, x_st, res0_st = to_storage(A64), to_storage(x64), to_storage(res64)
    torch.cuda.synchronize()

    # ---- accuracy, untimed: the reference's --error metric against the GPU fp64
    # plain kernel on the unrounded master data (gemv_benchmark.cu:219-232)
    ref = res64.clone()
    accblas.gemv(accblas.F64, rows, cols, 1.0, A64, lda, x64, 1, 1.0, ref, 1, stream)
    res_st = res0_st.clone()
    accblas.acc_gemv(ar, st, rows, cols, 1.0, A_st, lda, x_st, 1, 1.0, res_st, 1, stream)
    torch.cuda.synchronize()
    first_result = res_st.clone()
    # (the master's 8.6 GB stay in torch's allocator: device memory handed back to
    # the driver is cleared in the background, and launches that run meanwhile
    # stream 2-5 % slower for seconds - profiles/r03_gemv_data_dependence.txt)
    del A64

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def local_gemv(lo, hi):
        """rows [lo, hi) of the problem, in place in this rank's slice of res."""
        accblas.acc_gemv(ar, st, hi - lo, cols, 1.0, A_st[lo * lda:], lda, x_st, 1, 1.0,
                         res_st[lo:], 1, stream)
        return res_st[lo:hi]

    events = [accblas.Event(), accblas.Event()]

    def record(which):
        events[which].record(stream)

    def reduce_max(values):
        if dist is None:
            return values
        t = torch.tensor(values, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    # ---- the timed region: K steps of the whole problem, row blocks over ranks
    lo, hi, elapsed = strong_gemv(local_gemv, rows, dist, args.steps, args.warmup, barrier,
                                  record)
    local_rows = hi - lo
    kernel_ms = events[0].elapsed_ms(events[1]) / args.steps
    elapsed, kernel_ms = reduce_max([elapsed, kernel_ms])

    multi = None
    if world > 1:
        # the row block result equals the rows of the single-GPU product bit for bit
        res_check = res0_st.clone()
        accblas.acc_gemv(ar, st, local_rows, cols, 1.0, A_st[lo * lda:], lda, x_st, 1, 1.0,
                         res_check[lo:], 1, stream)
        torch.cuda.synchronize()
        same_bits = bool(torch.equal(res_check[lo:hi], first_result[lo:hi]))
        # end-to-end step WITH the all-gather of the result slices
        res_full = torch.empty(rows, dtype=accblas.torch_dtype(st), device=dev)
        gather_ok = rows % (world * GEMV_ROW_ALIGN) == 0
        t_gather = None
        if gather_ok:
            _, _, t_gather = strong_gemv(local_gemv, rows, dist, args.steps, args.warmup,
                                         barrier, gather_into=res_full)
        # weak scaling as a second field: every rank a full rows x cols block
        def weak_step():
            accblas.acc_gemv(ar, st, rows, cols, 1.0, A_st, lda, x_st, 1, 1.0, res_st, 1,
                             stream)
        t_weak = timed_loop(weak_step, args.steps, args.warmup, barrier)
        t_gather, t_weak = reduce_max([t_gather or 0.0, t_weak])
        local_bytes = st_bytes * (local_rows * cols + 2.0 * local_rows + cols)
        multi = {
            "mode": "strong: the rows x cols problem split into row blocks",
            "backend": backend,
            "world_size_seen_by_backend": dist.get_world_size(),
            "rows_per_gpu": local_rows,
            "kernel_us": kernel_ms * 1e3,
            "per_gpu_GBs": local_bytes / (kernel_ms * 1e-3) / 1e9,
            "per_gpu_frac_of_8TBs": local_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "end_to_end_us_no_gather": elapsed / args.steps * 1e6,
            "end_to_end_us_with_allgather":
                t_gather / args.steps * 1e6 if gather_ok else None,
            "row_block_equals_single_gpu_rows_bitwise": same_bits,
            "weak": {"rows_per_gpu": rows, "ms_per_step": t_weak / args.steps * 1e3,
                     "value_GFLOPs": 2.0 * rows * cols * world * args.steps / t_weak / 1e9},
        }

    # CPU-baseline sample: half of the rows (2 GiB of fp32 on the host), swept
    # `cpu_reps` times by all threads: about 10 core-seconds of CPU work
    sample_rows = min(rows, 16384)
    want_cpu = world == 1 and not args.no_cpu_baseline
    A_sample = A_st[: sample_rows * lda].cpu().numpy() if want_cpu else None
    extras = None
    if not args.no_other_configs and n == 32768:
        extras = other_configs(accblas, torch, dev, stream, dist, world, rank,
                               headline=(A_st, lda, x_st, res_st, st, ar, rows, cols))
    del A_st  # (stays in torch's allocator, see above)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    flops_per_step = 2.0 * rows * cols
    bytes_per_launch = st_bytes * (local_rows * cols + 2.0 * local_rows + cols)
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
    out = {
        "metric": "GEMV GFLOP/s + achieved HBM GB/s (% of peak) per (st_t,ar_t); "
                  "dot rel-error",
        "value": flops_per_step * args.steps / elapsed / 1e9,
        "unit": "GFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": args.ar,
        "data": "synthetic: U(-1,1) fp64 master drawn on the device (torch Philox), "
                "static_cast to the storage type; alpha=beta=1",
        "config": {
            "workload": f"GEMV {rows}x{cols}, storage={args.st} arithmetic={args.ar}, "
                        "acc_gemv through range<reduced_row_major<2,ar,st>> (BASELINE "
                        "configs[2], the north_star target)"
                        + (f", split into {world} row blocks of {local_rows} rows"
                           if world > 1 else ""),
            "rows": rows, "cols": cols, "lda": lda, "rows_per_gpu": local_rows,
            "sharding": "row blocks, x replicated, no collective in the timed step"
                        if world > 1 else "none",
            "flops_per_gemv": "2*rows*cols",
            "bytes_per_gemv": "s*(rows*cols + 2*rows + cols)",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "accblas::dev::kernel::gemv_vec",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "kernel_us": kernel_ms * 1e3,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "achieved_gflops_per_gpu": 2.0 * local_rows * cols / (kernel_ms * 1e-3) / 1e9,
        },
    }
    if multi is not None:
        out["multi_gpu"] = multi
    if n == 32768 and args.st == "f32" and world == 1:
        replay = recorded_traffic("gemv_32768_fp64_fp32")
        out["roofline"]["traffic"] = replay.get("traffic_bytes_per_launch")
        out["roofline"]["traffic_source"] = replay["traffic_source"]

    # ---- accuracy on rank 0: the reference's metric sum|ref - res| / sum|ref|
    # (gemv_benchmark.cu:101-124) against the fp64 kernel on the unrounded data
    ref_h = ref.cpu().numpy()
    got_h = first_result.cpu().numpy()
    out["rel_err_vs_fp64"] = float(np.abs(ref_h - got_h.astype(np.float64)).sum() /
                                   np.abs(ref_h).sum())
    out["rel_err_inputs"] = ("torch Philox U(-1,1), NOT the reference generator "
                             "(minstd_rand0(42), which only the CLIs use): compare the "
                             "magnitude with the published curve, not point for point")
    if world == 1 and not args.no_cpu_baseline:
        # ---- CPU baseline leg: the only place the oracle is used (as the
        # baseline and as the checker of a sample of the GPU result)
        import oracle_lib as oracle
        # the GPU box gives one-GPU jobs a share of 16 host cores
        threads = min(len(os.sched_getaffinity(0)), 16)
        A_rows = A_sample
        x_h = x_st.cpu().numpy()
        res_rows = res0_st[:sample_rows].cpu().numpy()
        g_all, g_one, res_cpu = cpu_baseline(oracle, ar, A_rows, lda, x_h, res_rows, threads)
        out["cpu_baseline"] = {
            "value": g_all,
            "unit": "GFLOP/s",
            "cores": threads,
            "kind": "port",
            "single_core_value": g_one,
            "sample": f"oracle host-accessor GEMV (same kernel maths, Ar={args.ar} "
                      f"St={args.st}) on the first {sample_rows} of {rows} rows of the "
                      f"same matrix, {threads} threads by row blocks, min of 12 sweeps "
                      f"(~10 core-seconds in all); single_core_value = 1 thread on "
                      f"{max(sample_rows // 8, 1)} rows",
        }
        # the reference's own CPU path: compare + pairwise reduce on the result
        t0 = time.perf_counter()
        for _ in range(20):
            oracle.compare(ref_h, got_h)
        out["cpu_baseline"]["reference_error_check_us"] = \
            (time.perf_counter() - t0) / 20 * 1e6
        out["parity_sample_rel_err_vs_oracle"] = oracle.rel_error(
            res_cpu.astype(np.float64), got_h[:sample_rows].astype(np.float64))
    if extras is not None:
        out["other_configs"] = extras
    # which binary these numbers belong to
    lib_path, lib_sha = accblas.library_identity()
    out["library"] = {"path": os.path.relpath(lib_path, REPO), "sha256": lib_sha}
    hash_file = os.path.join(os.path.dirname(lib_path), "SOURCE_HASH")
    if os.path.exists(hash_file):
        with open(hash_file) as f:
            out["library"]["source_hash"] = f.read().strip()
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
