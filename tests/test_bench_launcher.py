"""`python bench.py --gpus N` without a launcher (bench.launch_ranks): the
parent starts one child per GPU, relays rank 0's line when every rank succeeds,
and exits non-zero — with every other rank killed — when one rank fails or the
time limit expires. The children here are small scripts (no GPU); the last test
runs the real bench.py on this GPU-less box and expects a loud failure."""
import os
import subprocess
import sys
import textwrap
import time

import bench

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path, body):
    path = tmp_path / "child.py"
    path.write_text(textwrap.dedent(body))
    return [sys.executable, str(path)]


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    # a zombie still answers kill(0); it is not running any more
    with open(f"/proc/{pid}/stat") as f:
        return f.read().split(")")[-1].split()[0] != "Z"


def test_all_ranks_succeed_and_rank0_line_is_relayed(tmp_path, capfd):
    cmd = _script(tmp_path, """
        import os, sys
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        assert os.environ["LOCAL_RANK"] == os.environ["RANK"]
        print("[Gloo] Rank is connected: library chatter on stdout")
        print('{"rank": %d, "world": %d, "args": "%s"}' % (rank, world, " ".join(sys.argv[1:])))
    """)
    status = bench.launch_ranks(3, ["--gpus", "3"], command=cmd, timeout=60)
    captured = capfd.readouterr()
    assert status == 0
    # stdout carries rank 0's result line and nothing else
    assert captured.out == '{"rank": 0, "world": 3, "args": "--gpus 3"}\n'
    assert "[Gloo]" in captured.err


def test_ranks_that_print_no_result_line_fail_the_parent(tmp_path, capfd):
    cmd = _script(tmp_path, """
        print("no result here")
    """)
    status = bench.launch_ranks(2, ["--gpus", "2"], command=cmd, timeout=60)
    captured = capfd.readouterr()
    assert status != 0
    assert captured.out == ""


def test_one_failed_rank_stops_the_others_and_fails_the_parent(tmp_path, capfd):
    cmd = _script(tmp_path, f"""
        import os, sys, time
        rank = int(os.environ["RANK"])
        open(r"{tmp_path}/pid" + str(rank), "w").write(str(os.getpid()))
        if rank == 1:
            time.sleep(0.3)
            sys.exit(7)
        print("partial line of a run that did not finish")
        sys.stdout.flush()
        time.sleep(600)   # a rank waiting at a rendezvous that never completes
    """)
    t0 = time.monotonic()
    status = bench.launch_ranks(2, [], command=cmd, timeout=120)
    elapsed = time.monotonic() - t0
    captured = capfd.readouterr()
    assert status == 7 and elapsed < 30
    assert "rank 1 of 2 exited with 7" in captured.err
    assert captured.out == ""  # no result line from a failed job
    pid0 = int((tmp_path / "pid0").read_text())
    assert not _alive(pid0)


def test_a_signalled_rank_and_the_time_limit_fail_the_parent(tmp_path, capfd):
    cmd = _script(tmp_path, """
        import os, signal, time
        if os.environ["RANK"] == "0":
            time.sleep(0.2)
            os.kill(os.getpid(), signal.SIGKILL)
        time.sleep(600)
    """)
    assert bench.launch_ranks(2, [], command=cmd, timeout=120) == 128 + 9
    cmd = _script(tmp_path, "import time; time.sleep(600)")
    t0 = time.monotonic()
    assert bench.launch_ranks(2, [], command=cmd, timeout=1.0) == 124
    assert time.monotonic() - t0 < 30
    assert "did not finish within" in capfd.readouterr().err


def test_bench_without_a_launcher_fails_loudly_without_gpus():
    """The driver's calling shape (`python3 bench.py --gpus 2 ...`) on a box
    without GPUs: the ranks cannot bind a device, the parent must say so and
    exit non-zero instead of printing a line."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["ACCBLAS_BENCH_LAUNCH_TIMEOUT"] = "240"
    run = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2",
                          "--steps", "2", "--warmup", "1"], env=env, capture_output=True,
                         text=True, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        assert run.returncode == 0 and '"n_gpus": 2' in run.stdout
    else:
        assert run.returncode != 0
        assert run.stdout == ""
        assert "exited with" in run.stderr
