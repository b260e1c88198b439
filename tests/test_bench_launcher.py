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


def _parent_script(tmp_path, child_body, world=2):
    """A parent process running bench.launch_ranks over dummy ranks that write
    their pids and then wait for ever."""
    child = tmp_path / "rank.py"
    child.write_text(textwrap.dedent(child_body))
    parent = tmp_path / "parent.py"
    parent.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, r"{REPO}")
        import bench
        sys.exit(bench.launch_ranks({world}, [], command=[sys.executable, r"{child}"],
                                    timeout=300))
    """))
    return [sys.executable, str(parent)]


_WAITING_RANK = """
    import os, time
    open(r"{dir}/pid" + os.environ["RANK"], "w").write(str(os.getpid()))
    time.sleep(600)
"""


def _wait_for(paths, seconds=30):
    end = time.monotonic() + seconds
    while time.monotonic() < end:
        if all(p.exists() and p.read_text() for p in paths):
            return
        time.sleep(0.05)
    raise AssertionError("the dummy ranks did not start")


def _gone(pids, seconds=15):
    end = time.monotonic() + seconds
    while time.monotonic() < end:
        if not any(_alive(pid) for pid in pids):
            return True
        time.sleep(0.05)
    return False


def test_sigterm_to_the_parent_takes_every_rank_with_it(tmp_path):
    """A harness `timeout` or a cancelled job signals the parent: the ranks
    (each in its own session) must not be left running on the GPUs."""
    import signal
    for signum in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
        for old in tmp_path.glob("pid*"):
            old.unlink()
        parent = subprocess.Popen(_parent_script(tmp_path, _WAITING_RANK.format(dir=tmp_path)),
                                  stderr=subprocess.PIPE, text=True)
        pid_files = [tmp_path / "pid0", tmp_path / "pid1"]
        _wait_for(pid_files)
        pids = [int(p.read_text()) for p in pid_files]
        assert all(_alive(pid) for pid in pids)
        parent.send_signal(signum)
        _, err = parent.communicate(timeout=30)
        assert parent.returncode == 128 + signum, (parent.returncode, err)
        assert "the ranks were stopped" in err
        assert _gone(pids), f"ranks survived signal {signum} to the parent"


def test_a_parent_killed_outright_takes_every_rank_with_it(tmp_path):
    """SIGKILL cannot be handled: the ranks asked the kernel for SIGKILL on the
    death of their parent (PR_SET_PDEATHSIG)."""
    import signal
    parent = subprocess.Popen(_parent_script(tmp_path, _WAITING_RANK.format(dir=tmp_path)))
    pid_files = [tmp_path / "pid0", tmp_path / "pid1"]
    _wait_for(pid_files)
    pids = [int(p.read_text()) for p in pid_files]
    parent.send_signal(signal.SIGKILL)
    parent.wait(timeout=30)
    assert _gone(pids), "ranks survived the death of their parent"


def test_a_taken_rendezvous_port_starts_the_ranks_again(tmp_path, capfd):
    """bind-and-close can lose the port to another process before rank 0 binds
    it: rank 0 then reports the address in use and the job is started again."""
    marker = tmp_path / "attempt"
    cmd = _script(tmp_path, f"""
        import os, sys
        marker = r"{marker}"
        if os.environ["RANK"] == "0" and not os.path.exists(marker):
            open(marker, "w").write(os.environ["MASTER_PORT"])
            print("RuntimeError: The server socket has failed to listen on any local "
                  "network address. port: 1234, useIpv6: false, code: -98, name: "
                  "EADDRINUSE, message: address already in use", file=sys.stderr)
            sys.exit(1)
        print('{{"port": %s}}' % os.environ["MASTER_PORT"])
    """)
    status = bench.launch_ranks(2, [], command=cmd, timeout=60)
    captured = capfd.readouterr()
    assert status == 0
    assert captured.out.startswith('{"port": ')
    assert "starting the ranks again" in captured.err
