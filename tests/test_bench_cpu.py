"""CPU tests of the multi-rank plumbing of bench.py and of the collective watchdog (no GPU, no process group):
`python3 bench.py --gpus N` as typed starts its ranks as a CHILD job on 127.0.0.1 and relays the exit code; a phase that overruns its
budget ends the process with exit code 124 and one clear line."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_gpus_n_starts_a_child_job_on_loopback(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1", "--fp8"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = bench.launch_ranks(4)
    cmd = seen["cmd"]
    assert rc == 7                                              # the child's exit code is the parent's
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"   # the container hostname may not resolve
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--fp8"]      # the flags as typed reach every rank
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_without_world_size_never_touches_the_gpu_before_launching(monkeypatch):
    """--gpus 2 with WORLD_SIZE unset: main() must hand over to launch_ranks before any device call (a process that initialised HIP
    must not start / exec another program on this pool)."""
    sys.path.insert(0, ROOT)
    import bench
    import torch
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "launch_ranks", lambda n: 3)

    def boom(*a, **k):
        raise AssertionError("device call before the child job was started")

    monkeypatch.setattr(torch.cuda, "set_device", boom)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 3
    else:
        raise AssertionError("main() returned")


def test_watchdog_ends_the_process_with_124_when_a_phase_overruns():
    code = textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        from sfron import dp
        wd = dp.Watchdog(rank=3)
        wd.phase("stuck all-reduce", 0.2)
        time.sleep(30)            # the watchdog thread must end the process long before this returns
        print("not reached")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 124
    assert "rank 3" in r.stderr and "stuck all-reduce" in r.stderr and "not reached" not in r.stdout


def test_watchdog_stays_quiet_when_rearmed_and_stopped():
    code = textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        from sfron import dp
        wd = dp.Watchdog(rank=0)
        for i in range(4):
            wd.phase("step %d" % i, 1.5)
            time.sleep(0.6)
        wd.stop()
        time.sleep(1.2)
        print("done")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "done" in r.stdout and "watchdog" not in r.stderr
