"""bench.py's launch contract, checked without a GPU: --gpus N must never silently become one rank, and the host-side
dot product behind the 2^26 known answer is exact."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OG_BENCH_OVERSUBSCRIBE"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=300)


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "8", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 8 but WORLD_SIZE=2" in r.stderr
    assert '"n_gpus"' not in r.stdout


def test_more_gpus_than_visible_is_refused_not_downgraded():
    import torch
    if torch.cuda.device_count() >= 2:
        return  # only meaningful where fewer than 2 GPUs are visible (this container, a 1-GPU box)
    r = _run(["--gpus", "2", "--steps", "1"], {})
    assert r.returncode != 0 and "only" in r.stderr and "GPU(s) visible" in r.stderr
    assert '"n_gpus"' not in r.stdout


def test_host_dot_product_is_exact():
    sys.path.insert(0, ROOT)
    import bench
    from owshen_amd.api import FR_MODULUS, bytes_to_ints
    rng = np.random.default_rng(7)
    a = rng.integers(0, 256, (3000, 32), dtype=np.uint8)
    s = rng.integers(0, 256, (3000, 32), dtype=np.uint8)
    a[:5] = 255            # all-ones limbs: the largest partial sums
    s[:5] = 255
    want = sum(x * y for x, y in zip(bytes_to_ints(a), bytes_to_ints(s))) % FR_MODULUS
    assert bench.host_dot_mod_r(a, s) == want


class _FakeDist:
    """what bench.timed needs of Dist, on one rank and without a GPU"""
    class _Cuda:
        @staticmethod
        def synchronize():
            pass

    class _Torch:
        pass

    def __init__(self):
        self.torch = self._Torch()
        self.torch.cuda = self._Cuda()

    def fence(self):
        pass

    def all_times(self, t):
        return [t]

    def max_time(self, t):
        return t


def test_timed_runs_exactly_k_steps_and_refuses_results_that_differ():
    import importlib.util
    import pytest
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []

    def same():
        calls.append(1)
        return np.arange(8, dtype=np.uint8)

    d = _FakeDist()
    dt, out = bench.timed(d, same, 2, 5)
    assert len(calls) == 7 and dt > 0 and out.tolist() == list(range(8)) and d.last_steps_compared == 7
    k = [0]

    def drifting():
        k[0] += 1
        return np.full(8, 3 if k[0] == 4 else 1, dtype=np.uint8)

    with pytest.raises(SystemExit, match="different bytes"):
        bench.timed(_FakeDist(), drifting, 1, 4)


def test_host_cores_follow_the_container_quota(monkeypatch):
    """cpu_baseline.cores must be what the process can keep busy: the GPU boxes report 256 CPUs while their containers are held
    to 16 CPUs of run time (cgroup v2 cpu.max = "1600000 100000"); v1 (cpu.cfs_quota_us / cpu.cfs_period_us) and "no quota" too"""
    import builtins
    import io
    sys.path.insert(0, ROOT)
    import bench
    real_open = builtins.open
    monkeypatch.setattr(os, "cpu_count", lambda: 256)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)

    def fake(files):
        def _open(path, *a, **k):
            if isinstance(path, str) and path.startswith("/sys/fs/cgroup/"):
                if path in files:
                    return io.StringIO(files[path])
                raise FileNotFoundError(path)
            return real_open(path, *a, **k)
        return _open

    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}))
    n, info = bench.host_cores()
    assert n == 16 and info["os_cpu_count"] == 256 and info["cgroup_cpu_quota"] == 16.0 and info["usable"] == 16
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu.max": "max 100000\n"}))
    assert bench.host_cores()[0] == 256
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "800000\n", "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert bench.host_cores()[0] == 8
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "-1\n", "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert bench.host_cores()[0] == 256
    monkeypatch.setattr(builtins, "open", fake({}))
    assert bench.host_cores()[0] == 256


def test_a_rank_that_never_arrives_produces_an_error_line_within_the_watchdog(tmp_path):
    """a hung N-rank start-up must cost seconds, not the driver's timeout, and must leave a line.  Rank 0 of a
    2-rank gloo group whose rank 1 never comes up: `init_process_group` blocks in C; bench.py's CollectiveWatchdog (a timer thread)
    prints ONE JSON line -- "error", the ranks that reached the rendezvous, the N = 1 result of the fallback -- and leaves with
    exit code 3 within its limit."""
    import json
    import socket
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "hang.py"
    script.write_text(f"""
import importlib.util, os, sys
spec = importlib.util.spec_from_file_location("bench_wd", {BENCH!r})
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
wd = bench.CollectiveWatchdog(0, 2, "test{port}", 3.0, lambda: {{"metric": "withdraw proofs/sec (batch=1024)", "value": 123.0, "unit": "proofs/s", "n_gpus": 1}}).start()
wd.arrive()
import torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=2)   # rank 1 never arrives: blocks
print("UNREACHABLE")
""")
    t0 = time.time()
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120, cwd=ROOT)
    took = time.time() - t0
    assert p.returncode == 3, (p.returncode, p.stderr[-800:])
    assert took < 60, took
    assert "UNREACHABLE" not in p.stdout
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert "not through after 3 s" in line["error"] and line["ranks_arrived"] == [0] and "missing: [1]" in line["error"]
    assert line["n_gpus"] == 1 and line["n_gpus_requested"] == 2 and line["value"] == 123.0


def test_single_gpu_fallback_rewrites_the_command(monkeypatch):
    """the watchdog's fallback runs THIS command on one GPU: --gpus 1, a --batch-total turned into the rank's share, no --shard,
    no torch.distributed environment (checked on the argument rewriting alone: no GPU here)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_fb", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class _P:
        stdout = '{"value": 1}\n'

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return _P()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5", "--batch-total", "4096", "--shard", "windows"])
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,6,7,0,1,2,3")
    assert bench.single_gpu_fallback(2) == {"value": 1}
    assert seen["cmd"][2:] == ["--gpus", "1", "--steps", "20", "--warmup", "5", "--batch", "512"]
    assert "WORLD_SIZE" not in seen["env"] and "RANK" not in seen["env"] and seen["env"]["HIP_VISIBLE_DEVICES"] == "6"
