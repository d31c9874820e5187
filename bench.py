#!/usr/bin/env python3
"""bench.py -- withdraw proofs/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W [--workload prove|msm26|tree20] [--batch B] [--no-cpu] ...

--workload prove (default; BASELINE.json configs[1], and configs[3] when N > 1)
    A "step" proves one batch of withdraw statements end to end on the GPU: batched MiMC7 witness generation -> R1CS
    products -> H-polynomial (7 NTTs) -> 5 Pippenger MSMs -> proof assembly.  The circuit is the depth-32 MiMC7 Merkle
    withdraw circuit sized to n_wires = 2^18 / NTT domain 2^17 with synthetic padding gates; inputs (per-proof secrets,
    paths, blinding) are synthetic and resident in HBM when the timed region starts; the key comes from fixed toxic
    waste.  `value` is measured on the BASELINE-shaped circuit: dense padding, every wire has an A and a B base (two
    density rows, owshen_amd/circuit.py), i.e. the 0.9 x 2^20 G1 + 2^18 G2 points per proof that BASELINE.md section 2
    quotes.  The same line carries `sparse_padding` (the padding gates' natural query density: A ~50 %, B ~45 % of the
    wires have a base -- a lighter workload, reported for continuity with rounds 1-2), `roofline` (HBM, the contract's
    object) and `roofline_valu` (the bound that actually binds: VALU issue), and two short legs the driver thereby
    times as well: `msm26` (configs[2]) and `tree20` (configs[4]).
    N > 1: proofs are independent units -- each rank proves its own batch with a replicated key, no data-path
    collective (weak scaling); time = max over ranks between barriers.
--workload msm26 (configs[2])  one BN254 G1 MSM over 2^26 points (--log-n to shrink); N > 1: window-sharded
    (bases replicated, scalars identical, all-gather of the per-window points over RCCL).
--workload tree20 (configs[4])  MiMC7 Merkle tree over 2^20 leaves; N > 1: subtree per rank + all-gather of the roots.

N > 1 without a torch.distributed.run environment: the script re-launches itself under torch.distributed.run with N
ranks (one per GPU) and fails loudly if fewer than N GPUs are visible.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
TOXIC = (0x1F3A5C7E9B2D4F60718293A4B5C6D7E8F9, 0x2A4C6E8091B3D5F7, 0x3B5D7F91A3C5E7, 0x4C6E80A2C4E6, 0x5D7F91B3D5F7A9)
G1_POINT_BYTES, G2_POINT_BYTES = 96, 160   # algorithmic bytes per MSM point: affine base + 32 B scalar
NTT_ELEM_BYTES = 64                        # per element per transform
DTYPE = "u32 (9 x 29-bit-limb Montgomery, 254-bit modular integers)"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("prove", "msm26", "tree20", "plumbing"), default="prove")
    ap.add_argument("--batch", type=int, default=1024, help="proofs per step per GPU")
    ap.add_argument("--batch-total", type=int, default=None, help="proofs per step over ALL GPUs (BASELINE.json configs[3]: --gpus 8 "
                    "--batch-total 4096 = 512 per GPU); overrides --batch; a remainder goes to the lowest ranks")
    ap.add_argument("--depth", type=int, default=32)
    ap.add_argument("--natural", action="store_true", help="the natural circuit (no padding gates): ~2^15 constraints")
    ap.add_argument("--dense", action="store_true", help="dense padding only (the headline; skip the sparse-padding variant)")
    ap.add_argument("--sparse", action="store_true", help="make the padding-as-built (sparse) variant the headline and skip the other")
    ap.add_argument("--no-other", "--no-dense", dest="no_other", action="store_true", help="skip the secondary padding variant")
    ap.add_argument("--no-legs", action="store_true", help="skip the msm26 / tree20 legs of the default line")
    ap.add_argument("--ahead", action="store_true", help="prove: a step submits its batch and waits for the previous step's (one call kept "
                    "ahead, og_withdraw_prove_batch_submit_d) instead of one blocking call per step; measured +0.7 %% at 3 steps")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU-baseline leg")
    ap.add_argument("--in-process", action="store_true", help="prove: ONE process drives all --gpus N devices through og_multi_withdraw_prove_batch "
                    "(what a single-process node would call) instead of one process per GPU")
    ap.add_argument("--shard", choices=("proofs", "windows"), default="proofs", help="prove, --gpus N: `proofs` (default, the throughput form: rank g "
                    "proves its own slice, no data-path collective) or `windows` (BASELINE.json configs[3] as written: ONE batch on all N "
                    "GPUs together, rank g accumulates the MSM windows k = g mod N, one all-gather of 768 B per proof and rank)")
    ap.add_argument("--no-verify", action="store_true", help="prove: skip og_verify over every proof of the last timed step (A/B loops)")
    ap.add_argument("--no-isolated", action="store_true", help="prove: skip the extra serial (single-lane) steps -- the rocprofv3 PMC passes "
                    "profile the timed step alone, so that their per-launch averages are over exactly the launches the timed region has")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline sample budget")
    ap.add_argument("--log-n", type=int, default=None, help="msm26 / tree20: log2 of the size (default 26 / 20)")
    ap.add_argument("--precomp", action="store_true", help="msm26: per-window precomputed tables (round 3's form) instead of plain bases")
    ap.add_argument("--precomp-too", action="store_true", help="msm26: time the per-window-table variant beside the plain-bases headline")
    ap.add_argument("--no-precomp", action="store_true", help="(accepted for compatibility: plain bases are the default since round 4)")
    return ap.parse_args()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def ensure_ranks(args):
    """--gpus N must mean N ranks on N GPUs.  Under torch.distributed.run the environment says so; otherwise re-launch
    ourselves under it.  Never print an n_gpus = 1 line for --gpus 8."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: launch with --nproc-per-node {args.gpus}")
        return
    if args.gpus <= 1 or getattr(args, "in_process", False):
        return
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not os.environ.get("OG_BENCH_OVERSUBSCRIBE"):
        sys.exit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible (OG_BENCH_OVERSUBSCRIBE=1 allows a dry run "
                 "with several ranks per GPU)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] self-launch:", " ".join(cmd))
    sys.exit(subprocess.call(cmd))


class CollectiveWatchdog:
    """The first real N-rank run must fail loudly and cheaply: nothing multi-rank has met RCCL with N > 1 before
    the driver's SCALE run, and a hung `init_process_group("nccl")` / first collective would burn the driver's whole timeout and
    leave no line.  Every rank drops a marker file when it reaches the rendezvous; if start-up (init + first all-reduce + the
    self-test all-gather) is not through after `seconds` (default 480: RCCL's own start-up has taken 150 - 190 s with ONE rank on
    boxes of this pool, profiles/README.md round 5, so the limit leaves it room and still leaves the driver's 1 800 s room for the
    fallback), a timer thread -- the main thread may be stuck inside a C call -- makes
    rank 0 print ONE JSON line with "error", the ranks that arrived and the N = 1 result of its own GPU (`fallback()`: a fresh
    single-process run of the same command), and every rank leaves with exit code 3 (the others only after rank 0 is done: the
    launcher kills the group as soon as one rank exits)."""

    def __init__(self, rank, world, tag, seconds, fallback, what="process-group start-up"):
        import tempfile
        self.rank, self.world, self.seconds, self.fallback, self.what = rank, world, seconds, fallback, what
        self.dir = os.path.join(tempfile.gettempdir(), f"og_bench_{tag}")
        os.makedirs(self.dir, exist_ok=True)
        self._timer = None

    def arrive(self):
        with open(os.path.join(self.dir, f"rank_{self.rank}"), "w") as f:
            f.write(str(os.getpid()))

    def arrived(self):
        return sorted(int(n.split("_", 1)[1]) for n in os.listdir(self.dir) if n.startswith("rank_"))

    def _fire(self):
        try:
            if self.rank == 0:
                arrived = self.arrived()
                err = (f"{self.what} of {self.world} ranks not through after {self.seconds:.0f} s; ranks that reached the rendezvous: {arrived}"
                       + (f", missing: {[r for r in range(self.world) if r not in arrived]}" if len(arrived) < self.world else " (all arrived: the collective itself hangs)"))
                log("[bench] WATCHDOG: " + err)
                line = None
                try:
                    line = self.fallback()
                except Exception as e:  # noqa: BLE001
                    err += f"; the N = 1 fallback failed too ({type(e).__name__}: {e})"
                out = dict(line or {"metric": "withdraw proofs/sec (batch=1024)", "value": None, "unit": "proofs/s", "n_gpus": 1})
                out.update({"error": err, "n_gpus_requested": self.world, "ranks_arrived": arrived,
                            "note": "this is the N = 1 result of rank 0's GPU, printed because the N-rank start-up did not complete"})
                print(json.dumps(out), flush=True)
                with open(os.path.join(self.dir, "done"), "w") as f:
                    f.write("1")
            else:
                t0 = time.time()
                while not os.path.exists(os.path.join(self.dir, "done")) and time.time() - t0 < 1700:
                    time.sleep(1.0)
        finally:
            os._exit(3)

    def start(self):
        import threading
        self._timer = threading.Timer(self.seconds, self._fire)
        self._timer.daemon = True
        self._timer.start()
        return self

    def cancel(self):
        if self._timer is not None:
            self._timer.cancel()


def single_gpu_fallback(local_device):
    """the N = 1 line of this command on one GPU, from a fresh process (the watchdog's fallback): same arguments, --gpus 1, a
    --batch-total turned into this rank's share, no torch.distributed environment"""
    argv, out, skip = sys.argv[1:], [], False
    world = int(os.environ.get("WORLD_SIZE", "1"))
    for i, a in enumerate(argv):
        if skip:
            skip = False
            continue
        if a == "--gpus":
            out += ["--gpus", "1"]
            skip = True
        elif a.startswith("--gpus="):
            out.append("--gpus=1")
        elif a == "--batch-total":
            out += ["--batch", str(max(1, int(argv[i + 1]) // world))]
            skip = True
        elif a in ("--shard",):
            skip = True
        else:
            out.append(a)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                                                             "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "OG_BENCH_TEST_HANG_RANK")}
    vis = env.get("HIP_VISIBLE_DEVICES") or env.get("CUDA_VISIBLE_DEVICES")
    ids = vis.split(",") if vis else None
    env["HIP_VISIBLE_DEVICES"] = ids[local_device] if ids and local_device < len(ids) else str(local_device)
    env.pop("CUDA_VISIBLE_DEVICES", None)
    p = subprocess.run([sys.executable, os.path.abspath(__file__)] + out, env=env, capture_output=True, text=True, timeout=1500)
    return json.loads(p.stdout.strip().splitlines()[-1])


class Dist:
    """barriers + max-over-ranks of the elapsed time; RCCL first, gloo only if RCCL cannot start (dry runs)"""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        ndev = torch.cuda.device_count()
        if ndev == 0:
            sys.exit("bench.py: no GPU visible; owshen_amd has no CPU fallback")
        if self.world > ndev and not os.environ.get("OG_BENCH_OVERSUBSCRIBE"):
            sys.exit(f"bench.py: {self.world} ranks but only {ndev} GPU(s) visible")
        self.device = self.local_rank % ndev
        torch.cuda.set_device(self.device)
        self.backend = None
        self.self_test = None
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = os.environ.get("OG_BENCH_BACKEND", "nccl")
            # (the tag names THIS launch: the rendezvous port and the launcher's pid -- the parent of every rank -- so that a
            # marker left by an earlier run on the same port is not taken for an arrival)
            wd = CollectiveWatchdog(self.rank, self.world, f'{os.environ.get("MASTER_PORT", "0")}_{os.getppid()}', float(os.environ.get("OG_BENCH_WATCHDOG_S", "480")),
                                    lambda: single_gpu_fallback(self.device)).start()
            wd.arrive()
            if os.environ.get("OG_BENCH_TEST_HANG_RANK") == str(self.rank):   # (dry runs of the watchdog: this rank never reaches the rendezvous)
                os.remove(os.path.join(wd.dir, f"rank_{self.rank}"))
                time.sleep(3600)
            try:
                if backend == "nccl":
                    dist.init_process_group("nccl", device_id=torch.device("cuda", self.device))
                    t = torch.zeros(1, device="cuda")
                    dist.all_reduce(t)  # fail here, not inside the timed region
                    torch.cuda.synchronize()
                else:
                    dist.init_process_group(backend)
            except Exception as e:  # noqa: BLE001
                # A SCALE line that says "backend": "gloo" is not RCCL evidence: on a real N-GPU box an RCCL failure is fatal.
                # Only the oversubscribed dry run (several ranks on one GPU, where RCCL cannot start) may fall back.
                if not os.environ.get("OG_BENCH_OVERSUBSCRIBE"):
                    sys.exit(f"bench.py: rank {self.rank}: process group '{backend}' failed ({type(e).__name__}: {e}); refusing to fall "
                             "back to gloo (OG_BENCH_OVERSUBSCRIBE=1 allows it for dry runs, OG_BENCH_BACKEND=gloo asks for it)")
                log(f"[bench] rank {self.rank}: {backend} failed ({type(e).__name__}: {e}); oversubscribed dry run: falling back to gloo")
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = "gloo"
                dist.init_process_group("gloo")
            self.backend = backend
            got = dist.get_world_size()
            if got != self.world:
                sys.exit(f"bench.py: process group has {got} ranks, expected {self.world}")
            # self-test before anything is timed: a 1 MiB all-gather of bytes (the collective the window-sharded paths use), its
            # content checked and its latency put into the line (`ranks.self_test`)
            dev = "cuda" if backend == "nccl" else "cpu"
            mine = torch.full((1 << 20,), self.rank + 1, dtype=torch.uint8, device=dev)
            allb = torch.empty(self.world << 20, dtype=torch.uint8, device=dev)
            ts = []
            for _ in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                dist.all_gather_into_tensor(allb, mine)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            seen = [int(allb[(r << 20) + 12345].item()) for r in range(self.world)]
            if seen != [r + 1 for r in range(self.world)]:
                sys.exit(f"bench.py: rank {self.rank}: the self-test all-gather returned {seen}")
            self.self_test = {"all_gather_1MiB_per_rank_us": round(sorted(ts[1:])[len(ts[1:]) // 2] * 1e6, 1), "first_call_us": round(ts[0] * 1e6, 1),
                              "content_checked": True, "backend": backend}
            wd.cancel()

    def fence(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def all_times(self, dt):
        """every rank's elapsed time, on every rank (so that the line shows N ranks really ran)"""
        if self.world == 1:
            return [dt]
        t = self.torch.zeros(self.world, dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        t[self.rank] = dt
        self.dist.all_reduce(t)
        return [float(x) for x in t.tolist()]

    def all_values(self, x):
        """one number per rank, on every rank"""
        return self.all_times(float(x))

    def collective_info(self):
        info = {"backend": self.backend, "world": self.world, "self_test": self.self_test}
        try:
            v = self.torch.cuda.nccl.version()
            info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
        except Exception as e:  # noqa: BLE001
            info["rccl_version"] = f"unavailable ({type(e).__name__})"
        return info

    def all_objects(self, obj):
        """one picklable object per rank, on every rank"""
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def max_time(self, dt):
        if self.world == 1:
            return dt
        t = self.torch.tensor([dt], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed(dist, fn, warmup, steps, drain=None, period=1):
    """W untimed warmup steps, then EXACTLY K steps between barrier + synchronize fences; max over ranks.  `drain` completes
    whatever a step leaves in flight (a step that keeps one call ahead): it runs after the warm-up, so that the timed region
    starts idle, and after the K-th step INSIDE the timed region, so that all K batches are finished before the clock stops."""
    out = None
    seen = []                                 # every step's result: compared AFTER the clock stops (see below)

    def keep(new):
        nonlocal out
        if new is not None:
            out = new
            seen.append(new)

    for _ in range(warmup):
        keep(fn())
    if drain:
        keep(drain())
    dist.fence()
    marks = []                                # host clock after every step (a step is a blocking call unless --ahead): no extra sync
    t0 = time.perf_counter()
    for _ in range(steps):
        keep(fn())
        marks.append(time.perf_counter())
    if drain:
        keep(drain())
    dist.torch.cuda.synchronize()
    mine = time.perf_counter() - t0          # this rank's own work (before the closing barrier)
    dist.fence()
    dt = time.perf_counter() - t0
    dist.last_rank_times = dist.all_times(mine)
    dist.last_step_ms = [round((b - a) * 1e3, 3) for a, b in zip([t0] + marks[:-1], marks)]
    # every step ran the same inputs with the same blinding: the bytes must repeat.  A stream-ordering race in the stage
    # pipeline (scratch slots, call slots, events) would show up here, on all K x batch results, not only on the handful the
    # CPU leg re-proves.  Outside the timed region.
    blobs = [x.tobytes() for x in seen if hasattr(x, "tobytes")]
    dist.last_steps_compared = len(blobs)
    # `period` input sets are proved in turn (prove: 2, see ProveSetup.step): result k repeats result k - period
    if any(blobs[k] != blobs[k - period] for k in range(period, len(blobs))):
        sys.exit("bench.py: two steps over the same inputs produced different bytes -- the run is invalid")
    if period > 1 and len(blobs) > 1 and blobs[0] == blobs[1]:
        sys.exit("bench.py: two steps over DIFFERENT inputs produced the same bytes -- the run is invalid")
    return dist.max_time(dt), out


def step_stats(ms):
    """min / median / max of the timed steps' own durations (host clock around each blocking call)"""
    if not ms:
        return None
    v = sorted(ms)
    return {"min": v[0], "median": v[len(v) // 2] if len(v) & 1 else round((v[len(v) // 2 - 1] + v[len(v) // 2]) / 2, 3), "max": v[-1],
            "first": ms[0], "last": ms[-1], "n": len(ms)}


def device_identity(torch, device):
    """what makes "rank g ran on GPU g" a reading, not a claim: the device's UUID and PCI address as the runtime reports them"""
    out = {"index": int(device)}
    try:
        p = torch.cuda.get_device_properties(device)
        out["name"] = p.name
        if getattr(p, "uuid", None) is not None:
            out["uuid"] = str(p.uuid)
        if all(hasattr(p, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            out["pci"] = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        out["compute_units"] = int(p.multi_processor_count)
        out["hbm_bytes"] = int(p.total_memory)
    except Exception as e:  # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    out["visible_devices_env"] = {k: os.environ[k] for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if k in os.environ}
    return out


class GpuTelemetry:
    """Engine clock and socket power of this rank's GPU while the timed region runs, so that a slow line can be told from a
    slow box.  A reader thread on the HOST: amdgpu's hwmon files in sysfs (freq1_input = sclk in Hz, power1_average /
    power1_input in microwatts) every `period` seconds -- no GPU work, no subprocess, nothing inside the prover.  If sysfs
    is not there, `rocm-smi --json` is polled instead (a subprocess per sample, still outside the clock)."""

    def __init__(self, pci=None, period=0.25):
        import glob
        self.period, self.samples, self._stop, self._thr = period, [], None, None
        self.source, self.files = None, {}
        cards = []
        for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
            if not hw:
                continue
            slot = ""
            try:
                with open(os.path.join(dev, "uevent")) as f:
                    for line in f:
                        if line.startswith("PCI_SLOT_NAME="):
                            slot = line.strip().split("=", 1)[1].lower()
            except OSError:
                pass
            cards.append((slot, dev, hw[0]))
        pick = [c for c in cards if pci and c[0] == pci.lower()] or (cards if len(cards) == 1 else [])
        if pick:
            slot, dev, hw = pick[0]
            for key, names in (("sclk_hz", ("freq1_input",)), ("power_uw", ("power1_average", "power1_input")), ("temp_mc", ("temp1_input", "temp2_input"))):
                for nm in names:
                    path = os.path.join(hw, nm)
                    if os.path.exists(path):
                        self.files[key] = path
                        break
            if self.files:
                self.source = f"sysfs hwmon of {slot or dev}"
        if self.source is None:
            import shutil
            if shutil.which("rocm-smi"):
                self.source, self.period = "rocm-smi -c -P --json (device 0 of the visible set)", max(period, 1.0)

    def _read(self):
        if self.files:
            row = {}
            for k, path in self.files.items():
                try:
                    with open(path) as f:
                        row[k] = int(f.read().strip())
                except (OSError, ValueError):
                    pass
            return row
        try:
            txt = subprocess.run(["rocm-smi", "-c", "-P", "--json"], capture_output=True, text=True, timeout=10).stdout
            card = next(iter(json.loads(txt).values()))
            row = {}
            for k, v in card.items():
                kl = k.lower()
                if "sclk" in kl and "clock" in kl and "(" in str(v):
                    row["sclk_hz"] = int(float(str(v).split("(")[1].split("M")[0]) * 1e6)
                elif "power" in kl and "socket" in kl or "average graphics package power" in kl:
                    row["power_uw"] = int(float(v) * 1e6)
            return row
        except Exception:  # noqa: BLE001
            return {}

    def start(self):
        if self.source is None:
            return self
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                row = self._read()
                if row:
                    self.samples.append(row)
                self._stop.wait(self.period)

        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=15)
        return self.summary()

    def summary(self):
        if self.source is None:
            return {"source": None, "note": "no amdgpu hwmon in sysfs and no rocm-smi on this box: clock / power not sampled"}
        out = {"source": self.source, "samples": len(self.samples), "period_s": self.period}
        for key, name, scale in (("sclk_hz", "sclk_MHz", 1e-6), ("power_uw", "socket_power_W", 1e-6), ("temp_mc", "temp_C", 1e-3)):
            v = [r[key] * scale for r in self.samples if key in r]
            if v:
                out[name] = {"mean": round(sum(v) / len(v), 1), "min": round(min(v), 1), "max": round(max(v), 1)}
        return out


def pmc_profile():
    """measured per-launch counters of the dominant kernels at the headline launch size (profiles/pmc_traffic.json,
    produced by tools/pmc_round.sh + tools/pmc_summary.py from rocprofv3 --pmc passes over this same command)"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


# ---------------------------------------------------------------------------------------------------------------
# workload: prove
# ---------------------------------------------------------------------------------------------------------------

class ProveSetup:
    def __init__(self, ctx, args, rank, dense):
        import numpy as np
        from owshen_amd import circuit, groth16
        self.ctx, self.depth, self.dense = ctx, args.depth, dense
        self.blocking, self._pending = not bool(getattr(args, "ahead", False)), None
        t0 = time.time()
        self.n_pad3, self.n_pad2 = (0, 0) if args.natural else circuit.baseline_shape(args.depth, dense=dense)
        self.r1cs = circuit.withdraw_r1cs(ctx.mimc7_constants(), args.depth, self.n_pad3, self.n_pad2, dense=dense)
        self.blob, self.vk = groth16.setup(ctx, self.r1cs, *TOXIC)
        self.pk = groth16.ProvingKey(ctx, self.blob)
        self.m, self.d = self.pk.n_wires, 1 << self.pk.log_d
        self.density = self.pk.density()
        self.windows = self.pk.windows()      # window bits per query: a query of n points is n x ceil(255 / bits) additions per proof
        if rank == 0:
            log(f"[bench] circuit{' (dense padding)' if dense else ''}: n_wires={self.m} constraints={self.r1cs.n_constraints} "
                f"domain=2^{self.pk.log_d} nnz=({self.r1cs.a.nnz},{self.r1cs.b.nnz},{self.r1cs.c.nnz}) density={self.density} "
                f"key={len(self.blob) / 1e6:.0f} MB setup={time.time() - t0:.1f}s")
        B = self.batch = args.batch
        rng = np.random.Generator(np.random.PCG64(20241008 + rank))
        # TWO input sets, proved in turn: consecutive steps never see the same batch (nothing could be carried over from one
        # step to the next even if something cached results -- nothing does), and step k must still repeat step k - 2 byte for byte
        self.sets = []
        for _ in range(2):
            inputs = rng.integers(0, 256, (B, 8 + self.depth, 32), dtype=np.uint8)
            inputs[:, :, 31] &= 0x1F                       # < 2^253 < r
            inputs[:, 3, 20:] = 0                          # recipient: a 160-bit address
            inputs[:, 5, 8:] = 0                           # index: u64
            inputs[:, 6, 20:] = 0                          # token: a 160-bit address
            inputs[:, 7, 8:] = 0                           # chain id: u64
            if self.depth < 64:
                inputs[:, 5, :8] = (inputs[:, 5, :8].view(np.uint64) & np.uint64((1 << self.depth) - 1)).view(np.uint8)
            rs = rng.integers(0, 256, (B, 64), dtype=np.uint8)
            rs[:, 31] &= 0x1F
            rs[:, 63] &= 0x1F
            self.sets.append((ctx.to_device(inputs), rs))
        self.n_steps = 0
        self.last_set = 0
        self.inputs_d, self.rs = self.sets[0]
        self.public = [None, None]             # the public inputs og_withdraw_prove_batch_d hands back with each set's proofs

    def use(self, n):
        """restrict both input sets to their first n records (the batch-512 leg)"""
        self.sets = [(i[:n], r[:n]) for i, r in self.sets]
        self.batch = n
        self.n_steps = 0

    def step(self):
        """input records -> witnesses (batched MiMC7 walk) -> proofs, all on the GPU: one blocking og_withdraw_prove_batch_d.
        With --ahead a step SUBMITS its batch (og_withdraw_prove_batch_submit_d) and then waits for the previous step's: one
        call is kept ahead, the way a prover that is fed continuously runs, so a batch's cold start hides under the previous
        batch's last accumulations; `drain` finishes the batch still in flight (inside the timed region, after the K-th step)."""
        from owshen_amd import circuit
        self.last_set = self.n_steps & 1
        self.inputs_d, self.rs = self.sets[self.last_set]
        self.n_steps += 1
        # public_out is part of the call (root and nullifier_hash are COMPUTED by the witness walk: a caller cannot submit the
        # proof without them), so the timed call asks for it -- 192 B per proof more in the copy-out
        if self.blocking:
            proofs, self.public[self.last_set] = circuit.prove_from_inputs(self.ctx, self.pk, self.depth, self.inputs_d, self.rs,
                                                                           self.n_pad3, self.n_pad2, return_public=True)
            return proofs
        job = circuit.submit_from_inputs(self.ctx, self.pk, self.depth, self.inputs_d, self.rs, self.n_pad3, self.n_pad2, return_public=True)
        prev, self._pending = self._pending, (job, self.last_set)
        return self._finish(prev)

    def _finish(self, pending):
        if pending is None:
            return None
        job, which = pending
        proofs, self.public[which] = job.wait()
        return proofs

    def drain(self):
        prev, self._pending = self._pending, None
        return self._finish(prev)

    def windows_per_point(self):
        """{profile key: windows per accumulated point} for the G1 kernel (A, B, L, H queries, weighted by their points) and the
        G2 kernel (the B query)"""
        dn, w = self.density, {k: -(-255 // v) for k, v in self.windows.items()}
        g1 = sum(dn[k] * w[k] for k in "ablh") / float(sum(dn[k] for k in "ablh"))
        return {"accumulate_g1": g1, "accumulate_g2": float(w["b"])}

    def points(self):
        """MSM points actually accumulated per proof (after density compaction): (G1, G2)"""
        dn = self.density
        return dn["a"] + dn["b"] + dn["l"] + dn["h"], dn["b"]

    def algorithmic_bytes_per_proof(self):
        g1, g2 = self.points()
        return g1 * G1_POINT_BYTES + g2 * G2_POINT_BYTES + 7 * self.d * NTT_ELEM_BYTES

    def close(self):
        self.pk.close()


VALU_PEAK_CLOCK_GHZ = 2.4   # MI355X peak engine clock (MI355X_MICROARCH.md); 256 CUs x 4 SIMDs, one VALU wave-instruction per 4 cycles
N_SIMD = 1024
NWIN = 16


def roofline_of(prof, pmc, note, nwin=None):
    """dominant kernel = whichever bucket-accumulation kernel (G1 / G2) took more time.  Returns (hbm, valu):
    hbm   the contract's object: achieved = algorithmic bytes per launch / average launch duration (HIP events on the
          stream the kernel runs on, og_profile); traffic = measured FETCH_SIZE + WRITE_SIZE per launch (rocprofv3 PMC passes)
    valu  the bound that binds this kernel: achieved = VALU wave-instructions issued per second = (instructions per
          wave-level mixed addition, SQ_INSTS_VALU of the PMC passes -- a property of the binary) x (mixed additions of this
          run's launch) / (this run's launch duration); peak = 1024 SIMDs x clock / 4 cycles."""
    cands = []
    for key, name, pbytes in (("accumulate_g1", "k_accumulate_p<Fq> (G1 bucket accumulation)", G1_POINT_BYTES),
                              ("accumulate_g2", "k_accumulate_g2_lds (G2 bucket accumulation)", G2_POINT_BYTES)):
        ms, n, units = prof[key]
        if n:
            cands.append((ms, key, name, pbytes, n, units))
    if not cands:
        return None, None
    ms, key, name, pbytes, n, units = max(cands)
    alg = units * pbytes / n
    t = ms / n * 1e-3
    achieved = alg / t / 1e9 if ms > 0 else 0.0
    out = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "launches": n, "avg_launch_ms": round(ms / n, 4),
           "algorithmic_bytes_per_launch": int(alg), "points_per_launch": int(units / n), "note": note}
    valu = None
    k = pmc.get(key)
    if k:
        # measured with rocprofv3 PMC at a stated launch size; used as is when this run's launches have that size,
        # otherwise scaled per point and flagged
        same = abs(k["points_per_launch"] - units / n) <= 0.02 * k["points_per_launch"]
        out["traffic"] = int(k["hbm_bytes_per_launch"] if same else k["hbm_bytes_per_launch"] * (units / n) / k["points_per_launch"])
        out["traffic_source"] = (f"{pmc.get('source', 'profiles/pmc_traffic.json')}: FETCH_SIZE + WRITE_SIZE per launch"
                                 + ("" if same else " (scaled per point: this run's launch size differs from the profiled one)"))
        for f in ("l2_hit_rate",):
            if f in k:
                out[f] = k[f]
        if "valu_insts_per_madd" in k and ms > 0:
            # wave-level mixed additions of one launch (64 buckets per wave): points x windows, the windows averaged over
            # the queries this kernel serves, weighted by their points (the key's tables: og_pk_windows)
            wave_madds = units / n * (nwin or {}).get(key, NWIN) / 64.0
            ginst = k["valu_insts_per_madd"] * wave_madds / t / 1e9
            peak = N_SIMD * VALU_PEAK_CLOCK_GHZ / 4.0
            valu = {"bound": "valu", "kernel": name, "achieved": round(ginst, 2), "peak": round(peak, 2), "unit": "G wave-instructions/s",
                    "frac": round(ginst / peak, 4), "avg_launch_ms": round(ms / n, 4), "valu_insts_per_mixed_addition": k["valu_insts_per_madd"],
                    "note": f"peak = {N_SIMD} SIMDs x {VALU_PEAK_CLOCK_GHZ} GHz (peak engine clock) / 4 cycles per wave-instruction; instructions per "
                            "addition from SQ_INSTS_VALU (rocprofv3 PMC pass over this binary), additions and launch time from this run"}
            if "effective_clock_GHz" in k:
                clk = k["effective_clock_GHz"]
                valu["frac_at_profiled_clock"] = round(ginst / (N_SIMD * clk / 4.0), 4)
                valu["profiled_clock_GHz"] = clk   # the clock the PMC run's box held under this kernel; boxes differ by a few per
                # cent, so on a faster box this fraction can read slightly above 1 -- pmc_valu_util is the same-run figure
            for f in ("valu_util", "mad_issue_frac"):
                if f in k:
                    valu["pmc_" + f] = k[f]
    return out, valu


def isolated_step(ctx, dist, st):
    """one untimed, strictly serial step: kernel durations free of co-scheduling (what rocprof --stats of a single-lane run
    shows).  The serial path sorts the three density maps through ONE scratch slot, so a first serial step grows that slot to
    the largest query (hipMalloc inside the sort region: the 539-vs-147 ms digit_sort of round 2's driver line); it is run
    before the profiled one."""
    ctx.set_lanes(1)
    st.step()
    st.drain()
    dist.torch.cuda.synchronize()
    ctx.profile(True)
    t0 = time.perf_counter()
    st.step()
    st.drain()
    dist.torch.cuda.synchronize()
    isolated_step.last_ms = (time.perf_counter() - t0) * 1e3   # the `serial` leg: one strictly serial step (og_set_lanes(1))
    prof = ctx.profile_read()
    ctx.profile(False)
    ctx.set_lanes(2)
    return prof


def latency_leg(ctx, st, dist, sizes=(1, 8, 64), reps=8):
    """The one-request-per-call site (/root/reference/src/services/api_services/withdraw.rs:27-71): input records -> proofs for a
    call of 1 / 8 / 64 requests with this key, host clock around the blocking og_withdraw_prove_batch_d (records resident, proofs
    and public inputs copied out), median and best of `reps` calls after two warm-up calls."""
    from owshen_amd import circuit
    inputs_d, rs = st.sets[0]
    out = {}
    for b in sizes:
        if b > inputs_d.shape[0]:
            continue
        d, r = inputs_d[:b].contiguous(), rs[:b]
        ts = []
        for _ in range(reps + 2):
            dist.torch.cuda.synchronize()
            t0 = time.perf_counter()
            circuit.prove_from_inputs(ctx, st.pk, st.depth, d, r, st.n_pad3, st.n_pad2, return_public=True)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = sorted(ts[2:])
        out[f"requests_{b}"] = {"median_ms": round(ts[len(ts) // 2], 3), "min_ms": round(ts[0], 3), "plan": st.pk.plan(b)[0]}
    # the same calls with the MiMC7 chains walked on the host CPU (og_set_host_chains: opt-in, off in every other number of this
    # line): a request's walk is a chain of ~19 000 dependent products, 0.42 us each on a lone wave, 20-50 ns on a server core
    try:
        ctx.set_host_chains(64)
        hw = {}
        for b in sizes:
            if b > inputs_d.shape[0]:
                continue
            d, r = inputs_d[:b].contiguous(), rs[:b]
            ctx.set_host_chains(0)
            want = circuit.prove_from_inputs(ctx, st.pk, st.depth, d, r, st.n_pad3, st.n_pad2, return_public=True)
            ctx.set_host_chains(64)
            ts = []
            for _ in range(reps + 2):
                dist.torch.cuda.synchronize()
                t0 = time.perf_counter()
                got = circuit.prove_from_inputs(ctx, st.pk, st.depth, d, r, st.n_pad3, st.n_pad2, return_public=True)
                ts.append((time.perf_counter() - t0) * 1e3)
            ts = sorted(ts[2:])
            hw[f"requests_{b}"] = {"median_ms": round(ts[len(ts) // 2], 3), "min_ms": round(ts[0], 3),
                                   "same_bytes_as_the_kernels": got[0].tobytes() == want[0].tobytes() and got[1].tobytes() == want[1].tobytes()}
        # the deposit side's chain in the same mode: ONE leaf into the depth-32 commitment tree (og_mimc7_append_d; what mint_tx appends)
        import numpy as np
        leaf = ctx.to_device(np.full((1, 32), 7, dtype=np.uint8))
        app = {}
        for mode, bound in (("kernels", 0), ("host_chains", 64)):
            ctx.set_host_chains(bound)
            fr = ctx.to_device(np.zeros((32, 32), dtype=np.uint8))
            ts = []
            for _ in range(reps + 2):
                dist.torch.cuda.synchronize()
                t0 = time.perf_counter()
                _f2, root = ctx.mimc7_append(32, fr, 12345, leaf)
                dist.torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            app[mode] = {"median_ms": round(sorted(ts[2:])[len(ts[2:]) // 2], 3), "root": bytes(ctx.to_host(root)).hex()}
        hw["append_one_leaf_depth32"] = {"kernels_ms": app["kernels"]["median_ms"], "host_chains_ms": app["host_chains"]["median_ms"],
                                         "same_root": app["kernels"]["root"] == app["host_chains"]["root"]}
        hw["what"] = "og_set_host_chains(64): the requests' MiMC7 walks and proof assemblies on the host CPU (a thread per request), the MSMs / quotient / sorts on the GPU; append_one_leaf_depth32: og_mimc7_append_d in the same mode"
        hw["host_cores"] = host_cores()
        out["host_chains"] = hw
    except Exception as e:  # a leg can cost itself, never the line
        out["host_chains"] = {"error": repr(e)[:300]}
    finally:
        ctx.set_host_chains(0)
    return out


def deposit_leg(ctx, dist, args, n=8192, steps=2):
    """The other half of north_star's "deposit/withdraw circuits": the deposit statement (public commitment + depositor, private
    nullifier + secret, commitment = H(nullifier, secret): 735 wires, domain 2^10 -- oracle/py/deposit.py is the spec), a batch of
    `n` records -> proofs through og_deposit_prove_batch_d; every proof of the last step through og_verify, a sample byte-identical
    to the C restatement."""
    import numpy as np
    from owshen_amd import circuit, groth16
    rng = np.random.Generator(np.random.PCG64(20241009))
    r1 = circuit.deposit_r1cs_native(ctx)
    blob, vk = groth16.setup(ctx, r1, *TOXIC)
    pk = groth16.ProvingKey(ctx, blob)
    sets = []
    for _ in range(2):
        recs = rng.integers(0, 256, (n, 3, 32), dtype=np.uint8)
        recs[:, :, 31] &= 0x1F
        recs[:, 2, 20:] = 0                                  # depositor: a 160-bit address
        rs = rng.integers(0, 256, (n, 64), dtype=np.uint8)
        rs[:, 31] &= 0x1F
        rs[:, 63] &= 0x1F
        sets.append((ctx.to_device(recs), rs))
    state = {"k": 0, "pub": None}

    def step():
        ins, rs = sets[state["k"] & 1]
        state["k"] += 1
        proofs, state["pub"] = circuit.deposit_prove(ctx, pk, ins, rs, return_public=True)
        return proofs

    dt, proofs = timed(dist, step, 1, steps, None, period=2)
    ins, rs = sets[(state["k"] - 1) & 1]

    class _St:
        pass
    st = _St()
    st.vk = vk
    ver = verify_all(st, proofs, state["pub"]) if not args.no_verify else None
    same = None
    if not args.no_cpu:
        from oracle.c import binding as oc
        ck = oc.prepared_key_from_blob(blob)
        idx = [0, n // 2, n - 1]
        wit = ctx.to_host(circuit.deposit_witness(ctx, ins[idx].contiguous()))
        for j, t in enumerate(idx):
            assert proofs[t].tobytes() == ck.prove(wit[j], int.from_bytes(rs[t, :32].tobytes(), "little"), int.from_bytes(rs[t, 32:].tobytes(), "little")), \
                f"deposit proof {t} differs from the C restatement"
        same = {"proofs": len(idx), "indices": idx}
    mode, sizes = pk.plan(n)
    pk.close()
    return {"value": round(n * steps / dt, 3), "unit": "proofs/s", "steps": steps, "warmup": 1, "ms_per_step": round(dt / steps * 1e3, 3), "batch": n,
            "n_wires": r1.n_wires, "n_pub": r1.n_pub, "domain": 1 << r1.log_d, "sub_batch_plan": {"mode": mode, "sizes": sizes},
            "verified": (f"{ver['verified']} / {n}" if ver else None), "oracle_identical": same,
            "what": "the deposit statement (commitment = H(nullifier, secret), depositor bound): og_deposit_prove_batch_d, records resident in HBM"}


def window_shard_leg(ctx, st, dist, m, pks, sizes=(1, 16), world=8, reps=6):
    """Window-sharded proving measured on ONE GPU (the N-GPU form is `--gpus N --shard windows`; this leg prices its parts):
      through_one_rank   og_multi_withdraw_prove_sharded with one device: the whole call through the front / all-gather-free /
                         back split -- what the split itself costs against the unsharded call (`unsharded_ms`);
      share_of_8         ONE rank's share of an 8-rank call: the front half with win_world = 8 (windows k = r mod 8 of the 15:
                         rank 0 owns two, rank 7 one), timed for r = 0 and r = 7, and the back half over eight gathered blocks
                         (the other ranks' blocks produced the same way, untimed).  front + back is what an 8-GPU node would
                         take per call before the all-gather's own latency (768 B per proof and rank: microseconds over xGMI).
    Every form must produce the unsharded call's bytes."""
    import numpy as np
    from owshen_amd import circuit
    torch = dist.torch
    inputs_d, rs = st.sets[0]
    out = {"world": world, "collective_bytes_per_proof_per_rank": 768}

    def clock(fn):
        ts = []
        for _ in range(reps + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = sorted(ts[2:])
        return res, {"median_ms": round(ts[len(ts) // 2], 3), "min_ms": round(ts[0], 3)}

    for b in sizes:
        d, r = inputs_d[:b].contiguous(), rs[:b]
        host = ctx.to_host(d)
        want, t_un = clock(lambda: circuit.prove_from_inputs(ctx, st.pk, st.depth, d, r, st.n_pad3, st.n_pad2))
        got1, t_one = clock(lambda: m.withdraw_prove_sharded(pks, st.depth, host, r, st.n_pad3, st.n_pad2))
        assert got1.tobytes() == want.tobytes(), "the window-sharded call (one rank) differs from the unsharded call"
        parts, fronts = [], {}
        for rk in range(world):
            if rk in (0, world - 1):
                part, fronts[f"rank_{rk}"] = clock(lambda rk=rk: circuit.partials_from_inputs(ctx, st.pk, st.depth, d, rk, world, st.n_pad3, st.n_pad2))
            else:
                part = circuit.partials_from_inputs(ctx, st.pk, st.depth, d, rk, world, st.n_pad3, st.n_pad2)
            parts.append(part.clone())
        gathered = torch.cat(parts)
        got8, t_back = clock(lambda: st.pk.prove_from_partials(gathered, world, r))
        assert got8.tobytes() == want.tobytes(), "eight window shares do not add up to the unsharded proofs"
        out[f"requests_{b}"] = {"unsharded_ms": t_un, "through_one_rank_ms": t_one,
                                "share_of_8": {"front_ms": fronts, "back_ms": t_back,
                                               "front_plus_back_median_ms": round(max(v["median_ms"] for v in fronts.values()) + t_back["median_ms"], 3)},
                                "byte_identical_to_unsharded": True}
    return out


def rank_batch(batch_total, world, rank):
    """BASELINE.json configs[3] as written: a batch of T proofs over all N GPUs -- T / N each, a remainder to the lowest ranks (the rule
    og_multi_slice applies inside the library: 4096 over 8 = 512 each)"""
    return batch_total // world + (1 if rank < batch_total % world else 0)


def run_prove_in_process(args, dist, ctx, make_multi=None):
    """--in-process: ONE process drives all N GPUs through og_multi_withdraw_prove_batch -- what the reference's single-process
    node (/root/reference/src/cli/node.rs:71-76) would call: the library shards the batch (contiguous slices, og_multi_slice),
    one persistent host thread per device, key replicated, inputs and proofs in HOST memory (1.3 KB in, 448 B out per proof), no
    data-path collective.  Same circuit, same steps, same checks as the process-per-GPU mode; `ranks.devices` is the library's
    own reading of every rank's device (og_multi_device_info: HIP ordinal, PCI address, RCCL communicator size / rank)."""
    import numpy as np
    from owshen_amd import circuit, groth16, multi
    N = args.gpus
    dense = not args.sparse and not args.natural
    m = make_multi(N) if make_multi else multi.Multi(N)
    assert m.size == N, f"og_multi_init gave {m.size} devices, asked for {N}"
    depth = args.depth
    n_pad3, n_pad2 = (0, 0) if args.natural else circuit.baseline_shape(depth, dense=dense)
    r1cs = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2, dense=dense)
    blob, vk = groth16.setup(ctx, r1cs, *TOXIC)
    pks = m.load_key(blob)
    total = args.batch_total if args.batch_total is not None else args.batch * N
    slices = [m.slice(total, r) for r in range(N)]
    assert [hi - lo for lo, hi in slices] == [rank_batch(total, N, r) for r in range(N)], "og_multi_slice and the per-rank rule disagree"
    rng = np.random.Generator(np.random.PCG64(20241008))
    sets = []
    for _ in range(2):
        inputs = rng.integers(0, 256, (total, 8 + depth, 32), dtype=np.uint8)
        inputs[:, :, 31] &= 0x1F
        inputs[:, 3, 20:] = 0
        inputs[:, 5, 8:] = 0
        inputs[:, 6, 20:] = 0
        inputs[:, 7, 8:] = 0
        if depth < 64:
            inputs[:, 5, :8] = (inputs[:, 5, :8].view(np.uint64) & np.uint64((1 << depth) - 1)).view(np.uint8)
        rs = rng.integers(0, 256, (total, 64), dtype=np.uint8)
        rs[:, 31] &= 0x1F
        rs[:, 63] &= 0x1F
        sets.append((inputs, rs))
    state = {"n": 0, "pub": [None, None]}

    def step():
        k = state["n"] & 1
        state["n"] += 1
        proofs, state["pub"][k] = m.withdraw_prove_batch(pks, depth, sets[k][0], sets[k][1], n_pad3, n_pad2, return_public=True)
        state["last"] = k
        return proofs

    for _ in range(args.warmup):
        step()
    state["n"] = 0
    dt, proofs = timed(dist, step, 0, args.steps, None, period=2)
    assert proofs is not None and proofs.any()

    class _St:
        pass
    st = _St()
    st.vk = vk
    verified = None if args.no_verify else verify_all(st, proofs, state["pub"][state["last"]])
    devices = [{"rank": r, "slice": list(slices[r]), **m.device_info(r)} for r in range(N)]
    distinct = len({d["pci"] or d["device"] for d in devices})
    m.free_key(pks)
    m.close()
    return {
        "metric": f"withdraw proofs/sec (batch of {total} over {N} GPU(s), one process)", "value": round(total * args.steps / dt, 3), "unit": "proofs/s",
        "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "ranks": {"mode": "in-process: og_multi_withdraw_prove_batch (one process, one persistent host thread per device)", "world": N,
                  "per_rank_batch": [hi - lo for lo, hi in slices], "devices": devices, "distinct_devices": distinct,
                  "note": "devices[g] = og_multi_device_info(g): the HIP ordinal and PCI address rank g is bound to and RCCL's own account of "
                          "its communicator (comm_nranks = ncclCommCount; 0 with one device: no communicator is created)"},
        "step_ms": step_stats(list(dist.last_step_ms)),
        "repeatability": {"results_compared": dist.last_steps_compared, "byte_identical": True, "input_sets": 2,
                          "verified": f"{verified['verified']} / {total} proofs of the last timed step accepted by og_verify" if verified else None},
        "config": {"workload": ("natural depth-%d withdraw circuit" % depth) if args.natural else
                   f"BASELINE.json configs[{'3' if N > 1 else '1'}]: ONE batch of {total} withdraw proofs over {N} GPU(s) from one process, depth-{depth} MiMC7 "
                   f"Merkle circuit sized to n_wires=2^18 / NTT 2^17 ({'dense' if dense else 'sparse'} padding)",
                   "batch_total": total, "n_wires": r1cs.n_wires, "merkle_depth": depth, "inputs": "host memory (the withdraw path's boundary)",
                   "parallelism": f"proofs sharded across {N} device(s) inside the library, key replicated, no data-path collective"},
        "roofline": None, "cpu_baseline": None,
    }


def run_prove_window_sharded(args, dist, ctx):
    """--shard windows: BASELINE.json configs[3] as written -- ONE batch proved by all N GPUs together.  Every rank holds the same
    input records and blinding (same seed), walks the witnesses and the quotient itself and accumulates the MSM windows
    k = rank (mod N) of the five queries over its replica of the key; the partial points (768 B per proof and rank) meet in ONE
    all-gather over RCCL / xGMI (torch.distributed; never an all-reduce: curve points do not add limb-wise) and every rank
    assembles.  `value` = proofs of the batch / time: total work is fixed as N grows ("strong"); the throughput form is the
    default `--shard proofs`.  Also timed: a call of 1 and of 16 requests (the latency this split exists for)."""
    import numpy as np
    from owshen_amd import circuit, shard
    rank, world = dist.rank, dist.world
    dense = not args.sparse and not args.natural
    total = args.batch_total if args.batch_total is not None else args.batch
    la = argparse.Namespace(**vars(args))
    la.batch = total
    st = ProveSetup(ctx, la, 0, dense)          # seed of rank 0 on every rank: identical inputs
    group = None if world == 1 else dist.dist.group.WORLD

    def step():
        st.last_set = st.n_steps & 1
        ins, rs = st.sets[st.last_set]
        st.n_steps += 1
        proofs, st.public[st.last_set] = shard.prove_window_sharded(ctx, st.pk, rs, inputs_d=ins, depth=st.depth, n_pad3=st.n_pad3, n_pad2=st.n_pad2,
                                                                     group=group, return_public=True)
        return proofs

    for _ in range(args.warmup):
        step()
    st.n_steps = 0
    ident = device_identity(dist.torch, dist.device)
    tele = GpuTelemetry(ident.get("pci")).start()
    dt, proofs = timed(dist, step, 0, args.steps, None, period=2)
    telemetry = tele.stop()
    assert proofs is not None and proofs.any()
    digest = int.from_bytes(__import__("hashlib").sha256(proofs.tobytes()).digest()[:6], "big")
    digests = [int(x) for x in dist.all_values(digest)]
    assert len(set(digests)) == 1, f"the ranks assembled different proofs: {digests}"
    verified = verify_all(st, proofs, st.public[st.last_set]) if (rank == 0 and not args.no_verify) else None
    # against the unsharded call on rank 0's GPU (same key, same inputs): byte-identical
    same = None
    if rank == 0:
        ins, rs = st.sets[st.last_set]
        same = circuit.prove_from_inputs(ctx, st.pk, st.depth, ins, rs, st.n_pad3, st.n_pad2).tobytes() == proofs.tobytes()
        assert same, "window-sharded proofs differ from og_withdraw_prove_batch_d on the same inputs"
    lat = {}
    for b in (1, 16):
        if b > total:
            continue
        ins, rs = st.sets[0][0][:b].contiguous(), st.sets[0][1][:b]
        ts = []
        for _ in range(8):
            dist.fence()
            t0 = time.perf_counter()
            shard.prove_window_sharded(ctx, st.pk, rs, inputs_d=ins, depth=st.depth, n_pad3=st.n_pad3, n_pad2=st.n_pad2, group=group)
            ts.append(dist.max_time(time.perf_counter() - t0) * 1e3)
        ts = sorted(ts[2:])
        lat[f"requests_{b}"] = {"median_ms": round(ts[len(ts) // 2], 3), "min_ms": round(ts[0], 3)}
    mode, sizes = st.pk.plan(total)
    identities = dist.all_objects({**ident, "rank": rank, "pid": os.getpid(), "host": socket.gethostname(), "telemetry": telemetry})
    st.close()
    if rank != 0:
        return None
    distinct = len({(i.get("host"), i.get("uuid") or i.get("pci") or i.get("index")) for i in identities})
    return {
        "metric": f"withdraw proofs/sec (ONE batch of {total}, MSM windows sharded over {world} GPU(s))", "value": round(total * args.steps / dt, 3),
        "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "ranks": {**dist.collective_info(), "devices": identities, "distinct_devices": distinct, "self_test": getattr(dist, "self_test", None)},
        "step_ms": step_stats(list(dist.last_step_ms)), "latency": lat,
        "repeatability": {"results_compared": dist.last_steps_compared, "byte_identical": True, "input_sets": 2,
                          "all_ranks_assembled_the_same_proofs": True, "equals_the_unsharded_call_on_rank_0": same,
                          "verified": f"{verified['verified']} / {total}" if verified else None},
        "config": {"workload": f"BASELINE.json configs[3] as written: ONE batch of {total} withdraw proofs, MSM windows sharded over {world} GPU(s); "
                   f"{'natural depth-%d statement' % args.depth if args.natural else 'circuit sized to n_wires=2^18 / NTT 2^17 (' + ('dense' if dense else 'sparse') + ' padding)'}",
                   "batch_total": total, "n_wires": st.m, "domain": st.d, "sub_batch_plan": {"mode": mode, "sizes": sizes},
                   "query_window_bits": dict(st.windows),
                   "parallelism": f"windows k = rank (mod {world}) of every query per GPU, key replicated, witness / quotient replicated; one all-gather of "
                                  f"{768 * total} B per rank per call" + (f" over {dist.backend}" if dist.backend else " (one rank: no collective)")},
        "roofline": None, "cpu_baseline": None,
    }


def run_prove(args, dist, ctx):
    if getattr(args, "in_process", False):
        return run_prove_in_process(args, dist, ctx)
    if getattr(args, "shard", "proofs") == "windows":
        return run_prove_window_sharded(args, dist, ctx)
    rank, world = dist.rank, dist.world
    headline_dense = not args.sparse and not args.natural
    pad_name = "none" if args.natural else ("dense" if headline_dense else "sparse")
    if args.batch_total is not None:
        args.batch = rank_batch(args.batch_total, world, rank)
        assert args.batch >= 1, "--batch-total smaller than the number of GPUs"
    st = ProveSetup(ctx, args, rank, headline_dense)
    B = args.batch
    batches = [int(x) for x in dist.all_values(B)]
    for _ in range(args.warmup):
        st.step()
    st.drain()
    st.n_steps = 0
    ctx.profile(True)
    ident = device_identity(dist.torch, dist.device)
    tele = GpuTelemetry(ident.get("pci")).start()   # a host thread reading sysfs: clock and power WHILE the timed steps run
    dt, proofs = timed(dist, st.step, 0, args.steps, st.drain, period=2)
    telemetry = tele.stop()
    prof = ctx.profile_read()
    ctx.profile(False)
    assert proofs is not None and proofs.any(), "prover returned empty proofs"
    proved_inputs_d, proved_rs = st.sets[st.last_set]   # the batch `proofs` belongs to (the CPU leg re-proves a sample of it)
    proved_public = st.public[st.last_set].copy()
    step_ms = list(dist.last_step_ms)
    rank_times = list(dist.last_rank_times)
    steps_compared = dist.last_steps_compared
    value = sum(batches) * args.steps / dt
    # HBM accounting and the schedule of one step, per rank (so that a SCALE line explains itself)
    mem = ctx.mem_info()
    key_bytes = st.pk.hbm_bytes()
    plan_mode, plan_sizes = st.pk.plan(B)
    log(f"[bench] rank {rank}: batch {B}, {plan_mode} {plan_sizes}; HBM: key {key_bytes / 2**30:.2f} GiB, scratch {mem['scratch_bytes'] / 2**30:.1f} GiB "
        f"in {mem['scratch_buffers']} buffers, device {(mem['device_total_bytes'] - mem['device_free_bytes']) / 2**30:.1f} of {mem['device_total_bytes'] / 2**30:.0f} GiB in use")
    rank_scratch = [int(x) for x in dist.all_values(mem["scratch_bytes"])]
    rank_in_use = [int(x) for x in dist.all_values(mem["device_total_bytes"] - mem["device_free_bytes"])]
    pmc = pmc_profile().get(pad_name if pad_name != "none" else "sparse", {})
    roofline, roofline_valu = roofline_of(prof, pmc, "timed region (pipelined: a launch shares the GPU with the other streams' kernels). "
                                          "Modular big-integer path: bound by integer multiply-add VALU issue, not HBM -- see roofline_valu (DESIGN.md 4.1, 5)",
                                          st.windows_per_point())
    breakdown = {k: round(v[0] / args.steps, 3) for k, v in prof.items()}
    if args.no_isolated:
        roofline_isolated = roofline_valu_isolated = breakdown_isolated = None
    else:
        prof1 = isolated_step(ctx, dist, st)
        roofline_isolated, roofline_valu_isolated = roofline_of(prof1, pmc, "extra untimed single-lane step", st.windows_per_point())
        breakdown_isolated = {k: round(v[0], 3) for k, v in prof1.items()}

    # every proof of the last timed step in front of the product's verifier (all ranks: each verifies its own batch)
    verified = verify_all(st, proofs, proved_public) if not args.no_verify else None
    n_verified = [int(x) for x in dist.all_values(verified["verified"] if verified else 0)]
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from owshen_amd import circuit
        cpu = cpu_baseline_prove(ctx, st, proved_inputs_d, proved_rs, proofs, args.cpu_seconds, plan_sizes=plan_sizes if plan_mode == "stage pipeline" else None)
    g1, g2 = st.points()
    cfg_density = dict(st.density)
    alg_mb = st.algorithmic_bytes_per_proof() / 1e6
    m, d = st.m, st.d
    leg512 = None
    if world == 1 and B == 1024 and not args.natural and not args.no_legs:
        # BASELINE.json configs[3] is "a batch of 4096 over 8 GPUs" = 512 proofs per GPU per call: the per-GPU rate at THAT
        # batch size (a shorter ramped plan, a larger cold-start share), measured here on one GPU with the same key
        st.use(512)
        k5 = max(2, min(args.steps, 4))
        dt5, p5 = timed(dist, st.step, 1, k5, st.drain, period=2)
        assert p5 is not None and p5.any()
        mode5, sizes5 = st.pk.plan(512)
        leg512 = {"value": round(512 * k5 / dt5, 3), "unit": "proofs/s", "steps": k5, "warmup": 1, "ms_per_step": round(dt5 / k5 * 1e3, 3),
                  "batch_per_gpu": 512, "sub_batch_plan": {"mode": mode5, "sizes": sizes5},
                  "what": "the per-GPU share of BASELINE.json configs[3] (4096 proofs over 8 GPUs = 512 per GPU per call), same key and "
                          "circuit as the headline, one blocking call per step; `python bench.py --gpus 8 --batch-total 4096` runs the "
                          "configuration itself"}
    lat = inproc = wshard = None
    serial = None
    if breakdown_isolated is not None and getattr(isolated_step, "last_ms", None):
        serial = {"value": round(B * 1e3 / isolated_step.last_ms, 3), "unit": "proofs/s", "ms_per_step": round(isolated_step.last_ms, 3),
                  "what": "ONE extra untimed step with og_set_lanes(1): every sub-batch strictly serial on one stream and one scratch slot "
                          "(the step whose stage times are stage_ms_per_step_isolated); value / this = what the stage pipeline buys"}
    if world == 1 and rank == 0 and not args.natural and not args.no_legs:
        # the call site's own shape: 1 / 8 / 64 requests per call with the headline's key
        lat = latency_leg(ctx, st, dist)
        # the same batch through ONE process and the library's own device layer (og_multi_withdraw_prove_batch: host records in,
        # proofs out -- what the reference's single-process node would call), and the window-sharded form priced on this GPU
        try:
            from owshen_amd import multi
            ctx.release_scratch()             # this leg's library-owned context needs the room the main context's slots hold
            mm = multi.Multi(1)
            pks = mm.load_key(st.blob)
            host_in = ctx.to_host(proved_inputs_d)
            got = mm.withdraw_prove_batch(pks, st.depth, host_in, proved_rs, st.n_pad3, st.n_pad2)   # warm-up: scratch, first touch
            assert got.tobytes() == proofs.tobytes(), "og_multi_withdraw_prove_batch differs from og_withdraw_prove_batch_d on the same batch"
            kip = 2
            dist.torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(kip):
                mm.withdraw_prove_batch(pks, st.depth, host_in, proved_rs, st.n_pad3, st.n_pad2, return_public=True)
            dti = time.perf_counter() - t0
            inproc = {"value": round(B * kip / dti, 3), "unit": "proofs/s", "steps": kip, "warmup": 1, "ms_per_step": round(dti / kip * 1e3, 3),
                      "byte_identical_to_the_timed_call": True,
                      "what": "og_multi_withdraw_prove_batch on one device: the batch as HOST records (1.3 KB in, 448 B out per proof over PCIe), "
                              "one process, the library's own context and worker -- `bench.py --in-process --gpus N` is the N-device form"}
            wshard = window_shard_leg(ctx, st, dist, mm, pks)
            mm.free_key(pks)
            mm.close()
        except Exception as e:  # noqa: BLE001 -- a leg, never a reason to lose the line
            log(f"[bench] in-process / window-shard legs skipped: {type(e).__name__}: {e}")
            inproc = inproc or {"error": f"{type(e).__name__}: {e}"}
    zkey_leg = None
    if world == 1 and rank == 0 and not args.natural and not args.no_legs:
        # the headline's own key out as a snarkjs .zkey and back in (og_zkey_export / og_zkey_import): the load-time path a node
        # whose key comes from a circom / snarkjs ceremony takes once.  Not part of a step; timed on the host clock.
        try:
            from owshen_amd import groth16, zkey as zk
            vkb = groth16.vk_to_bytes(st.vk)
            t0 = time.perf_counter()
            zdata = zk.export_zkey(ctx, st.blob, vkb)
            t1 = time.perf_counter()
            pk2, vk2 = zk.import_zkey(ctx, zdata)
            t2 = time.perf_counter()
            nl, nh = m - st.pk.n_pub - 1, d - 1
            tail = sum((x + 31) // 32 * 32 for x in (64 * m, 64 * m, 128 * m, 64 * nl, 64 * nh))
            zkey_leg = {"export_s": round(t1 - t0, 3), "import_s": round(t2 - t1, 3), "zkey_bytes": len(zdata), "n_wires": m, "domain": d,
                        "queries_byte_identical": pk2[-tail:] == st.blob[-tail:] and pk2[80:592] == st.blob[80:592], "verifying_key_identical": vk2 == vkb,
                        "what": "the headline's key -> .zkey (file Montgomery form, ffjavascript's root order, odd-coset Lagrange H section by an "
                                "inverse DFT over 2^17 G1 points) -> back (the forward DFT): every group element of the five queries must return "
                                "byte for byte; tests/test_gpu_zkey.py proves with the re-imported key"}
            del zdata, pk2
        except Exception as e:  # noqa: BLE001 -- a leg, never a reason to lose the line
            log(f"[bench] zkey leg skipped: {type(e).__name__}: {e}")
            zkey_leg = {"error": f"{type(e).__name__}: {e}"}
    st.close()

    other = None
    if not args.natural and not args.no_other and not args.dense and not args.sparse:
        # the other padding variant next to the headline: same circuit size, the padding gates' natural query density
        dist.torch.cuda.empty_cache()
        so = ProveSetup(ctx, args, rank, not headline_dense)
        k2 = max(1, min(args.steps, 3))
        dt2, p2 = timed(dist, so.step, 1, k2, so.drain, period=2)
        assert p2 is not None and p2.any()
        prof2 = isolated_step(ctx, dist, so)
        og1, og2 = so.points()
        opmc = pmc_profile().get("sparse" if headline_dense else "dense", {})
        other = {"value": round(B * k2 * world / dt2, 3), "unit": "proofs/s", "steps": k2, "warmup": 1,
                 "ms_per_step": round(dt2 / k2 * 1e3, 3), "n_dense": dict(so.density), "g1_points_per_proof": og1,
                 "g2_points_per_proof": og2, "algorithmic_MB_per_proof": round(so.algorithmic_bytes_per_proof() / 1e6, 1),
                 "roofline_isolated": roofline_of(prof2, opmc, "untimed single-lane step", so.windows_per_point())[0],
                 "stage_ms_per_step_isolated": {k: round(v[0], 3) for k, v in prof2.items()},
                 "what": ("padding as built: a padding wire sits on one side of one gate, so the A query keeps ~50 % and the B query ~45 % of "
                          "the wires -- a LIGHTER workload than BASELINE.json configs[1]; rounds 1-2 quoted it as the headline")
                 if headline_dense else
                 ("every wire has an A and a B base (two density rows: (sum of all wires) * 0 = 0 and 0 * (sum) = 0); "
                  "BASELINE.md section 2's 0.9 x 2^20 G1 + 2^18 G2 points per proof")}
        so.close()

    # A yardstick measured in this run, on this box: the generated Montgomery product (fe_mul, the very asm statement the mixed
    # additions are made of) as a chain per lane at the G1 kernel's occupancy (3 waves per SIMD).  It replaces the ASSUMED peak of
    # roofline_valu (4 cycles per wave-instruction at the peak clock) with a measured one -- in products per second, and in
    # wave-instructions per second (205 instructions per product: 162 v_mad_u64_u32, 9 v_mul_lo_u32, masks and carries).
    if world == 1 and rank == 0 and not args.natural:
        try:
            import torch
            n_cu = torch.cuda.get_device_properties(dist.device).multi_processor_count
            nl = n_cu * 4 * 64 * 3
            g = torch.Generator().manual_seed(7)
            xh = torch.randint(0, 256, (nl, 32), dtype=torch.uint8, generator=g)
            xh[:, 31] &= 0x1F
            xd, yd = xh.cuda(), xh.flip(0).contiguous().cuda()
            ctx.field_mulchain(1, xd, yd, 64)
            it = 4096
            ms_y = min(ctx.field_mulchain(1, xd, yd, it) for _ in range(3))
            mm = nl * it / (ms_y * 1e-3)
            yard = {"kernel": "k_mulchain<Fq>: fe_mul (mont_gfx950.inc, one asm statement of 205 instructions) chained per lane, 3 waves per SIMD",
                    "mulmod_per_s": round(mm, 1), "wave_instructions_per_s_G": round(mm * 205 / 64 / 1e9, 2), "lanes": nl, "products_per_lane": it,
                    "ms": round(ms_y, 3)}
            for rv in (roofline_valu, roofline_valu_isolated):
                if rv:
                    rv["measured_yardstick"] = dict(yard, frac=round(rv["achieved"] / yard["wave_instructions_per_s_G"], 4),
                                                    note="achieved / the wave-instruction rate of the product chain measured in this run: the accumulation "
                                                         "kernel against what the same box sustains on the bare product (a mixed addition's squarings and "
                                                         "fused reductions carry fewer masks / carries per multiply-add than a lone product, so ~1.1 is the kernel at the product's own rate)")
            del xd, yd
        except Exception as e:  # noqa: BLE001 -- a yardstick, never a reason to lose the line
            log(f"[bench] product-chain yardstick skipped: {type(e).__name__}: {e}")

    natural = None
    if world == 1 and rank == 0 and not args.natural and not args.no_legs:
        # what `withdraw_handler` would actually prove: the depth-32 statement without padding gates (26 385 wires), a batch of 4096
        import copy
        dist.torch.cuda.empty_cache()
        ctx.release_scratch()
        na = copy.copy(args)
        na.natural, na.batch, na.batch_total, na.ahead = True, 4096, None, False
        sn = ProveSetup(ctx, na, rank, False)
        kn = 2
        dtn, pn = timed(dist, sn.step, 1, kn, sn.drain, period=2)
        assert pn is not None and pn.any()
        moden, sizesn = sn.pk.plan(na.batch)
        vern = verify_all(sn, pn, sn.public[sn.last_set]) if not args.no_verify else None
        cpun = None
        if not args.no_cpu:
            nin, nrs = sn.sets[sn.last_set]
            cpun = cpu_baseline_prove(ctx, sn, nin, nrs, pn, 0.0, plan_sizes=sizesn if moden == "stage pipeline" else None)
        natural = {"value": round(na.batch * kn / dtn, 3), "unit": "proofs/s", "steps": kn, "warmup": 1, "ms_per_step": round(dtn / kn * 1e3, 3),
                   "batch": na.batch, "n_wires": sn.m, "domain": sn.d, "sub_batch_plan": {"mode": moden, "sizes": sizesn},
                   "query_window_bits": dict(sn.windows), "verified": (f"{vern['verified']} / {na.batch}" if vern else None),
                   "oracle_identical": (cpun or {}).get("oracle_identical"), "cpu_proofs_per_s": (cpun or {}).get("value"),
                   "latency": latency_leg(ctx, sn, dist),
                   "what": "the natural depth-32 withdraw statement (no padding gates): what /root/reference/src/services/api_services/withdraw.rs:27-71 "
                           "would prove per request; batch 4096 = the throughput form, latency = 1 / 8 / 64 requests per call"}
        sn.close()
        ctx.release_scratch()
    deposit = None
    if world == 1 and rank == 0 and not args.natural and not args.no_legs:
        try:
            deposit = deposit_leg(ctx, dist, args)
        except Exception as e:  # noqa: BLE001 -- a leg, never a reason to lose the line
            log(f"[bench] deposit leg skipped: {type(e).__name__}: {e}")
            deposit = {"error": f"{type(e).__name__}: {e}"}
        ctx.release_scratch()

    legs = {}
    if world == 1 and not args.natural and not args.no_legs:
        # BASELINE.json configs[2] and configs[4] as short legs of the default line, so that the driver's run times them too
        import copy
        dist.torch.cuda.empty_cache()
        la = copy.copy(args)
        la.steps, la.warmup, la.no_cpu, la.log_n, la.precomp, la.precomp_too = 2, 1, True, None, False, True
        ctx.release_scratch()
        legs["msm26"] = compact_leg(run_msm(la, dist, ctx))
        ctx.release_scratch()
        la.steps = 3
        legs["tree20"] = compact_leg(run_tree(la, dist, ctx))

    identities = dist.all_objects({**ident, "rank": rank, "pid": os.getpid(), "host": socket.gethostname(), "telemetry": telemetry,
                                   "step_ms": step_stats(step_ms)})
    if rank != 0:
        return None
    distinct = len({(i.get("host"), i.get("uuid") or i.get("pci") or i.get("index")) for i in identities})
    acc_iso = (breakdown_isolated or {}).get("accumulate_g1")
    pad_text = {"dense": "dense padding: every wire has an A and a B base", "sparse": "padding density as built: see n_dense", "none": "no padding"}[pad_name]
    out = {
        "metric": "withdraw proofs/sec (batch=1024)" if args.batch_total is None else f"withdraw proofs/sec (batch of {args.batch_total} over {world} GPU(s))",
        "value": round(value, 3), "unit": "proofs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "ranks": {**dist.collective_info(), "per_rank_proofs_per_s": [round(b * args.steps / t, 2) for b, t in zip(batches, rank_times)],
                  "per_rank_batch": batches, "per_rank_scratch_bytes": rank_scratch, "per_rank_hbm_in_use_bytes": rank_in_use,
                  "devices": identities, "distinct_devices": distinct,
                  "note": "devices[g] = rank g's own reading of the GPU it ran on (torch.cuda.get_device_properties: UUID, PCI address), its "
                          "pid, its clock / power samples and its per-step times; `world` is the size of the communicator the barriers ran on; "
                          "distinct_devices must equal n_gpus"},
        "step_ms": step_stats(step_ms),
        "box": {"sclk_MHz": (telemetry or {}).get("sclk_MHz"), "socket_power_W": (telemetry or {}).get("socket_power_W"),
                "temp_C": (telemetry or {}).get("temp_C"), "telemetry_source": (telemetry or {}).get("source"),
                "samples": (telemetry or {}).get("samples"),
                "accumulate_g1_isolated_ms_per_step": acc_iso,
                "note": "clock / power: sampled by a host thread during the timed steps (sysfs hwmon, no GPU work).  accumulate_g1_isolated = "
                        "the G1 bucket accumulation of ONE extra serial step (single-lane, nothing beside it): a box-speed index -- the pool's "
                        "boxes differ by a few per cent in exactly this kernel (profiles/README.md lists the index of every quoted run)"},
        "repeatability": {"results_compared": steps_compared, "byte_identical": True, "input_sets": 2,
                          "verified": (f"{sum(n_verified)} / {sum(batches)} proofs of the last timed step accepted by og_verify" if verified else None),
                          "og_verify": verified,
                          "oracle_identical": (cpu or {}).get("oracle_identical"),
                          "summary": (f"{sum(n_verified)} / {sum(batches)} verified" if verified else "not verified") +
                          (f", {cpu['oracle_identical']['proofs']} oracle-identical across {cpu['oracle_identical']['sub_batches_covered']} of "
                           f"{cpu['oracle_identical']['sub_batches']} sub-batches" if cpu else ""),
                          "note": "the timed steps alternate between two input batches (different witnesses and blinding); step k must "
                                  "repeat step k - 2 byte for byte and differ from step k - 1, or the run aborts; after the clock stops every "
                                  "proof of the last step goes through og_verify with the public inputs the call returned, and the C "
                                  "restatement re-proves the first and last proof of every sub-batch of the plan (byte equality)"},
        "config": {"workload": ("natural depth-%d withdraw circuit" % args.depth) if args.natural else
                   f"BASELINE.json configs[1]: batch of {B} withdraw proofs per GPU, depth-{args.depth} MiMC7 Merkle circuit sized to "
                   f"n_wires=2^18 / NTT 2^17 with synthetic padding gates ({pad_text}); "
                   f"{g1} G1 + {g2} G2 MSM points accumulated per proof after density compaction",
                   "batch_per_gpu": B, "batch_total": sum(batches), "n_wires": m, "domain": d, "merkle_depth": args.depth,
                   "sub_batch_plan": {"mode": plan_mode, "sizes": plan_sizes},
                   "hbm": {"key_bytes": key_bytes, "scratch_bytes_reserved": mem["scratch_bytes"], "scratch_buffers": mem["scratch_buffers"],
                           "device_in_use_bytes": mem["device_total_bytes"] - mem["device_free_bytes"], "device_total_bytes": mem["device_total_bytes"],
                           "note": "after the timed steps: the resident key (CSR + per-window query tables) and the prover's scratch arena "
                                   "(two sub-batch slots + call-level buffers), og_mem_info / og_pk_bytes"},
                   "n_dense": cfg_density, "g1_points_per_proof": g1, "g2_points_per_proof": g2,
                   "query_window_bits": dict(st.windows),
                   "padding": {"dense": "dense (every wire in A and B: BASELINE.md section 2's point counts)", "none": "none",
                               "sparse": "sparse (A ~50 %, B ~45 % of the wires)"}[pad_name],
                   "calls": "one blocking og_withdraw_prove_batch_d per step" if not args.ahead else
                   "a step submits its batch (og_withdraw_prove_batch_submit_d) and waits for the previous step's: one call kept ahead; "
                   "all K batches complete inside the timed region",
                   "parallelism": f"proofs sharded across {world} GPU(s), key replicated, no data-path collective"
                   + (f" (barriers / max-time over {dist.backend})" if dist.backend else "")},
        "roofline": roofline,
        "roofline_valu": roofline_valu,
        "roofline_isolated": roofline_isolated,
        "roofline_valu_isolated": roofline_valu_isolated,
        "cpu_baseline": cpu,
        "stage_ms_per_step": breakdown,
        "stage_ms_per_step_isolated": breakdown_isolated,
        "algorithmic_MB_per_proof": round(alg_mb, 1),
    }
    if other:
        out["sparse_padding" if headline_dense else "dense_padding"] = other
    if leg512:
        out["batch512"] = leg512
    for k, v in (("serial", serial), ("latency", lat), ("in_process", inproc), ("window_sharded", wshard), ("natural", natural), ("deposit", deposit), ("zkey", zkey_leg)):
        if v is not None:
            out[k] = v
    out.update(legs)
    return out


def compact_leg(line):
    """the fields of a msm26 / tree20 bench line worth carrying inside the default line"""
    if line is None:
        return None
    cfg = line.get("config", {})
    keep = {k: line[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step") if k in line}
    keep["workload"] = cfg.get("workload")
    for k in ("known_answer", "parity", "table_build_s", "ms_per_msm_including_table_build", "precomputed_tables", "table_bytes"):
        if cfg.get(k) is not None:
            keep[k] = cfg[k]
    rf = line.get("roofline") or {}
    keep["roofline"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "accumulate_ms_per_launch") if k in rf}
    if line.get("stage_ms_per_step"):
        keep["stage_ms_per_step"] = line["stage_ms_per_step"]
    if line.get("step_ms"):
        keep["step_ms"] = line["step_ms"]
    if line.get("box"):
        keep["box"] = line["box"]
    if line.get("cpu_baseline"):
        keep["cpu_baseline"] = line["cpu_baseline"]
    if line.get("with_window_tables"):
        keep["with_window_tables"] = line["with_window_tables"]
    return keep


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """(usable, info): the hardware threads this PROCESS can actually keep busy -- its scheduler affinity capped by the container's
    CPU quota (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`).  os.cpu_count() reports the machine: 256 on the
    pool's GPU boxes, whose containers are held to 16 CPUs of run time (`cpu.max` = 1600000 100000, read in round 5) -- the
    "256 threads" of rounds 1 - 4's cpu_baseline lines were 256 threads time-sliced over 16 cores."""
    n = os.cpu_count() or 1
    info = {"os_cpu_count": n}
    try:
        aff = len(os.sched_getaffinity(0))
        info["affinity"] = aff
    except (AttributeError, OSError):
        aff = n
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        info["cgroup_cpu_quota"] = round(quota, 2)
    usable = max(1, min(aff, int(quota + 0.5) if quota else aff))
    info["usable"] = usable
    return usable, info


def plan_sample(plan_sizes, want):
    """proof indices that put EVERY sub-batch of the stage pipeline in front of the oracle: the first and the last proof of
    each sub-batch (sub-batch k runs in scratch slot k mod 2, so k and k + 2 straddle a slot's reuse), then midpoints until
    `want` indices are reached.  Returns (indices, sub-batch of each index)."""
    bounds, lo = [], 0
    for sz in plan_sizes:
        bounds.append((lo, lo + sz - 1))
        lo += sz
    idx = []
    for a, b in bounds:
        for i in (a, b):
            if i not in idx:
                idx.append(i)
    depth = 2
    while len(idx) < want and depth <= 64:
        grew = False
        for a, b in bounds:
            for j in range(1, depth):
                i = a + (b - a) * j // depth
                if len(idx) < want and i not in idx:
                    idx.append(i)
                    grew = True
        depth *= 2
        if not grew and depth > 64:
            break
    which = []
    for i in idx:
        which.append(next(k for k, (a, b) in enumerate(bounds) if a <= i <= b))
    return idx, which


def verify_all(st, proofs, public):
    """og_verify (the product's CPU verifier, the `burn_tx` seam: /root/reference/src/blockchain/tx/burn_tx.rs:11-32 accepts a
    burn or refuses it) over EVERY proof of the last timed step, with the public inputs the same call returned -- after the
    clock has stopped, on the host's cores (ctypes releases the GIL).  A proof of one statement must also be refused for its
    neighbour's public inputs (a sample)."""
    from concurrent.futures import ThreadPoolExecutor
    from owshen_amd import groth16
    vkb = groth16.vk_to_bytes(st.vk)
    n = proofs.shape[0]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(min(2 * host_cores()[0], 256)) as ex:
        ok = list(ex.map(lambda i: groth16.verify(vkb, public[i], proofs[i].tobytes()), range(n)))
        cross = list(ex.map(lambda i: groth16.verify(vkb, public[(i + 1) % n], proofs[i].tobytes()), range(0, n, max(1, n // 8)))) if n > 1 else []
    dt = time.perf_counter() - t0
    bad = [i for i, v in enumerate(ok) if not v]
    if bad:
        sys.exit(f"bench.py: og_verify refuses {len(bad)} of the {n} proofs of the last timed step (first: {bad[0]}) -- the run is invalid")
    if any(cross):
        sys.exit("bench.py: og_verify accepts a proof for another statement's public inputs -- the run is invalid")
    return {"verified": n, "of": n, "refused_for_a_neighbours_inputs": len(cross), "seconds": round(dt, 2),
            "verifier": "og_verify (CPU, EIP-197 predicate), public inputs as returned by the timed call"}


def cpu_baseline_prove(ctx, st, inputs_d, rs, gpu_proofs, budget_s, group_threads=16, plan_sizes=None):
    """The C restatement of the prover (oracle/c, TEST INFRASTRUCTURE) timed on this host's cores over a bounded sample of
    the same batch; doubles as an end-of-run parity check at full size.  Throughput form: ONE PROOF PER CORE GROUP of
    `group_threads` threads (its MSMs window-parallel inside the group), cpu_count / group_threads proofs side by side,
    whole waves until the budget is spent -- every hardware thread busy with independent proofs, the way a CPU prover farm
    would run.  Round 4 measured the alternatives on a 2 x EPYC 9575F box (256 threads): 64 proofs x 4 threads 0.97 proofs/s
    (one 66 s wave), round 3's 3 proofs x ~85 threads 1.13: the host saturates at ~1 proof/s with this C code whatever the
    split, so the default keeps a wave short (16 proofs x 16 threads, ~17 s)."""
    from concurrent.futures import ThreadPoolExecutor
    from owshen_amd import circuit
    os.environ.setdefault("OG_ORACLE_NATIVE", "1")   # tune the C restatement for THIS host (built here, -march=native)
    from oracle.c import binding as oc
    ck = oc.prepared_key_from_blob(st.blob)
    ncpu, cores_info = host_cores()                   # what the container may use, not what the machine has
    if ncpu < 64:
        group_threads = 4                             # a small host: 4 threads per proof, usable / 4 proofs side by side
    group_threads = max(1, min(group_threads, ncpu))
    groups = max(1, ncpu // group_threads)
    B = inputs_d.shape[0]

    def one(args):
        i, wit = args
        r = int.from_bytes(rs[i, :32].tobytes(), "little")
        s = int.from_bytes(rs[i, 32:].tobytes(), "little")
        p = ck.prove(wit, r, s, threads=group_threads)
        assert p == gpu_proofs[i].tobytes(), f"GPU proof {i} differs from the CPU restatement"
        return 1

    # The sample covers the whole sub-batch plan of the timed call: first and last proof of every sub-batch (the stage pipeline
    # rotates two scratch slots, so sub-batches k and k + 2 share one), filled up with midpoints to a whole wave
    order, which = plan_sample(plan_sizes or [B], max(groups, 2 * len(plan_sizes or [B])))
    order = [i for i in order if i < B]
    import torch
    done, t_total, waves, proved = 0, 0.0, 0, []
    with ThreadPoolExecutor(groups) as ex:
        must = min(len(order), 2 * len(plan_sizes or [B]))  # the parity duty: first + last proof of every sub-batch, whatever the budget
        while done < len(order) and (done < must or t_total < budget_s):
            idx = order[done:done + groups]
            sel = torch.as_tensor(idx, device=inputs_d.device)
            wit_d = circuit.witness(ctx, st.depth, inputs_d[sel].contiguous(), st.n_pad3, st.n_pad2)  # this wave's witnesses (GPU-generated)
            wits = [ctx.to_host(wit_d[k]) for k in range(len(idx))]
            del wit_d
            t0 = time.perf_counter()
            list(ex.map(one, zip(idx, wits)))
            t_total += time.perf_counter() - t0
            done += len(idx)
            proved += idx
            waves += 1
    subs = sorted({which[order.index(i)] for i in proved})
    return {"value": round(done / t_total, 4), "unit": "proofs/s", "cores": min(ncpu, groups * group_threads),
            "kind": "port", "sample": f"{done} proof(s) of the same batch, {groups} at a time x {group_threads} threads each ({waves} wave(s), "
            f"{t_total:.1f} s), byte-identical to the GPU proofs; the sample is the first and last proof of every sub-batch of the timed "
            f"call's plan (+ midpoints): indices {sorted(proved)}; own C restatement" +
            (" built -O3 -march=native on this host" if getattr(oc, "NATIVE", False) else "") +
            " -- the reference has no prover (SURVEY.md 0.1)", "host_cpus": cores_info["os_cpu_count"], "host_cores": cores_info,
            "cpu_model": cpu_model(),
            "proofs_in_flight": groups, "threads_per_proof": group_threads,
            "oracle_identical": {"proofs": done, "indices": sorted(proved), "sub_batches_covered": len(subs), "sub_batches": len(plan_sizes or [B])}}


# ---------------------------------------------------------------------------------------------------------------
# workload: msm26 (BASELINE.json configs[2]) -- one G1 MSM over 2^log_n points
# ---------------------------------------------------------------------------------------------------------------

def run_msm(args, dist, ctx):
    """one BN254 G1 MSM over 2^log_n points.  The headline is the PLAIN-bases path (no per-window tables: 4 GB of bases, nothing
    to build); with --precomp-too (the default leg does) the per-window-table variant of round 3 is timed beside it."""
    import numpy as np
    import torch
    from owshen_amd import api, groth16, shard
    rank, world = dist.rank, dist.world
    log_n = args.log_n or 26
    n = 1 << log_n
    # bases P_i = a_i G generated on the GPU (identical on every rank: same seed), scalars uniform with a few zeros / ones
    g = torch.Generator(device="cuda").manual_seed(26)
    a = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    a[:, 31] &= 0x0F
    s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    s[:, 31] &= 0x0F
    s[::16] = 0
    s[1::16] = 0
    s[1::16, 0] = 1
    t0 = time.time()
    pts = ctx.scalar_mul(1, groth16.G1_GEN_BYTES, a)
    t_gen = time.time() - t0
    modes = [True] if args.precomp else [False]
    if getattr(args, "precomp_too", False) and world == 1 and not args.precomp:
        modes.append(True)
    alg = n * G1_POINT_BYTES
    results, gots = [], []
    for precomp in modes:
        t0 = time.time()
        bases = api.Bases(ctx, 1, pts, 16, precomp)
        torch.cuda.synchronize()
        t_tab = time.time() - t0

        def step():
            if world == 1:
                return bases.msm(s)[0]
            return shard.msm_window_sharded(bases, s)

        for _ in range(args.warmup):
            step()
        ctx.profile(True)
        tele = GpuTelemetry(device_identity(torch, dist.device).get("pci"), period=0.02).start()   # host thread, sysfs: the clock the steps ran at
        dt, got = timed(dist, step, 0, args.steps)
        telemetry = tele.stop()
        prof = ctx.profile_read()
        ctx.profile(False)
        ms = dt / args.steps * 1e3
        step_ms = list(dist.last_step_ms)
        gots.append(bytes(got.tobytes()))
        bases.close()
        ctx.release_scratch()
        acc_n = prof["accumulate_g1"][1]
        acc_ms = prof["accumulate_g1"][0] + prof["heavy_g1"][0]
        results.append({"precomp": precomp, "ms": ms, "step_ms": step_ms, "telemetry": telemetry, "t_tab": t_tab, "acc_ms_per_launch": round(acc_ms / acc_n, 3) if acc_n else None,
                        "stages": {k2: round(v[0] / args.steps, 3) for k2, v in prof.items() if v[1]}})
    # known answer, AFTER the clocks have stopped, and with no leg of it from the library under test:
    # sum_i s_i (a_i G) = (sum a_i s_i mod r) G with the dot product taken on the HOST (numpy half-limb products, exact), k G by the
    # C restatement (oracle/c, the checker), and a random sample of the GPU-generated bases compared with the C restatement's a_i G
    check = None
    if rank == 0:
        import numpy as np
        from oracle.c import binding as oc
        t0 = time.time()
        a_h = a.cpu().numpy()
        k = host_dot_mod_r(a_h, s.cpu().numpy())
        want = oc.fixed_base_g1(np.frombuffer(groth16.G1_GEN_BYTES, dtype=np.uint8), api.ints_to_bytes([k]))[0].tobytes()
        idx = np.random.default_rng(n).choice(n, min(n, 2048), replace=False)
        got_b = pts[torch.from_numpy(idx).to(pts.device)].cpu().numpy().tobytes()
        assert got_b == oc.fixed_base_g1(np.frombuffer(groth16.G1_GEN_BYTES, dtype=np.uint8), a_h[idx]).tobytes(), "GPU-generated bases differ from the C restatement's a_i G"
        for g in gots:
            assert g == want, "MSM differs from the known answer (sum a_i s_i) G"
        del a_h
        check = (f"== (sum a_i s_i mod r) G: dot product on the host, k G and a sample of {len(idx)} bases from the C restatement "
                 f"({time.time() - t0:.1f} s, after the timed region)")
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline_msm(ctx, a, s, args.cpu_seconds)
    del pts
    if rank != 0:
        return None
    head = results[0]
    ms, precomp = head["ms"], head["precomp"]

    def roof(r):
        return {"bound": "hbm", "kernel": "whole MSM (digit sort + bucket accumulation [k_accumulate_p / k_accumulate_heavy<Fq>] + reduction)",
                "achieved": round(alg / (r["ms"] * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(alg / (r["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": None, "algorithmic_bytes": alg,
                "accumulate_ms_per_launch": r["acc_ms_per_launch"],
                "note": "96 B per point (64 B base + 32 B scalar); VALU-bound modular arithmetic (DESIGN.md 5)"}

    out = {
        "metric": "BN254 G1 MSM (2^%d points): points/sec" % log_n, "value": round(n / (ms * 1e-3), 1), "unit": "points/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[2]: one BN254 G1 MSM over 2^{log_n} points, scalars resident in HBM; "
                   + ("per-window precomputed tables" if precomp else "plain bases (one bucket set per window, nothing precomputed)"),
                   "n": n, "window_bits": 16, "precomputed_tables": precomp, "table_bytes": n * 64 * (16 if precomp else 1),
                   "launch_form": None if precomp or world > 1 or n < (1 << 22) else "two window halves side by side on the context's two lanes (og_msm_d, "
                   "DESIGN.md 4.6): the halves' stage regions overlap in time, so stage_ms_per_step sums to more than the step",
                   "table_build_s": round(head["t_tab"], 3), "base_generation_s": round(t_gen, 3),
                   "ms_per_msm_including_table_build": round(ms + head["t_tab"] * 1e3, 1) if precomp else round(ms, 3),
                   "parallelism": "1 GPU" if world == 1 else f"window-sharded over {world} GPUs: bases replicated, rank g takes windows "
                   f"k = g mod {world}, all-gather of the per-window points ({dist.backend})", "known_answer": check},
        "roofline": roof(head),
        "cpu_baseline": cpu,
        "step_ms": dict(step_stats(head["step_ms"]) or {}, each=head["step_ms"]),
        "box": {k: (head["telemetry"] or {}).get(k) for k in ("sclk_MHz", "socket_power_W", "temp_C", "samples", "source")},
        "stage_ms_per_step": head["stages"],
    }
    if len(results) > 1:
        r = results[1]
        out["with_window_tables"] = {"ms_per_step": round(r["ms"], 3), "value": round(n / (r["ms"] * 1e-3), 1), "unit": "points/s",
                                     "table_bytes": n * 64 * 16, "table_build_s": round(r["t_tab"], 3),
                                     "ms_per_msm_including_table_build": round(r["ms"] + r["t_tab"] * 1e3, 1), "roofline": roof(r),
                                     "stage_ms_per_step": r["stages"], "known_answer": "same in-run check",
                                     "what": "round 3's form: tab[k][i] = 2^(16 k) P_i for the 16 windows (68.7 GB at 2^26 points), one bucket set"}
    return out


def host_dot_mod_r(a_bytes, s_bytes):
    """sum_i a_i s_i mod r for uint8 [n,32] little-endian arrays, on the HOST (the oracle-side leg of the known answer):
    16-bit half-limbs as float64, one BLAS product per 2^20-element chunk -- every partial sum is below
    2^16 * 2^16 * 2^20 = 2^52, so float64 is exact -- recombined with Python integers."""
    import numpy as np
    from owshen_amd.api import FR_MODULUS
    n = a_bytes.shape[0]
    a16 = a_bytes.view(np.uint16).reshape(n, 16)
    s16 = s_bytes.view(np.uint16).reshape(n, 16)
    total = 0
    CH = 1 << 20
    for lo in range(0, n, CH):
        m = a16[lo:lo + CH].astype(np.float64).T @ s16[lo:lo + CH].astype(np.float64)   # [16,16], exact integers < 2^52
        for p in range(16):
            for q in range(16):
                total += int(m[p, q]) << (16 * (p + q))
    return total % FR_MODULUS


def cpu_baseline_msm(ctx, a, s, budget_s):
    """C restatement of the Pippenger MSM (oracle/c) on a bounded sample of the same points (2^24: a few seconds), window-parallel"""
    import numpy as np
    from owshen_amd import groth16
    os.environ.setdefault("OG_ORACLE_NATIVE", "1")
    from oracle.c import binding as oc
    ns = min(a.shape[0], 1 << 24)
    pts = ctx.scalar_mul(1, groth16.G1_GEN_BYTES, a[:ns]).cpu().numpy()
    sc = s[:ns].cpu().numpy()
    t0 = time.perf_counter()
    oc.msm_g1(pts, sc)
    dt = time.perf_counter() - t0
    return {"value": round(ns / dt, 1), "unit": "points/s", "cores": min(oc.THREADS, 16), "kind": "port",
            "sample": f"the first 2^{ns.bit_length() - 1} points of the same MSM ({dt:.1f} s); own C restatement (signed 16-bit windows, "
            "one thread per window)", "host_cpus": os.cpu_count(), "host_cores": host_cores()[1]}


# ---------------------------------------------------------------------------------------------------------------
# workload: tree20 (BASELINE.json configs[4]) -- MiMC7 Merkle tree over 2^log_n leaves
# ---------------------------------------------------------------------------------------------------------------

def run_tree(args, dist, ctx):
    import numpy as np
    from owshen_amd import shard
    rank, world = dist.rank, dist.world
    log_n = args.log_n or 20
    n = 1 << log_n
    assert world & (world - 1) == 0 and n % world == 0, "tree20: the number of GPUs must be a power of two"
    rng = np.random.Generator(np.random.PCG64(2))
    leaves = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    leaves[:, 31] &= 0x1F
    lo, hi = shard.partition(n, world, rank)
    local = ctx.to_device(leaves[lo:hi])

    def step():
        return shard.tree_build_sharded(ctx, local)[0]

    for _ in range(args.warmup):
        step()
    import torch
    tele = GpuTelemetry(device_identity(torch, dist.device).get("pci"), period=0.005).start()
    dt, root = timed(dist, step, 0, args.steps)
    telemetry = tele.stop()
    ms = dt / args.steps * 1e3
    step_ms = list(dist.last_step_ms)
    alg = 32 * n + 32 * (n - 1)
    cpu, check = None, None
    if rank == 0:
        os.environ.setdefault("OG_ORACLE_NATIVE", "1")
        from oracle.c import binding as oc
        t0 = time.perf_counter()
        want = oc.mimc7_tree_build(leaves)[-1].tobytes()
        t_cpu = time.perf_counter() - t0
        assert bytes(root) == want, "GPU tree root differs from the C restatement"
        check = "root == the C restatement's root over the same 2^%d leaves" % log_n
        if not args.no_cpu:
            cpu = {"value": round((n - 1) / t_cpu, 1), "unit": "hashes/s", "cores": oc.THREADS, "kind": "port",
                   "sample": f"the whole 2^{log_n}-leaf tree ({t_cpu:.1f} s); own C restatement", "host_cpus": os.cpu_count(), "host_cores": host_cores()[1]}
    if rank != 0:
        return None
    return {
        "metric": "MiMC7 Merkle tree rebuild (2^%d leaves): hashes/sec" % log_n, "value": round((n - 1) / (ms * 1e-3), 1), "unit": "hashes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[4]: full MiMC7 (91 rounds, MultiMiMC7 2-to-1) Merkle tree over 2^{log_n} leaves resident in HBM",
                   "n_leaves": n, "parallelism": "1 GPU" if world == 1 else f"subtree per rank, all-gather of {world} roots ({dist.backend}), "
                   "top levels rehashed on every rank", "parity": check},
        "roofline": {"bound": "hbm", "kernel": "k_mimc7_tree_level (all levels)", "achieved": round(alg / (ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": None, "algorithmic_bytes": alg,
                     "note": "32 B per leaf read + 32 B per node written; 728 Fr multiplications per hash: VALU-bound"},
        "step_ms": dict(step_stats(step_ms) or {}, each=step_ms),
        "box": {k: (telemetry or {}).get(k) for k in ("sclk_MHz", "socket_power_W", "temp_C", "samples", "source")},
        "cpu_baseline": cpu,
    }


# ---------------------------------------------------------------------------------------------------------------
# workload: plumbing (BASELINE.json configs[0]) -- ONE withdraw proof on the CPU, verified, no GPU anywhere
# ---------------------------------------------------------------------------------------------------------------

def run_plumbing(args):
    """BASELINE.json configs[0]: "single withdraw proof (Merkle depth 32) on the reference Rust CPU prover, verified vs the
    contracts/ Solidity verifier (plumbing, no GPU)".  The reference has neither, so the stand-ins are this repo's
    own test infrastructure and its one CPU-side product piece -- and NO HIP library is loaded, torch is not imported:
      statement + witness   oracle/py/withdraw.py (the spec), the natural depth-`depth` circuit
      key                   oracle/py/keygen.py: scalars from the Python oracle, fixed-base multiples from the C restatement,
                            bytes in the product's OWPK0001 / OWVK0001 formats
      prover                the C restatement (oracle/c), from that blob
      verifier              og_verify in libowshen_verify.so (the PRODUCT's verifier, host-only build: the `burn_tx` seam,
                            /root/reference/src/blockchain/tx/burn_tx.rs:11) and the WithdrawVerifier.sol word model
                            (oracle/py/evm_model.py: /root/reference/contracts/src/Owshen.sol:66-78's replacement) on the words
                            owshen_amd/evm.py emits
    The line's `value` is the CPU prover's proofs/s on this one statement: a plumbing number, not a baseline."""
    import random
    t_all = time.perf_counter()
    os.environ.setdefault("OG_ORACLE_NATIVE", "1")
    from oracle.c import binding as oc
    from oracle.py import evm_model, fields, keygen, mimc7, withdraw as spec
    from owshen_amd import evm, verify_only
    depth = args.depth
    rnd = random.Random(0x2A)
    leaves = {i: mimc7.hash2(20241008, i) for i in (0x2A, 0x2B)}     # leaves = MiMC7(seed || i); the path of index 0x2A
    nullifier, secret, amount, token, chain_id = rnd.randrange(fields.R), rnd.randrange(fields.R), 10 ** 18, rnd.randrange(1 << 160), 1387
    recipient = rnd.randrange(1 << 160)
    index = 0x2A & ((1 << depth) - 1)
    siblings = [leaves[0x2B]] + [rnd.randrange(fields.R) for _ in range(depth - 1)]
    t0 = time.perf_counter()
    n_wires, n_pub, cons, z = spec.build(depth, nullifier, secret, amount, recipient, index, siblings, token=token, chain_id=chain_id)
    t_spec = time.perf_counter() - t0
    t0 = time.perf_counter()
    pk_blob, vk_blob = keygen.setup_blobs(n_wires, n_pub, cons, *TOXIC)
    t_key = time.perf_counter() - t0
    ck = oc.prepared_key_from_blob(pk_blob)
    wit = b"".join(int(v).to_bytes(32, "little") for v in z)
    import numpy as np
    wit = np.frombuffer(wit, dtype=np.uint8).reshape(n_wires, 32)
    r, s_ = rnd.randrange(fields.R), rnd.randrange(fields.R)
    ck.prove(wit, r, s_)                                             # warm (thread pool, page faults)
    k, t0 = 0, time.perf_counter()
    while k < 3 or time.perf_counter() - t0 < 2.0:
        proof = ck.prove(wit, r, s_)
        k += 1
    t_prove = (time.perf_counter() - t0) / k
    public = z[1:1 + n_pub]
    t0 = time.perf_counter()
    ok_lib = verify_only.verify(vk_blob, public, proof)
    t_verify = time.perf_counter() - t0
    bad_lib = verify_only.verify(vk_blob, [public[0]] + [(public[1] + 1) % fields.R] + public[2:], proof)
    vkw, pw, iw = evm.vk_to_evm_words(vk_blob), evm.proof_words(proof), evm.public_inputs_to_evm_words(public)
    t0 = time.perf_counter()
    ok_evm = evm_model.verify_proof_model(vkw, pw, iw)
    t_evm = time.perf_counter() - t0
    bad_evm = evm_model.verify_proof_model(vkw, pw, iw[:2] + [(iw[2] + 1) % (1 << 160)] + iw[3:])       # another recipient
    if not (ok_lib and ok_evm) or bad_lib or bad_evm:
        sys.exit(f"bench.py plumbing: verifier verdicts are wrong (og_verify {ok_lib}/{bad_lib}, contract model {ok_evm}/{bad_evm})")
    loaded = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if ".so" in ln})
    hip = [x for x in loaded if any(t in os.path.basename(x) for t in ("amdhip", "libhsa", "rccl", "libowshen_gpu", "libtorch", "libc10"))]
    if hip or "torch" in sys.modules:
        sys.exit(f"bench.py plumbing: a GPU-side library was loaded ({hip or 'torch'}) -- this workload must run without one")
    return {
        "metric": "single withdraw proof on the CPU prover, verified (plumbing, no GPU)", "value": round(1.0 / t_prove, 4), "unit": "proofs/s",
        "n_gpus": 0, "steps": k, "warmup": 1, "ms_per_step": round(t_prove * 1e3, 2), "higher_is_better": True, "scaling": "none",
        "vs_baseline": None, "dtype": "u64 (4 x 64-bit-limb Montgomery, the C restatement)", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[0]: ONE withdraw proof, natural depth-{depth} MiMC7 Merkle statement ({n_wires} wires, "
                   f"{len(cons)} constraints, 6 public inputs), proved by the C restatement of the prover (the reference has none), accepted by "
                   "og_verify (libowshen_verify.so: the product's CPU verifier, no ROCm) and by the WithdrawVerifier.sol word model; no GPU, no "
                   "HIP library, no torch in the process",
                   "n_wires": n_wires, "n_constraints": len(cons), "merkle_depth": depth, "leaf_index": index,
                   "key_bytes": len(pk_blob), "host_cpus": os.cpu_count(), "cpu_model": cpu_model(), "prover_threads": oc.THREADS},
        "plumbing": {"accepted_by_og_verify": bool(ok_lib), "accepted_by_contract_model": bool(ok_evm),
                     "refused_with_another_nullifier_hash": not bad_lib, "refused_with_another_recipient": not bad_evm,
                     "seconds": {"statement_and_witness": round(t_spec, 2), "key_generation": round(t_key, 2), "prove": round(t_prove, 3),
                                 "og_verify": round(t_verify, 3), "contract_model": round(t_evm, 2), "total": round(time.perf_counter() - t_all, 1)},
                     "gpu_libraries_loaded": hip, "shared_objects": [os.path.basename(x) for x in loaded if "owshen" in x or "oracle" in x]},
        "roofline": None,
        "cpu_baseline": {"value": round(1.0 / t_prove, 4), "unit": "proofs/s", "cores": min(oc.THREADS, os.cpu_count() or 1), "kind": "port",
                         "sample": f"{k} proofs of this one statement; own C restatement -- the reference has no prover (SURVEY.md 0.1)"},
    }


def main():
    args = parse_args()
    if args.workload == "plumbing":   # (before anything imports torch or the GPU library)
        print(json.dumps(run_plumbing(args)), flush=True)
        return
    ensure_ranks(args)
    dist = Dist(args)
    from owshen_amd import api
    ctx = api.Context(dist.device)
    out = {"prove": run_prove, "msm26": run_msm, "tree20": run_tree}[args.workload](args, dist, ctx)
    if dist.rank == 0 and dist.world > dist.torch.cuda.device_count():
        out["config"]["oversubscribed"] = f"{dist.world} ranks on {dist.torch.cuda.device_count()} GPU(s): a dry run of the N-rank code path, not a measurement"
    if dist.rank == 0:
        print(json.dumps(out), flush=True)
    dist.close()


if __name__ == "__main__":
    main()
