#!/usr/bin/env python3
"""Headline benchmark: particle-pair interactions per second of the direct-summation KDK step.

    python bench.py --gpus N --steps K --warmup W

For N > 1 the command launches its own N ranks (one CHILD process per GPU with RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set,
started before this process has touched torch or the GPU; a failing rank makes the command fail) unless it already runs under a
launcher (WORLD_SIZE set — the driver's `python -m torch.distributed.run ... bench.py --gpus N` form). Extra switches:
`--general-mass` times the 12-op per-source-mass kernel instead of the equal-mass variant, `--dtype f64 --n 524288` the fp64
kernel, `--backend gloo --same-device` rehearses the N-rank bench on ONE GPU (tests/test_gpu_bench_contract.py).

Workload (BASELINE.json metric): Plummer sphere, N = 2^20 particles, fp32, Hénon units, equal
masses, eps = 0.01, dt = 1/256, seed 0 — synthetic, generated on the host and resident in HBM before the timed
region. One "step" is one kick-drift-kick step = one all-pairs force evaluation (N^2 ordered interactions) plus
the O(N) kick/drift kernels and, for N > 1 GPUs, the ring exchange of source chunks (sink decomposition, strong
scaling: the total problem is fixed, every rank owns N/P sinks).

N > 1 times BOTH schedules of the sharded force evaluation (bachelor_thesis_n_body_amd/dist.py) in the one invocation, W warm-up +
exactly K timed steps each, on the same ranks and state: first the one-sided ring SURVEY.md §8(e) names (P-1 position hops, no
accelerations travel), then the pair-symmetric schedule (P//2 hops, partial sums returned). The contract's keys (`value`,
`ms_per_step`, `ring`, `roofline.*`) are the FASTER one's and say which (`roofline.schedule`); `roofline.one_sided` and
`roofline.pair_symmetric` hold both (ms_per_step, kernel_ms_per_step, ring incl. the return leg's back_comm_ms / back_exposed_ms /
add_ms). `--schedule` restricts the run to one of them. Every rank runs a stall watchdog (NB_BENCH_STALL_S, default 120 s without a
progress mark): if a schedule stops advancing after another one completed, rank 0 still prints the line for the completed one with
`roofline.stalled` naming where the other stopped, and the ranks exit; a self-launched run is also supervised by the parent
(per-rank heartbeat files; all ranks killed by PID and exit code 4 if none advances).

Rank 0 prints ONE JSON line. Besides the contract's keys it carries
  roofline     — the dominant kernel (accel_sym_f32_kernel) against the gfx950 vector-FP32 peak, at the conventional
                 20 flop per interaction; durations from HIP events on the launch stream
  cpu_baseline — the C oracle (oracle/nbody_oracle.c, OpenMP, fp64) timed on this host on a bounded sample of the
                 same workload (the first M sinks against all N sources), rank 0 / 1 GPU only; `numpy`: the NumPy oracle at
                 N = 16 384 (SURVEY.md §8d leg (i)); `host`: CPU model, logical CPUs, threads used.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FLOP_PER_INTERACTION = 20.0        # Nyland-Harris-Prins convention, SURVEY.md §8(d)
PEAK_FP32_VECTOR_TFLOPS = 157.3    # MI355X_MICROARCH.md "Peak FP32 (vector)"
PEAK_FP64_VECTOR_TFLOPS = 78.6
PEAK_CLOCK_MHZ = 2400.0            # the clock both peaks are quoted at (256 CUs x 256 flop/clk x 2.4 GHz)
EXECUTED_FLOP_ONE_SIDED = 19.0     # one-sided kernels, per ordered interaction: 3 sub, 3 fma, rsqrt, 3 mul, 3 fma (equal masses: 18)
EXECUTED_FLOP_SYM = 12.0           # pair-symmetric kernels: 24 per unordered pair = 12 per delivered (ordered) interaction


def measured_traffic():
    """HBM bytes per accel launch from the committed rocprofv3 PMC passes of THIS command (tools/profile.sh ->
    profiles/r04_bench_n1_uniform_rocprofv3_summary.txt: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE). bench.py
    cannot run the profiler around itself, so the number is the profile's, or None if the file is absent."""
    f = ROOT / "profiles" / "r04_bench_n1_uniform_rocprofv3_summary.txt"
    try:
        rd = wr = None
        for line in f.read_text().splitlines():
            if line.startswith("HBM read bytes/launch"):
                rd = float(line.split("~")[1].split()[0])
            if line.startswith("HBM write bytes/launch"):
                wr = float(line.split("~")[1].split()[0])
        return None if rd is None or wr is None else rd + wr
    except OSError:
        return None


def _is_profiler_var(key, value):
    """Environment a rocprofv3 / rocprofiler-sdk session injects into the program it profiles."""
    if key.startswith(("ROCPROF", "ROCPROFILER_", "ROCP_", "ROCTX_")) or key in ("HSA_TOOLS_LIB", "HSA_TOOLS_REPORT_LOAD_FAILURE"):
        return True
    return key == "LD_PRELOAD" and "rocprof" in value.lower()


def under_profiler():
    return any(_is_profiler_var(k, v) for k, v in os.environ.items())


def live_traffic(args, timeout_s=150):
    """HBM bytes per accel launch of THIS workload, measured now: two child runs of this script (one step, no CPU baseline) under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, counters only — the guide's prescription), FETCH_SIZE doubled
    (gfx950 reports half the bytes of wide coalesced reads: MI355X_MICROARCH.md §HBM). Children are started as subprocesses (never an exec
    from this GPU-initialised process). Returns (bytes, note) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    if under_profiler():
        return None, "already under a profiler (nested counter sessions are not started)"
    vals = {}
    child_env = {k: v for k, v in os.environ.items() if not _is_profiler_var(k, v)}
    child_env["TMPDIR"] = "/tmp"
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="nb_pmc_", dir="/tmp")
        cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, str(Path(__file__).resolve()),
               "--steps", "1", "--warmup", "0", "--n", str(args.n), "--dtype", args.dtype, "--no-cpu-baseline", "--traffic", "none",
               "--no-clock-probe"]        # counter collection serialises kernels: the probe wave would hold the GPU for its whole window
        if args.general_mass:
            cmd.append("--general-mass")
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=child_env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} exited with {r.returncode}"
            per = []
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if ("accel_f" in row["Kernel_Name"] or "accel_sym_f" in row["Kernel_Name"]) and "reduce" not in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        per.append(float(row["Counter_Value"]))
            if not per:
                return None, f"no {counter} rows for the accel kernel"
            vals[counter] = sum(per) / len(per)
        except (subprocess.TimeoutExpired, OSError) as e:
            return None, f"rocprofv3 --pmc {counter}: {e}"
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate child passes of this command, 1 step), KiB x 1024, FETCH x2 (gfx950)"


class KernelTimer:
    """HIP-event pairs around every accel launch, recorded on the stream the kernels run on (torch's current
    stream, which the engine binds with nb_set_stream)."""

    def __init__(self, torch):
        self.torch, self.pairs, self.enabled = torch, [], False

    def begin(self):
        if self.enabled:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record()
            self._open = e

    def end(self, interactions):
        if self.enabled:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record()
            self.pairs.append((self._open, e, interactions))

    # One rank: ONE pair around the K steps of the timed region instead of a pair per launch — two event records per step are themselves
    # 10-20 us of a step (N = 32 768: a 0.22 ms step), and the steps of one call are two launches each with nothing of the host between them.
    def span_begin(self):
        self._span = self.torch.cuda.Event(enable_timing=True)
        self._span.record()

    def span_end(self, launches, interactions):
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        self.span = (self._span, e, int(launches), interactions)

    def summary(self):
        if getattr(self, "span", None) is not None:
            a, b, k, inter = self.span
            return (a.elapsed_time(b) / max(1, k), inter, k)
        ms = [a.elapsed_time(b) for a, b, _ in self.pairs]
        inter = [i for _, _, i in self.pairs]
        return (sum(ms) / len(ms), sum(inter) / len(inter), len(ms)) if ms else (0.0, 0.0, 0)


def host_identity(omp_threads=None):
    """The host the CPU legs ran on: logical CPUs, what this process may use, the model string, and the OpenMP thread count of the C leg."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return {"cpu_count": os.cpu_count(), "cpus_available_to_this_process": avail, "model": model, "omp_threads": omp_threads,
            "omp_num_threads_env": os.environ.get("OMP_NUM_THREADS")}


def cpu_baseline_numpy(state, eps, n_src=16384, budget_s=8.0):
    """SURVEY.md §8(d) leg (i): the NumPy fp64 oracle (oracle/nbody_oracle.py: chunked einsum, whatever threads NumPy's BLAS-free
    kernels use = 1) on the first `n_src` particles of the workload; interactions/s of an O(N^2) kernel does not depend on N, so a
    bounded number of sinks is timed (about `budget_s` seconds) and the sample says so."""
    try:
        from oracle import nbody_oracle as ora
    except ImportError:
        return None
    n_src = min(n_src, state.n)
    pos = np.ascontiguousarray(state.pos[:n_src], dtype=np.float64)
    mass = np.ascontiguousarray(state.mass[:n_src], dtype=np.float64)
    probe = min(256, n_src)
    t0 = time.perf_counter()
    ora.compute_forces(pos, mass, eps, sinks=pos[:probe])
    rate = probe * n_src / max(time.perf_counter() - t0, 1e-9)
    m_snk = int(min(n_src, max(probe, budget_s * rate / n_src)))
    t0 = time.perf_counter()
    ora.compute_forces(pos, mass, eps, sinks=pos[:m_snk])
    t = time.perf_counter() - t0
    return {"value": m_snk * n_src / t, "unit": "interactions/s", "cores": 1, "kind": "port",
            "sample": f"NumPy oracle fp64 (oracle/nbody_oracle.py): first {m_snk} sinks x first {n_src} sources of the same Plummer workload, {t:.1f} s"}


def cpu_baseline(state, eps, budget_s=12.0, numpy_budget_s=8.0):
    """C oracle, fp64, OpenMP over sinks: first M sinks of the workload against all N sources (SURVEY.md §8(d) leg (ii): the host C build
    of the same kernel on the cores one GPU's share of the host has). `numpy`: leg (i), the NumPy oracle at N = 16 384. `host`: which
    machine that was."""
    so = ROOT / "oracle" / "libnbody_oracle.so"
    if not so.exists():
        return None
    lib = C.CDLL(str(so))
    # one GPU's share of the host: 16 cores (libgomp is already initialised by torch, so set it explicitly)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(16, avail))
    lib.nbo_set_threads(C.c_int(threads))
    threads = int(lib.nbo_max_threads())
    p = np.ascontiguousarray(state.pos.T.astype(np.float64))
    m = np.ascontiguousarray(state.mass.astype(np.float64))
    n = p.shape[1]
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def run(m_snk):
        a = np.empty((3, m_snk))
        t0 = time.perf_counter()
        lib.nbo_accel_f64_omp(C.c_int64(m_snk), vp(p[0]), vp(p[1]), vp(p[2]), C.c_int64(n), vp(p[0]), vp(p[1]), vp(p[2]),
                              vp(m), C.c_double(eps), C.c_double(1.0), vp(a[0]), vp(a[1]), vp(a[2]))
        return time.perf_counter() - t0

    probe = max(threads * 4, 64)
    run(probe)                                  # warm the threads
    t = run(probe)
    rate = probe * n / t
    m_snk = int(min(n, max(probe, (budget_s * rate / n) // threads * threads)))
    t = run(m_snk)
    out = {"value": m_snk * n / t, "unit": "interactions/s", "cores": threads, "kind": "port",
           "sample": f"C oracle fp64 OpenMP: first {m_snk} sinks x all {n} sources of the same Plummer workload, {t:.1f} s",
           "host": host_identity(threads)}
    if numpy_budget_s and numpy_budget_s > 0:
        out["numpy"] = cpu_baseline_numpy(state, eps, budget_s=numpy_budget_s)
    return out


class Heartbeat:
    """Progress marks of one rank. Every call stamps the in-process clock (read by the rank's own `StallWatchdog` thread) — two
    assignments, nothing else on the timed path. When the parent of a self-launched run asked for it (NB_BENCH_HEARTBEAT_DIR) a daemon
    thread rewrites a small per-rank file with the label of the latest mark whenever it has changed (every `flush_s` seconds), whose
    modification time and content the parent watches."""

    def __init__(self, rank, directory=None, flush_s=0.2):
        self.rank, self.label, self.t = rank, "start", time.monotonic()
        directory = directory if directory is not None else os.environ.get("NB_BENCH_HEARTBEAT_DIR")
        self.path = os.path.join(directory, "rank%d" % rank) if directory else None
        if self.path:
            import threading
            self._write("start")
            threading.Thread(target=self._flush, args=(flush_s,), name="nb-bench-heartbeat", daemon=True).start()

    def __call__(self, label):
        self.label, self.t = label, time.monotonic()

    def _write(self, label):
        try:
            with open(self.path, "w") as f:
                f.write(label + "\n")
        except OSError:
            pass

    def _flush(self, every):
        written = "start"
        while True:
            time.sleep(every)
            label = self.label
            if label != written:
                self._write(label)
                written = label


class StallWatchdog:
    """A daemon thread beside the rank's main thread: when the heartbeat has not moved for `stall_s` seconds (a collective every rank is
    alive in but none leaves: a schedule bug, not a crash) it calls `on_stall(label)` — which reports what was measured before the stall —
    and ends the process with the code that returns (os._exit: the main thread sits in a device synchronise that will never return)."""

    def __init__(self, heartbeat, stall_s, on_stall):
        import threading
        self.hb, self.stall_s, self.on_stall = heartbeat, float(stall_s), on_stall
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, name="nb-bench-watchdog", daemon=True)
        self._th.start()

    def _run(self):
        while not self._stop.wait(min(1.0, self.stall_s / 4.0)):
            if time.monotonic() - self.hb.t > self.stall_s:
                code = 3
                try:
                    code = int(self.on_stall(self.hb.label))
                finally:
                    sys.stdout.flush()
                    sys.stderr.flush()
                    os._exit(code)

    def stop(self):
        self._stop.set()


def supervise(procs, heartbeat_dir, stall_s, poll_s=0.05):
    """Wait for the rank processes `procs` (subprocess.Popen). Returns the exit code of the command: 0 when every rank returned 0; the first
    failing rank's code otherwise (the surviving ranks are then terminated by PID, not left waiting in a collective); 4 when no rank's
    heartbeat file advanced for `stall_s` seconds although all are alive — an all-alive deadlock: every rank is killed (by PID; these are
    this process' own children) and the last mark each one reached is printed on stderr."""
    rc, live = 0, list(procs)
    last_change, seen = time.monotonic(), {}
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            last_change = time.monotonic()
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in live:                                  # a dead rank leaves the others stuck in a collective
                    q.terminate()
        if heartbeat_dir and stall_s and live:
            for r in range(len(procs)):
                try:
                    m = os.stat(os.path.join(heartbeat_dir, "rank%d" % r)).st_mtime_ns
                except OSError:
                    continue
                if seen.get(r) != m:
                    seen[r], last_change = m, time.monotonic()
            if time.monotonic() - last_change > stall_s:
                marks = []
                for r in range(len(procs)):
                    try:
                        marks.append("rank %d: %s" % (r, open(os.path.join(heartbeat_dir, "rank%d" % r)).read().strip()))
                    except OSError:
                        marks.append("rank %d: no heartbeat" % r)
                print("bench.py: no rank advanced for %.0f s — killing all ranks. Last marks: %s" % (stall_s, "; ".join(marks)), file=sys.stderr, flush=True)
                for q in live:
                    q.kill()
                for q in live:
                    q.wait()
                return rc or 4
        time.sleep(poll_s)
    return rc


def stall_seconds():
    """Seconds without progress after which a multi-rank run is declared stalled (NB_BENCH_STALL_S; default 120: RCCL's first connection
    set-up and the generation of the workload are slow, a step is milliseconds)."""
    try:
        return max(1.0, float(os.environ.get("NB_BENCH_STALL_S", "120")))
    except ValueError:
        return 120.0


def self_launch(args, argv=None, child=None) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (never an exec; nothing in THIS
    process has imported torch or touched the GPU) with the environment torch.distributed.run would give them (RANK,
    LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT), let rank 0's JSON line through on stdout and return
    non-zero if any rank failed or — `supervise` — if all ranks are alive but none advances (each rank also runs its own
    StallWatchdog, which fires first and still reports what it had measured; the parent's limit is 30 s longer)."""
    import shutil
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    hb_dir = tempfile.mkdtemp(prefix="nb_bench_hb_")
    procs = []
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NB_BENCH_HEARTBEAT_DIR=hb_dir)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this host driver
            env.setdefault("OMP_NUM_THREADS", "1")
            procs.append(subprocess.Popen([sys.executable, child or str(Path(__file__).resolve())] + (sys.argv[1:] if argv is None else list(argv)), env=env))
        return supervise(procs, hb_dir, stall_seconds() + float(os.environ.get("NB_BENCH_PARENT_MARGIN_S", "30")))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(hb_dir, ignore_errors=True)


def single_process(args) -> int:
    """The same workload through nb_create_multi (include/nbody_amd.h): one process, N devices, every device pulls the other owners'
    source blocks peer to peer while its force kernel works on the block before. No torch in this path."""
    import bachelor_thesis_n_body_amd as nb
    from bachelor_thesis_n_body_amd import _lib
    ndev = _lib.load().nb_device_count()
    if ndev <= 0:
        print("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)", file=sys.stderr)
        return 1
    devices = [0] * args.gpus if args.same_device else list(range(args.gpus))
    if max(devices) >= ndev:
        print(f"--gpus {args.gpus}: only {ndev} device(s) visible; use --same-device to rehearse on one GPU", file=sys.stderr)
        return 1
    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    eps, dt, G = 0.01, 1.0 / 256.0, 1.0
    state = nb.plummer_sphere(args.n, seed=0, dtype=np_dtype)
    if args.general_mass:
        state.mass[-1] = np.nextafter(state.mass[-1], np_dtype(1.0))
    with nb.Engine(args.n, np_dtype, devices=devices) as eng:
        eng.set_state(state)
        eng.accel(eps, G, fetch=False)
        for _ in range(args.warmup):
            eng.step(dt, 1, eps, G)
        t0 = time.perf_counter()
        eng.step(dt, args.steps, eps, G)          # ONE call: no host synchronisation between the steps; returns synchronised
        elapsed = time.perf_counter() - t0
        symmetric = eng.last_accel_symmetric()                  # the own-chunk launch of device 0
        pairwise = bool(eng.get_option("multi_schedule"))       # the schedule the context runs: its option, not a guess from the last launch
    value = float(args.n) * float(args.n) * args.steps / elapsed
    peak = PEAK_FP32_VECTOR_TFLOPS if args.dtype == "f32" else PEAK_FP64_VECTOR_TFLOPS
    achieved = value * FLOP_PER_INTERACTION / 1e12 / len(devices)
    out = {"metric": "particle-pair interactions/s", "value": value, "unit": "interactions/s", "n_gpus": args.gpus, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"plummer_sphere N={args.n} {args.dtype} eps=0.01 dt=1/256, KDK step = 1 all-pairs force evaluation", "n": args.n,
                      "parallelism": f"nb_create_multi: one process, sink-decomposition x{args.gpus}, peer pulls of source blocks",
                      "masses": "general (one mass perturbed by 1 ulp)" if args.general_mass else "equal"},
           "world_size": args.gpus, "backend": "hipMemcpyPeerAsync (single process)", "same_device": bool(args.same_device), "ring": None,
           "roofline": {"bound": "vector_" + ("fp32" if args.dtype == "f32" else "fp64"),
                        "kernel": ("accel_sym_%s_kernel" if symmetric else "accel_%s_kernel") % args.dtype, "pair_symmetry": bool(symmetric),
                        "pair_symmetry_scope": None if not symmetric else ("every launch" if (pairwise or args.gpus == 1) else "own-chunk launch of each device only"),
                        "schedule": None if args.gpus == 1 else ("pair-symmetric: each pair of chunks worked on by one owner, partial sums pushed to the other" if pairwise else "one-sided: every device sums all chunks for its own sinks"),
                        "variant": "general" if args.general_mass else "uniform_mass", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                        "frac": achieved / peak, "flop_per_interaction": FLOP_PER_INTERACTION,
                        "note": "per device, from the whole step (exchange and O(N) kernels included); kernel events are not visible from outside the library",
                        "traffic": None}}
    print(json.dumps(out), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--mode", choices=["ring", "allgather"], default="ring")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="nccl = RCCL over xGMI (default); gloo for rehearsals")
    ap.add_argument("--same-device", action="store_true", help="all ranks share cuda:0 (one-GPU rehearsal; needs --backend gloo)")
    ap.add_argument("--general-mass", action="store_true", help="perturb one mass by 1 ulp: times the per-source-mass kernel")
    ap.add_argument("--single-process", action="store_true",
                    help="drive the N GPUs from THIS process through the C-ABI's multi-device context (nb_create_multi: peer pulls of the "
                         "source blocks, no torch.distributed); with --same-device all N shards sit on GPU 0 (rehearsal)")
    ap.add_argument("--one-sided-schedule", action="store_true",
                    help="N > 1: every rank sums all N chunks one-sidedly for its own sinks (rounds 1-2) instead of the pair-symmetric schedule (A/B)")
    ap.add_argument("--schedule", choices=["both", "pair-symmetric", "one-sided"], default="both",
                    help="N > 1: which schedule(s) of the force evaluation to time (dist.py). 'both' (default): the one-sided ring, then the "
                         "pair-symmetric schedule, on the same ranks and state — the line's contract keys are the faster one's, roofline.one_sided / "
                         ".pair_symmetric hold both")
    ap.add_argument("--inject-stall", default=None, help=argparse.SUPPRESS)      # test hook: <rank>:<region>:<step>
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clock-probe", action="store_true",
                    help="do not run the sustained-clock probe beside the timed region (for runs under rocprofv3 --pmc, which serialises kernels)")
    ap.add_argument("--traffic", choices=["live", "profile", "none"], default="live",
                    help="roofline.traffic at N = 1: 'live' = measure HBM bytes per launch now with two rocprofv3 --pmc child passes (~25 s), "
                         "falling back to the committed profile; 'profile' = the committed profile's figure; 'none' = null")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.same_device and args.backend == "nccl" and args.gpus > 1 and not args.single_process:
        raise SystemExit("--same-device needs --backend gloo (RCCL refuses two ranks on one device)")

    if args.single_process:
        raise SystemExit(single_process(args))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist
    import bachelor_thesis_n_body_amd as nb
    from bachelor_thesis_n_body_amd.dist import HopProbe

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} disagrees with WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no device cuda:{local_rank} ({torch.cuda.device_count()} visible); use --same-device --backend gloo to rehearse on one GPU")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    # ---- progress marks and the stall watchdog (N > 1 only: a deadlocked collective would otherwise sit until the caller's time limit) ----
    hb = Heartbeat(rank) if world > 1 else (lambda *a, **k: None)
    done = {}                      # schedule name -> measured region (filled as regions complete; read by the watchdog)
    emit = {"fn": None}            # set once the JSON builder exists

    def on_stall(label):
        # every rank completed the same regions (they end with a barrier + all-reduce), so every rank takes the same exit code
        if done and emit["fn"] is not None:
            if rank == 0:
                print(f"bench.py: rank 0 made no progress for {stall_seconds():.0f} s at '{label}': reporting the region(s) that completed", file=sys.stderr, flush=True)
                emit["fn"](stalled_at=label)
            else:
                time.sleep(5.0)    # rank 0 prints first
            return 0
        print(f"bench.py: rank {rank} made no progress for {stall_seconds():.0f} s at '{label}' and no region had completed", file=sys.stderr, flush=True)
        return 3
    watchdog = StallWatchdog(hb, stall_seconds(), on_stall) if world > 1 else None

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    hb("process group up")

    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    eps, dt, G = 0.01, 1.0 / 256.0, 1.0
    state = nb.plummer_sphere(args.n, seed=0, dtype=np_dtype)
    if args.general_mass:       # one mass 1 ulp up: same physics to ~1e-7/N, but no longer the equal-mass special case
        state.mass[-1] = np.nextafter(state.mass[-1], np_dtype(1.0))
    hb("workload generated")

    # ---- which schedule(s) of the N-rank force evaluation are timed (dist.py): both by default, the one-sided ring FIRST ----
    # (it is the schedule SURVEY.md §8(e) names and the simpler message pattern; if the pair-symmetric one then stalled on real links the
    # run would still report a measured number)
    schedule = "one-sided" if args.one_sided_schedule else args.schedule
    if world == 1:
        plan = [None]
    elif os.environ.get("NB_PAIRWISE_SCHEDULE", "1") == "0" or schedule == "one-sided":
        plan = [False]
    elif schedule == "pair-symmetric":
        plan = [True]
    else:
        plan = [False, True]
    eng = nb.ShardedEngine(args.n, np_dtype, device=device, mode=args.mode, pairwise=plan[0])
    eng.heartbeat = hb if world > 1 else None
    eng.set_state(state)
    hb("state resident")

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    inject = None
    if args.inject_stall:          # test hook: "<rank>:<region index>:<step>" — that rank stops taking part there (tests/test_gpu_bench_contract.py)
        inject = tuple(int(v) for v in args.inject_stall.split(":"))

    def timed_region(idx, pairwise):
        name = "single" if pairwise is None else ("pair-symmetric" if pairwise else "one-sided")
        if pairwise is not None:
            eng.set_schedule(pairwise)
        timer = KernelTimer(torch)
        eng.kernel_timer = timer
        probe = HopProbe(torch, device) if (world > 1 and args.mode == "ring") else None
        eng.hop_probe = probe
        eng.accel(eps, G)                      # initial a(x): not part of a step
        for w in range(args.warmup):
            eng.step(dt, 1, eps, G)
            hb(f"{name}: warm-up step {w + 1}")
        sync()
        hb(f"{name}: warm-up done")
        # sustained shader clock of this rank's GPU over the timed region (one sleeping wave on a private stream, nb_clock_probe_*; its window is
        # ended by a word written on the compute stream behind the last step, so the synchronisation below never waits for it):
        # boxes differ by several per cent in the clock they hold under this kernel; cycles per interaction is the box-independent figure
        clock = None if args.no_clock_probe else nb.ClockProbe(local_rank, 5000.0)
        timer.enabled = True
        if probe is not None:
            probe.enabled = True
        t0 = time.perf_counter()
        if world == 1 and inject is None:
            # one rank: the K steps as ONE call, as a run would make it — between the steps of a call the kernel that completes the accelerations
            # also applies the closing half kick and the next opening half kick and drift (two launches per step; nb_accel_kdk_dev)
            timer.enabled = False
            timer.span_begin()
            eng.step(dt, args.steps, eps, G)
            timer.span_end(args.steps, float(args.n) * float(args.n))
            hb(f"{name}: {args.steps} steps enqueued")
        else:
            for k in range(args.steps):
                if inject is not None and inject == (rank, idx, k):
                    time.sleep(1e6)
                eng.step(dt, 1, eps, G)
                hb(f"{name}: step {k + 1} of {args.steps} enqueued")
        if clock is not None:
            clock.stop(torch.cuda.current_stream().cuda_stream)
        sync()
        elapsed = time.perf_counter() - t0
        hb(f"{name}: timed region synchronised")
        timer.enabled = False
        if probe is not None:
            probe.enabled = False
        clock_mhz, clock_ms, clock_xcd = clock.end_xcd() if clock is not None else (None, None, None)
        clock_min = clock_mhz
        if world > 1:
            t = torch.tensor([elapsed, -(clock_mhz or 0.0)], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0].item())
            clock_min = -float(t[1].item()) if clock_mhz else None
        symmetric = eng.k.last_accel_symmetric() if world == 1 else eng.own_chunk_symmetric
        k_ms, k_inter, k_count = timer.summary()
        reg = {"name": name, "pairwise": bool(eng.pairwise), "elapsed": elapsed, "symmetric": bool(symmetric),
               "sym_all": bool(symmetric and (world == 1 or eng.pairwise)),      # N ranks: the pair-symmetric schedule makes every hop two-sided
               "k_ms": k_ms, "k_inter": k_inter, "k_count": k_count, "ring": probe.summary() if probe is not None else None,
               "clock_mhz": clock_mhz, "clock_ms": clock_ms, "clock_min": clock_min, "clock_xcd": clock_xcd}
        hb(f"{name}: region complete")
        return reg

    peak = PEAK_FP32_VECTOR_TFLOPS if args.dtype == "f32" else PEAK_FP64_VECTOR_TFLOPS
    interactions = float(args.n) * float(args.n) * args.steps

    def brief(reg):
        """One schedule's timed region in the A/B block of the line."""
        ach = reg["k_inter"] * FLOP_PER_INTERACTION / (reg["k_ms"] * 1e-3) / 1e12 if reg["k_ms"] > 0 else 0.0
        return {"ms_per_step": reg["elapsed"] / args.steps * 1e3, "value": interactions / reg["elapsed"],
                "kernel_ms_per_step": reg["k_ms"] * reg["k_count"] / max(1, args.steps), "launches_timed": reg["k_count"],
                "avg_launch_ms": reg["k_ms"], "interactions_per_launch": reg["k_inter"], "frac": ach / peak,
                "sustained_clock_mhz": reg["clock_mhz"], "ring": reg["ring"]}

    def emit_line(stalled_at=None):
        """Rank 0's ONE JSON line from the regions completed so far: the contract's keys describe the FASTER schedule timed in this run."""
        regs = dict(done)
        best = min(regs.values(), key=lambda g: g["elapsed"])
        elapsed, symmetric, sym_all = best["elapsed"], best["symmetric"], best["sym_all"]
        k_ms, k_inter, k_count = best["k_ms"], best["k_inter"], best["k_count"]
        clock_mhz, clock_ms, clock_min = best["clock_mhz"], best["clock_ms"], best["clock_min"]
        value = interactions / elapsed
        achieved = k_inter * FLOP_PER_INTERACTION / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        out = {
            "metric": "particle-pair interactions/s",
            "value": value,
            "unit": "interactions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"plummer_sphere N={args.n} {args.dtype} eps=0.01 dt=1/256, KDK step = 1 all-pairs force evaluation",
                       "n": args.n, "parallelism": f"sink-decomposition x{world}, {args.mode} exchange of source chunks" + (
                           f" ({world // 2} of {world - 1} hops: each pair of chunks is worked on by one of its owners, partial sums returned point to point)"
                           if (world > 1 and best["pairwise"]) else ""),
                       "masses": "general (one mass perturbed by 1 ulp)" if args.general_mass else "equal"},
            "world_size": world,
            "backend": (args.backend if args.backend == "gloo" else "nccl (RCCL)") if world > 1 else None,
            "same_device": bool(args.same_device) if world > 1 else None,
            # rank 0's ring hops inside the timed region: exchange time vs the kernel it was posted beside, and what of it the
            # compute stream still had to wait for (0 = hidden); pair-symmetric schedule: the return leg too (back_*, add_ms)
            "ring": best["ring"],
            "roofline": {"bound": "vector_" + ("fp32" if args.dtype == "f32" else "fp64"),
                         "kernel": ("accel_sym_%s_kernel" if sym_all else "accel_%s_kernel") % args.dtype,
                         "variant": "uniform_mass" if eng.uniform_mass else "general",
                         # achieved = ORDERED interactions per launch x 20 flop / launch time — the convention the metric is quoted in
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "flop_per_interaction": FLOP_PER_INTERACTION,
                         # the pair-symmetric kernel evaluates every UNORDERED pair once and gives both particles their term (Newton III):
                         # N^2 ordered interactions are DELIVERED per launch, (N^2 + N * set) / 2 pair terms are EVALUATED, at 24 flop each
                         # (3 sub, 3 fma, rsqrt, 2 mul, 6 fma) = 12 flop per delivered interaction instead of the convention's 20
                         "pair_symmetry": bool(symmetric),
                         "pair_symmetry_scope": (None if not symmetric else "every launch" if sym_all else "own-chunk launch of each rank only"),
                         "schedule": (None if world == 1 else "pair-symmetric: each pair of chunks worked on by one owner, partial sums returned" if best["pairwise"] else "one-sided: every rank sums all chunks for its own sinks"),
                         "pair_terms_evaluated_per_launch_approx": (k_inter / 2.0 if sym_all else k_inter),   # PMC count (one v_rsq per term): profiles/r04_*_rocprofv3_summary.txt
                         "executed_flop_per_interaction": EXECUTED_FLOP_SYM if sym_all else EXECUTED_FLOP_ONE_SIDED,
                         "frac_executed_flops": achieved / peak * (EXECUTED_FLOP_SYM if sym_all else EXECUTED_FLOP_ONE_SIDED) / FLOP_PER_INTERACTION,
                         "interactions_per_launch": k_inter, "avg_launch_ms": k_ms, "launches_timed": k_count,
                         "launch_timing": ("one HIP-event pair around the K force evaluations of the timed region (each: force kernel + its reduce / leapfrog-tail kernel; "
                                           "+ one opening kick-drift launch)" if world == 1 and inject is None else "one HIP-event pair per force launch"),
                         # rank 0's accel launches per step, back to back: what ms_per_step would be with a free exchange and free O(N) kernels
                         "kernel_ms_per_step": k_ms * k_count / max(1, args.steps),
                         "traffic": None,
                         "traffic_source": "NOT measured in this run: HBM bytes per launch from the rocprofv3 --pmc passes of this same command committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE: the partial-sum slabs of the pair-symmetric kernel); algorithmic bytes = 28*N"},
        }
        r = out["roofline"]
        if world > 1:
            # the A/B of the two N-rank schedules, each timed for `steps` steps in THIS run on the same ranks and state; the contract's keys
            # above are the faster one's (`schedule`). `stalled`: a schedule that stopped advancing (the watchdog then printed this line)
            r["schedule_selection"] = "faster of the schedules timed in this run (each: warm-up steps, then exactly `steps` timed steps)"
            r["one_sided"] = brief(regs["one-sided"]) if "one-sided" in regs else None
            r["pair_symmetric"] = brief(regs["pair-symmetric"]) if "pair-symmetric" in regs else None
            if stalled_at is not None:
                r["stalled"] = {"at": stalled_at, "no_progress_for_s": stall_seconds(),
                                "schedules_not_completed": [("pair-symmetric" if pw else "one-sided") for pw in plan
                                                            if ("pair-symmetric" if pw else "one-sided") not in regs]}
        if clock_mhz:
            simds = 256 * 4
            r["sustained_clock_mhz"] = clock_mhz            # mean over the XCDs (one probe wave each)
            r["sustained_clock_mhz_per_xcd"] = best["clock_xcd"]
            r["clock_window_ms"] = clock_ms
            if world > 1:
                r["sustained_clock_mhz_min_over_ranks"] = clock_min
            # the peak is quoted at 2 400 MHz; what this box could have reached at the clock it actually held
            r["frac_at_sustained_clock"] = achieved / (peak * clock_mhz / PEAK_CLOCK_MHZ)
            r["cycles_per_64_interactions_per_simd"] = (k_ms * 1e-3 * clock_mhz * 1e6 * simds / (k_inter / 64.0)) if k_inter > 0 else None
        profile_note = r["traffic_source"]
        r["algorithmic_bytes_per_launch"] = (28 if args.dtype == "f32" else 56) * args.n
        if world > 1 or args.traffic == "none":
            r["traffic_source"] = "not collected (N > 1 or --traffic none)"
        else:
            got, note = live_traffic(args) if args.traffic == "live" else (None, "--traffic profile")
            if got is not None:
                r["traffic"], r["traffic_source"] = got, note
            elif args.n == 1 << 20 and args.dtype == "f32" and not args.general_mass:
                r["traffic"], r["traffic_source"] = measured_traffic(), profile_note + f" [live measurement unavailable: {note}]"
            else:
                r["traffic_source"] = f"not available: {note}"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(state, eps)
        print(json.dumps(out), flush=True)

    emit["fn"] = emit_line
    for idx, pairwise in enumerate(plan):
        try:
            reg = timed_region(idx, pairwise)
        except Exception as e:     # a schedule that FAILS after another completed: report what was measured (the other ranks' watchdogs do the same)
            if not done:
                raise
            print(f"bench.py: rank {rank}: schedule {idx} failed: {e!r}", file=sys.stderr, flush=True)
            if rank == 0:
                emit_line(stalled_at=f"exception: {e!r}")
            sys.stdout.flush()
            os._exit(0)            # no destroy_process_group: the peers may be stuck in the failed exchange
        done[reg["name"]] = reg
    if watchdog is not None:
        watchdog.stop()
    if rank == 0:
        emit_line()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
