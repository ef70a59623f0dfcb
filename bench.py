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

Rank 0 prints ONE JSON line. Besides the contract's keys it carries
  roofline     — the dominant kernel (accel_f32_kernel) against the gfx950 vector-FP32 peak, at the conventional
                 20 flop per interaction; durations from HIP events on the launch stream
  cpu_baseline — the C oracle (oracle/nbody_oracle.c, OpenMP, fp64) timed on this host on a bounded sample of the
                 same workload (the first M sinks against all N sources), rank 0 / 1 GPU only.
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
    profiles/r03_bench_n1_uniform_rocprofv3_summary.txt: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE). bench.py
    cannot run the profiler around itself, so the number is the profile's, or None if the file is absent."""
    f = ROOT / "profiles" / "r03_bench_n1_uniform_rocprofv3_summary.txt"
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

    def summary(self):
        ms = [a.elapsed_time(b) for a, b, _ in self.pairs]
        inter = [i for _, _, i in self.pairs]
        return (sum(ms) / len(ms), sum(inter) / len(inter), len(ms)) if ms else (0.0, 0.0, 0)


def cpu_baseline(state, eps, budget_s=12.0):
    """C oracle, fp64, OpenMP over sinks: first M sinks of the workload against all N sources."""
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
    return {"value": m_snk * n / t, "unit": "interactions/s", "cores": threads, "kind": "port",
            "sample": f"C oracle fp64 OpenMP: first {m_snk} sinks x all {n} sources of the same Plummer workload, {t:.1f} s"}


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (never an exec; nothing in THIS
    process has imported torch or touched the GPU) with the environment torch.distributed.run would give them (RANK,
    LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, MASTER_PORT), let rank 0's JSON line through on stdout and return
    non-zero if any rank failed (the surviving ranks are then terminated by PID, not left waiting in a collective)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this host driver
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
    rc, live = 0, list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in live:                                  # a dead rank leaves the others stuck in a collective
                    q.terminate()
        time.sleep(0.05)
    return rc


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
        symmetric = eng.last_accel_symmetric()
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
                        "schedule": "pair-symmetric: each pair of chunks worked on by one owner, partial sums pushed to the other" if symmetric else "one-sided",
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    eps, dt, G = 0.01, 1.0 / 256.0, 1.0
    state = nb.plummer_sphere(args.n, seed=0, dtype=np_dtype)
    if args.general_mass:       # one mass 1 ulp up: same physics to ~1e-7/N, but no longer the equal-mass special case
        state.mass[-1] = np.nextafter(state.mass[-1], np_dtype(1.0))
    eng = nb.ShardedEngine(args.n, np_dtype, device=device, mode=args.mode, pairwise=False if args.one_sided_schedule else None)
    eng.set_state(state)
    timer = KernelTimer(torch)
    eng.kernel_timer = timer
    probe = HopProbe(torch, device) if (world > 1 and args.mode == "ring") else None
    eng.hop_probe = probe

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    eng.accel(eps, G)                      # initial a(x): not part of a step
    for _ in range(args.warmup):
        eng.step(dt, 1, eps, G)
    sync()
    # sustained shader clock of this rank's GPU over the timed region (one sleeping wave on a private stream, nb_clock_probe_*; its window is
    # ended by a word written on the compute stream behind the last step, so the synchronisation below never waits for it):
    # boxes differ by several per cent in the clock they hold under this kernel; cycles per interaction is the box-independent figure
    clock = None if args.no_clock_probe else nb.ClockProbe(local_rank, 5000.0)
    timer.enabled = True
    if probe is not None:
        probe.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step(dt, 1, eps, G)
    if clock is not None:
        clock.stop(torch.cuda.current_stream().cuda_stream)
    sync()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if probe is not None:
        probe.enabled = False
    clock_mhz, clock_ms = clock.end() if clock is not None else (None, None)
    clock_min = clock_mhz
    if world > 1:
        t = torch.tensor([elapsed, -(clock_mhz or 0.0)], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        clock_min = -float(t[1].item()) if clock_mhz else None

    symmetric = eng.k.last_accel_symmetric() if world == 1 else eng.own_chunk_symmetric
    sym_all = symmetric and (world == 1 or eng.pairwise)       # N ranks: the pair-symmetric schedule makes every hop two-sided
    interactions = float(args.n) * float(args.n) * args.steps
    value = interactions / elapsed
    k_ms, k_inter, k_count = timer.summary()
    peak = PEAK_FP32_VECTOR_TFLOPS if args.dtype == "f32" else PEAK_FP64_VECTOR_TFLOPS
    achieved = k_inter * FLOP_PER_INTERACTION / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0

    if rank == 0:
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
                           if (world > 1 and eng.pairwise) else ""),
                       "masses": "general (one mass perturbed by 1 ulp)" if args.general_mass else "equal"},
            "world_size": world,
            "backend": (args.backend if args.backend == "gloo" else "nccl (RCCL)") if world > 1 else None,
            "same_device": bool(args.same_device) if world > 1 else None,
            # rank 0's ring hops inside the timed region: exchange time vs the kernel it was posted beside, and what of it the
            # compute stream still had to wait for (0 = hidden)
            "ring": probe.summary() if probe is not None else None,
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
                         "schedule": (None if world == 1 else "pair-symmetric: each pair of chunks worked on by one owner, partial sums returned" if eng.pairwise else "one-sided: every rank sums all chunks for its own sinks"),
                         "pair_terms_evaluated_per_launch_approx": (k_inter / 2.0 if sym_all else k_inter),   # PMC count (one v_rsq per term): profiles/r03_*_rocprofv3_summary.txt
                         "executed_flop_per_interaction": EXECUTED_FLOP_SYM if sym_all else EXECUTED_FLOP_ONE_SIDED,
                         "frac_executed_flops": achieved / peak * (EXECUTED_FLOP_SYM if sym_all else EXECUTED_FLOP_ONE_SIDED) / FLOP_PER_INTERACTION,
                         "interactions_per_launch": k_inter, "avg_launch_ms": k_ms, "launches_timed": k_count,
                         # rank 0's accel launches per step, back to back: what ms_per_step would be with a free exchange and free O(N) kernels
                         "kernel_ms_per_step": k_ms * k_count / max(1, args.steps),
                         "traffic": None,
                         "traffic_source": "NOT measured in this run: HBM bytes per launch from the rocprofv3 --pmc passes of this same command committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE: the partial-sum slabs of the pair-symmetric kernel); algorithmic bytes = 28*N"},
        }
        r = out["roofline"]
        if clock_mhz:
            simds = 256 * 4
            r["sustained_clock_mhz"] = clock_mhz
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
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
