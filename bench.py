#!/usr/bin/env python3
"""bench.py - IF Msamples/s (and x real-time) through acquisition + tracking on MI355X.

One "step" = the reference's postProcessing hot path on one synthetic 37.0 s int8 record resident
in HBM: AcquisitionResult.acquire (32 PRNs, 2 x 1 ms coherent blocks, 29 Doppler bins, fine search)
-> preRun -> TrackingResult.track (8 channels x 37 000 ms).  BASELINE.json configs[1] + configs[2].

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 runs one process per GPU.  Either the driver starts the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the
environment), or - when WORLD_SIZE is not set - this script starts them itself, as children, before it
touches the GPU (`launch_ranks`), and exits non-zero when the node has fewer than N devices.  The ranks meet over
a socket rendezvous of their own (softgnss-python_amd/rendezvous.py: barrier, max over ranks, the RCCL unique id);
PyTorch is not imported.  Every rank
tracks its own 8 channels on its own copy of the record (weak scaling, BASELINE.json config 5) and
searches 32/N of the PRNs, the peaks being all-gathered with RCCL (config 4's exchange).  The
`acq_config4` leg reports BASELINE.json config 4 itself: 32 PRNs x 10 ms non-coherent, PRNs sharded over
the ranks, gather included, next to the same search on one GPU.  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6  # MI355X vector fp64 datasheet peak (256 CUs x 4 SIMDs x 16 FMA lanes x 2 x 2.4 GHz)
REALTIME_MSPS = 38.192
PKG_NAME = os.environ.get("SGX_BENCH_PKG", "softgnss-python_amd")   # (the CPU test of the N > 1 plumbing names a stand-in)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ms", type=int, default=37000, help="code periods tracked (default: full config)")
    ap.add_argument("--channels", type=int, default=8, help="tracking channels per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--many-channels", type=int, default=3072,
                    help="extra leg at N=1: channels of the many-channel (bandwidth-regime) tracking run, 0 = skip")
    ap.add_argument("--many-ms", type=int, default=500)
    ap.add_argument("--concurrent", type=int, default=3,
                    help="extra leg at N=1: this many independent records processed at once on one GPU (0 = skip)")
    ap.add_argument("--cpu-trk-ms", type=int, default=1000, help="ms of oracle tracking timed per channel (BASELINE.md 3)")
    ap.add_argument("--no-config4", action="store_true", help="skip the acq_config4 leg")
    ap.add_argument("--no-from-file", action="store_true", help="skip the from_file leg (the step from a record file)")
    ap.add_argument("--eager", action="store_true",
                    help="the host looks at every stage's result before it queues the next (the reference's call sequence "
                         "as written); default at one rank: the three stages queued, one wait")
    return ap.parse_args()


# ---- self-launch -----------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def count_devices_in_child():
    """Number of HIP devices, asked in a child process: the launcher itself must stay clear of the GPU runtime
    (it goes on to start other programs)."""
    code = ("import importlib,sys; sys.path.insert(0, %r); "
            "print(importlib.import_module(%r)._native.device_count())" % (ROOT, PKG_NAME))
    try:
        r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        return int(r.stdout.decode().strip().splitlines()[-1])
    except Exception:   # noqa: BLE001 - reported as "no devices"
        return 0


def launch_ranks(n, argv):
    """Start `n` ranks of this script (one per GPU), relay rank 0's JSON line, return the exit code."""
    have = count_devices_in_child()
    if have < n:
        sys.stderr.write("[bench] --gpus %d asked for, %d HIP device(s) visible: refusing to report a %d-GPU number\n"
                         % (n, have, n))
        return 3
    port = _free_port()
    token = os.urandom(16).hex()   # the ranks' rendezvous admits only holders of it (softgnss-python_amd/rendezvous.py)
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), SGX_DEVICE=str(r), MASTER_ADDR="127.0.0.1", SGX_RDV_TOKEN=token,
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # Watch ALL ranks: the first one that fails ends the others (they would wait for it at the next barrier), and
    # nothing waits longer than the overall limit.
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("SGX_BENCH_TIMEOUT", "3600"))
    codes = [None] * n
    failed = False
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if any(c not in (None, 0) for c in codes) or time.time() > deadline:
            failed = True
            break
        time.sleep(0.05)
    if failed:
        for r, p in enumerate(procs):
            if codes[r] is None:
                p.terminate()
        for r, p in enumerate(procs):
            if codes[r] is None:
                try:
                    codes[r] = p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    codes[r] = p.wait()
    reader.join(10)
    if not failed:
        sys.stdout.write((out0[0] if out0 else b"").decode(errors="replace"))
        sys.stdout.flush()
        return 0
    sys.stderr.write("[bench] ranks failed or timed out (exit codes %s); the others were stopped\n" % codes)
    return 1


# ---- CPU baseline (numpy oracle; runs before this process touches the GPU) ---------------------------
def _cpu_acq(job):
    from oracle import softgnss_oracle as orc
    host, prns = job
    t0 = time.perf_counter()
    r = orc.acquire(orc.OracleSettings(), host, as_written=True, prn_indices=prns)
    return time.perf_counter() - t0, r


def _cpu_trk(job):
    from oracle import softgnss_oracle as orc
    host, ch, ms = job
    s_trk = orc.OracleSettings(numberOfChannels=1, msToProcess=float(ms))
    t0 = time.perf_counter()
    out = orc.track(s_trk, ch, host)
    assert out is not None
    return time.perf_counter() - t0


def cpu_baseline(pkg, scene, n_code, args, total_samples, n_ch, ms):
    """the numpy oracle (a port of the reference's algorithm, as written) on the host cores of
    this box, one process per core up to 8 (PRNs and channels are independent; numpy's FFT and ufuncs are
    single-threaded).  Config 2 (32 PRNs, as written) runs IN FULL; config 3 is timed on `cpu_trk_ms` ms x n_ch
    channels and scaled linearly to `ms`."""
    from concurrent.futures import ProcessPoolExecutor
    from oracle import softgnss_oracle as orc
    workers = max(1, min(8, os.cpu_count() or 1))
    host = pkg.synth.generate(scene, pkg.synth.record_length(n_code, args.cpu_trk_ms))
    shards = [list(range(32))[w::workers] for w in range(workers)]
    with ProcessPoolExecutor(max_workers=workers) as ex:
        t0 = time.perf_counter()
        res = list(ex.map(_cpu_acq, [(host[:11 * n_code], sh) for sh in shards]))
        t_acq = time.perf_counter() - t0                       # all 32 PRN searches, `workers` at a time
        merged = res[0][1]
        for _, r in res[1:]:
            for k in ("carrFreq", "codePhase", "peakMetric"):
                merged[k] = merged[k] + r[k]                    # disjoint PRN sets: the other entries are zero
        chans = orc.pre_run(orc.OracleSettings(numberOfChannels=n_ch), merged)
        jobs = []
        for i in range(n_ch):
            if chans["PRN"][i] != 0:
                one = dict(PRN=chans["PRN"][i:i + 1], acquiredFreq=chans["acquiredFreq"][i:i + 1],
                           codePhase=chans["codePhase"][i:i + 1], status=chans["status"][i:i + 1])
                jobs.append((host, one, args.cpu_trk_ms))
        t0 = time.perf_counter()
        t_each = list(ex.map(_cpu_trk, jobs))
        t_trk = time.perf_counter() - t0                       # n_ch channels x cpu_trk_ms, `workers` at a time
    full = t_acq + t_trk * (ms / float(args.cpu_trk_ms))
    # ONE worker alone on the box (nothing else running): 4 of the 32 PRN searches (x 8) and one channel's tracking
    # (x the number of channels), both scaled linearly - the other seven cores idle, so no memory-bandwidth contention
    t_acq1, _ = _cpu_acq((host[:11 * n_code], list(range(32))[0::8]))
    t_trk1 = _cpu_trk(jobs[0]) if jobs else 0.0
    one_core = t_acq1 * 8.0 + t_trk1 * len(jobs) * (ms / float(args.cpu_trk_ms))
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": total_samples / full / 1e6, "unit": "Msamples/s", "cores": workers, "kind": "port",
            "sample": "numpy oracle in %d processes: config 2 in full (as-written acquisition of all 32 PRNs on 11 ms, "
                      "%.2f s) + %d channels x %d ms of tracking (%.2f s) scaled linearly to %d ms"
                      % (workers, t_acq, len(jobs), args.cpu_trk_ms, t_trk, ms),
            "seconds_extrapolated": full, "acq_config2_seconds": t_acq,
            "host_cpu_count": os.cpu_count(), "host_cpu_model": model,
            "single_core_value": total_samples / one_core / 1e6, "single_core_seconds_extrapolated": one_core,
            "single_core_acq_config2_seconds": t_acq1 * 8.0,
            "note": "`value` = %d worker processes at once (`cores`); `single_core_value` = ONE process alone on the idle "
                    "box: 4 PRN searches x 8 (%.2f s measured) + 1 channel x %d ms x %d channels (%.2f s measured), scaled "
                    "linearly to the full workload" % (workers, t_acq1, args.cpu_trk_ms, len(jobs), t_trk1)}


# ---- optional GPU legs -------------------------------------------------------------------------------
def leg(label, fn, *a):
    """Run an optional leg on the main thread; an exception is reported in the JSON line, never fatal."""
    try:
        return fn(*a)
    except Exception as e:   # noqa: BLE001
        return {"error": "%s: %r" % (label, e)}


def concurrent_records(pkg, s, scene, rec_len, n_code, local, args):
    """With one workgroup per (channel, unit) - the 10-member layout, SGX_TRK_ARMS=3 - a receiver keeps 80 of the 256 CUs
    busy, so a GPU can serve three independent records at once (the headline's 30-member layout takes 240 CUs for one
    record: lowest latency, no room for a second).  Each thread below owns a context (stream, scratch, record) and runs
    the same step as the headline measurement in that layout; the aggregate is reported next to the headline, never
    instead of it."""
    import threading
    n = args.concurrent
    os.environ["SGX_TRK_ARMS"] = "3"
    ready = threading.Barrier(n + 1)
    go = threading.Barrier(n + 1)
    fin = threading.Barrier(n + 1)
    errors = []

    def worker(k):
        try:
            # three stream priority classes = three disjoint sets of hardware queues: the persistent kernels of
            # different records then never queue up behind each other
            with pkg.engine.private_context(s, local, priority=(-1, 1, 0)[k % 3]) as ctx:
                rec = ctx.synth(scene, rec_len)
                signal = pkg.DeviceSignal(rec, 0, 11 * n_code)

                def one():
                    acq = pkg.AcquisitionResult(s, device=local)
                    acq.acquire(signal)
                    acq.preRun()
                    trk = pkg.TrackingResult(acq, device=local)
                    trk.track(pkg.DeviceFile(rec))
                    if trk.series is None:
                        raise RuntimeError("tracking ran out of record")

                one()
                ready.wait(600)
                go.wait(600)
                for _ in range(args.steps):
                    one()
                ctx.sync()
                fin.wait(600)
                rec.free()
        except Exception as e:   # noqa: BLE001 - reported in the JSON line
            errors.append(repr(e))
            for b in (ready, go, fin):
                b.abort()

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(n)]
    for t in threads:
        t.start()
    try:
        ready.wait(600)
        t0 = time.perf_counter()
        go.wait(600)
        fin.wait(600)
        dt = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        dt = None
    for t in threads:
        t.join()
    os.environ.pop("SGX_TRK_ARMS", None)
    if dt is None or errors:
        return {"records": n, "error": "; ".join(errors) or "barrier broken"}
    value = n * float(rec_len) * args.steps / dt / 1e6
    return {"records": n, "steps_each": args.steps, "value": value, "unit": "Msamples/s",
            "x_realtime_aggregate": value / REALTIME_MSPS, "ms_per_step_each": dt / args.steps * 1e3,
            "note": "independent records on one GPU at once (one context, stream and 1.4 GB record per thread), each in the "
                    "10-members-per-channel layout (80 CUs); every record is still processed at its own latency-bound rate"}


def pmc_file(suffix, match):
    """A committed rocprofv3 PMC summary under profiles/ (counters cannot be read from inside the process):
    the newest file `*<suffix>` whose "workload" dict contains `match`."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith(suffix):
            try:
                with open(os.path.join(pdir, name)) as f:
                    d = json.load(f)
            except (OSError, ValueError):
                continue
            w = d.get("workload", {})
            if all(w.get(k) == v for k, v in match.items()):
                best = (d, "profiles/" + name)
    return best


def trace_avg_ms(kernel_prefix):
    """Average duration (ms) of a kernel in the newest committed `profiles/*_kernel_trace_stats.csv` (rocprofv3
    --kernel-trace --stats of this command at its headline workload): the figure the live HIP-event time must agree with."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_kernel_trace_stats.csv"):
            try:
                with open(os.path.join(pdir, name)) as f:
                    for line in f:
                        if line.startswith('"') and kernel_prefix in line.split('"')[1]:
                            cols = line.rsplit('"', 1)[1].split(",")
                            best = (float(cols[3]) / 1e3, "profiles/" + name)
                            break
            except (OSError, ValueError, IndexError):
                continue
    return best


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("SGX_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # one process per GPU
    if world != args.gpus:
        if world > 1:
            args.gpus = world
        else:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is 1" % args.gpus)
    pkg = importlib.import_module(PKG_NAME)
    shard = importlib.import_module("softgnss-python_amd.shard")
    rendezvous = importlib.import_module("softgnss-python_amd.rendezvous")

    s = pkg.Settings()
    s.msToProcess = float(args.ms)
    s.numberOfChannels = args.channels
    n_code = s.samplesPerCode
    scene = pkg.synth.Scene.default()
    rec_len = pkg.synth.record_length(n_code, args.ms)

    # The CPU baseline forks worker processes: it runs first, while this process has not loaded the GPU runtime.
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = leg("cpu_baseline", cpu_baseline, pkg, scene, n_code, args, float(rec_len), args.channels, args.ms)

    group = rendezvous.HostGroup(rank, world)
    if pkg._native.device_count() < max(1, local + 1):
        raise SystemExit("bench.py needs an MI355X per rank (rank %d wants device %d): libsgx has no CPU path"
                         % (rank, local))
    ctx = pkg.engine.get_context(s, local)

    # ---- peak gather transport ----------------------------------------------------------------
    gather = shard.LocalGather()
    if world > 1:
        try:
            uid = group.broadcast(pkg._native.Comm.unique_id() if rank == 0 else None)
            gather = shard.RcclGather(pkg._native.Comm(ctx, world, rank, uid))
            gather.allgather(shard.pack_peaks([], dict(), 1))   # warm the communicator
        except Exception as e:   # flagged, never silent
            sys.stderr.write("[bench] rank %d: RCCL gather unavailable (%s); using host gather\n" % (rank, e))
            gather = shard.HostGather(group)
        names = group.gather(gather.name)
        if len(set(names)) != 1:   # a mixed transport would deadlock: everybody falls back, and says so
            gather = shard.HostGather(group)

    # ---- synthetic record, generated in HBM (bit-identical to softgnss-python_amd/synth.py) --------
    rec = ctx.synth(scene, rec_len)
    signal = pkg.DeviceSignal(rec, 0, 11 * n_code)

    def device_sync():
        ctx.sync()   # everything this rank queued runs on its context's stream

    barrier = group.barrier
    max_over_ranks = group.max

    # result buffers live in pinned host memory (the kernel writes its per-millisecond records straight into
    # them); pinning is slow, so the two buffers the steady state alternates between are created during setup
    warm = [pkg._native.pinned_empty((args.channels, 13, args.ms)) for _ in range(2)]
    del warm

    last = {}

    def step():
        # One rank: the three stages are QUEUED (AcquisitionResult(deferred=True): the search, preRun on the device, the
        # tracking kernel) and the host waits once; several ranks: the search is sharded and its peaks gathered first.
        # Either way every stage's work is done inside the step, and the results are those of the eager calls bit for
        # bit (tests/test_gpu_parity.py: test_deferred_step_equals_the_eager_one).
        acq = pkg.AcquisitionResult(s, device=local, deferred=(world == 1 and not args.eager))
        shard.acquire_sharded(acq, signal, rank, world, gather)
        acq.preRun()
        trk = pkg.TrackingResult(acq, device=local)
        trk.track(pkg.DeviceFile(rec))
        if trk.series is None:
            raise RuntimeError("tracking ran out of record")
        acq.results                      # (the look at the search's page: no waiting left after a chained run)
        last["acquire_ms"] = ctx.timing()["acquire_ms"]
        last["chained"] = bool(trk.chained)
        last["track_ms"] = trk.kernel_ms
        last["series"] = trk.series
        last["acq"] = acq
        return trk

    for _ in range(args.warmup):
        step()
    trk_kernel_ms = []
    acq_ms = []
    barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        trk_kernel_ms.append(last["track_ms"])
        acq_ms.append(last["acquire_ms"])
    device_sync()
    barrier()
    elapsed_own = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed_own)
    # what every rank measured on its own GPU (rank 0 reports the extremes next to the max-over-ranks figure)
    per_rank = group.gather({"rank": rank, "device": local, "step_ms": elapsed_own / args.steps * 1e3,
                             "track_kernel_ms": float(np.mean(trk_kernel_ms)), "acquire_ms": float(np.mean(acq_ms)),
                             "peak_gather": gather.name})

    # ---- BASELINE.json config 4: 32 PRNs x 10 ms non-coherent, PRNs sharded, RCCL gather included ----------
    cfg4 = None
    if not args.no_config4:
        sig4 = pkg.DeviceSignal(rec, 0, 20 * n_code)
        reps = 5

        def acq4(r, w, g):
            a = pkg.AcquisitionResult(s, device=local)
            shard.acquire_sharded(a, sig4, r, w, g, n_blocks=10, noncoh=True)
            return a

        acq4(rank, world, gather)
        barrier()
        device_sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            a4 = acq4(rank, world, gather)
        device_sync()
        barrier()
        t_sharded = max_over_ranks(time.perf_counter() - t0) / reps * 1e3
        dev_sharded = max_over_ranks(ctx.timing()["acquire_ms"])
        if rank == 0:
            # the same search on ONE GPU, timed in the same process (the other ranks wait at the barrier below)
            one = shard.LocalGather()
            acq4(0, 1, one)
            device_sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                a1 = acq4(0, 1, one)
            device_sync()
            t_one = (time.perf_counter() - t0) / reps * 1e3
            dev_one = ctx.timing()["acquire_ms"]
            same = bool(np.array_equal(a1.codePhase, a4.codePhase) and np.array_equal(a1.carrFreq, a4.carrFreq) and
                        np.array_equal(a1.internals["freqBin"], a4.internals["freqBin"]))
            emu = None
            if world == 1:
                # what the slowest rank of an 8-GPU run executes, timed on THIS GPU: each 4-PRN shard alone (device
                # time of the whole sgx_acquire call, best of three), the slowest of the eight; no gather in it
                worst = 0.0
                for rk in range(8):
                    prns = list(shard.plan_shards(32, 8)[rk])
                    ae = pkg.AcquisitionResult(s, device=local)
                    ae.acquire(sig4, n_blocks=10, noncoh=True, prn_indices=prns)
                    ts = []
                    for _ in range(3):
                        ae.acquire(sig4, n_blocks=10, noncoh=True, prn_indices=prns)
                        ts.append(ctx.timing()["acquire_ms"])
                    worst = max(worst, min(ts))
                # ... and the same shards END TO END through shard.acquire_sharded as a rank calls it (sgx_acquire_sharded
                # without a communicator: search, device-side pack, publish, one look, merge), wall clock, mean of 5 calls
                worst_wall = 0.0
                lg = shard.LocalGather()
                for rk in range(8):
                    acq4(rk, 8, lg)
                    device_sync()
                    t0 = time.perf_counter()
                    for _ in range(5):
                        acq4(rk, 8, lg)
                    device_sync()
                    worst_wall = max(worst_wall, (time.perf_counter() - t0) / 5 * 1e3)
                emu = {"emulated_8rank_ms": worst, "emulated_speedup": dev_one / worst,
                       "emulated_wall_ms": worst_wall, "emulated_wall_speedup": t_one / (worst_wall + 0.03),
                       "emulated_wall_note": "wall clock of the slowest 4-PRN shard through shard.acquire_sharded (one library "
                                             "call: search, pack on the device, one look, merge) against ms_n1, the wall clock "
                                             "of the whole search on this GPU; + 0.03 ms for the ncclAllGather of 160 bytes "
                                             "per rank, which no single GPU can run",
                       # what a rank spends beyond an eighth of the one-GPU search: the part that does not shard
                       "emulated_remainder_us": (worst - dev_one / 8.0) * 1e3,
                       "emulated_note": "device time of the slowest 4-PRN shard of an 8-rank run, timed on this one GPU, against "
                                        "device_ms_n1; the ncclAllGather of 160 bytes per rank is not in it"}
            cfg4 = {"workload": "configs[3]: 32 PRNs x 10 ms non-coherent, %d PRN/GPU, peaks all-gathered (%s)"
                                % (len(shard.plan_shards(32, world)[0]), gather.name),
                    "ms": t_sharded, "ms_n1": t_one, "speedup_vs_n1": t_one / t_sharded,
                    "device_ms_slowest_rank": dev_sharded, "device_ms_n1": dev_one,
                    "sharded_result_equals_single_gpu": same,
                    "note": "wall time per search incl. host glue and the gather, max over ranks; ms_n1 = the whole "
                            "search on rank 0's GPU in the same run"}
            if emu:
                cfg4.update(emu)
        barrier()

    # ---- accounting ----------------------------------------------------------------------------
    series = last["series"]
    n_act = series.shape[0]
    acq = last["acq"]
    start_pos = np.array([acq.channels.codePhase[i] for i in range(n_act)])
    streamed = float(np.sum(series[:, 0, -1] - start_pos))              # sum over channels of sum blksize
    b_trk = streamed + n_act * args.ms * 13 * 8.0                       # SURVEY.md section 8(d) B_trk
    k_ms = float(np.mean(trk_kernel_ms))
    achieved = b_trk / (k_ms * 1e-3) / 1e9
    samples_per_step = float(rec_len)                                   # IF samples of the record one rank consumes
    value = world * samples_per_step * args.steps / elapsed / 1e6

    if rank == 0:
        traffic = pmc_file("_pmc_trk_kernel.json", {"channels": args.channels, "ms": args.ms})
        chain = pmc_file("_trk_chain.json", {"channels": args.channels, "ms": args.ms})
        tk = ctx.timing()
        trk_name = {2: "trk2_kernel", 3: "trk_kernel_tp", 4: "trk_kernel_multi", 5: "trk3_kernel"}.get(tk.get("track_kernel"), "trk2_kernel")
        prof_avg = trace_avg_ms(trk_name) if (args.channels, args.ms) == (8, 37000) else None
        read_gbs, copy_gbs = ctx.stream_rates(1 << 30, 5)
        out = {
            "metric": "IF Msamples/s through acquisition + tracking (x real-time = value / 38.192)",
            "value": value, "unit": "Msamples/s", "x_realtime": value / REALTIME_MSPS,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]+configs[2]: 32-PRN acquisition (2x1 ms coherent, 29 bins, fine search) "
                                   "+ %d-channel DLL/PLL tracking x %d ms on one %.3f GB int8 record @38.192 Msps per GPU"
                                   % (args.channels, args.ms, rec_len / 1e9),
                       "channels_per_gpu": args.channels, "channels_active_per_gpu": int(n_act), "ms": args.ms,
                       "prns_per_gpu": len(shard.plan_shards(32, world)[0]), "record_samples": rec_len,
                       "peak_gather": gather.name},
            "acquire_ms": float(np.mean(acq_ms)), "track_kernel_ms": k_ms,
            "host_glue_ms": elapsed / args.steps * 1e3 - k_ms - float(np.mean(acq_ms)),
            "step_mode": ("queued: deferred acquisition, preRun on the device, tracking kernel behind it, one wait "
                          "(sgx_acquire_begin / sgx_track_chained)") if last.get("chained") else
                         "eager: the host looks at the search, runs preRun, then queues the tracking kernel",
            "us_per_code_period": k_ms * 1e3 / args.ms,
            "per_rank": {k: {"min": min(r[k] for r in per_rank), "max": max(r[k] for r in per_rank)}
                         for k in ("step_ms", "track_kernel_ms", "acquire_ms")},
            "per_rank_peak_gather": sorted(set(r["peak_gather"] for r in per_rank)),
            "roofline": {"kernel": trk_name, "bound": "latency", "bound_note": "frac is still the HBM fraction north_star asks "
                         "for; the kernel is bound by the latency of its per-block chain (limited_by)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic[0]["hbm_bytes_per_launch"] if traffic else None,
                         "traffic_source": (traffic[1] + " (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes of this "
                                            "command; FETCH_SIZE x2, gfx950 correction)") if traffic else None,
                         "algorithmic_bytes_per_launch": b_trk,
                         "kernel_avg_ms": k_ms, "kernel_avg_ms_profile": prof_avg[0] if prof_avg else None,
                         "kernel_avg_ms_profile_source": prof_avg[1] if prof_avg else None,
                         "measured_stream_read_gbs": read_gbs, "measured_stream_copy_gbs": copy_gbs,
                         "frac_of_measured_read": achieved / read_gbs,
                         "workgroups_per_channel": tk.get("track_members"),
                         # WHY the fraction is what it is: the per-block dependency chain in shader cycles (SGX_TRK_PROFILE
                         # phase times of the committed run) and how busy the occupied CUs' vector ALUs are (PMC passes)
                         "limited_by": "latency of the per-block chain, not bandwidth",
                         "chain_cycles": ({k: chain[0][k] for k in ("final_pass", "exchange", "loop_filter", "block") if k in chain[0]}
                                          if chain else None),
                         "chain_cycles_source": chain[1] if chain else None,
                         "valu_busy_frac_on_the_occupied_cus": traffic[0].get("valu_busy_frac_on_the_occupied_cus") if traffic else None,
                         "occupied_cus": traffic[0].get("occupied_cus") if traffic else None,
                         "valu_insts_per_sample": traffic[0].get("valu_insts_per_sample") if traffic else None,
                         "note": "37 000 dependent steps per channel; %d channels x %d workgroups on as many of the 256 CUs, "
                                 "each on a latency-bound chain (final pass -> exchange -> loop filter): not bandwidth-bound "
                                 "(DESIGN.md section 4.1)" % (args.channels, tk.get("track_members", 0))},
        }
        if cfg4 is not None:
            out["acq_config4"] = cfg4
        out["roofline_acq"] = leg("roofline_acq", acq_roofline, pkg, ctx, s, signal, local, n_code)
        if world == 1 and args.many_channels > 0:
            out["roofline_many_channels"] = leg("roofline_many_channels", many_channels_leg, pkg, ctx, rec, acq, args,
                                                read_gbs, n_code)
        if world == 1 and args.many_channels > 0:
            out["many_channels_other_types"] = leg("many_channels_other_types", many_typed_leg, pkg, ctx, rec, acq, n_code,
                                                   1024, min(300, max(20, args.ms - 20)))
        if world == 1 and args.many_channels > 0:
            out["float_records"] = leg("float_records", float_records_leg, pkg, ctx, rec, acq, n_code,
                                       min(2000, max(20, args.ms - 20)))
        if world == 1 and not args.no_from_file:
            out["from_file"] = leg("from_file", from_file_leg, pkg, ctx, s, rec, rec_len, n_code, local, args, series,
                                   elapsed / args.steps * 1e3)
        if world == 1 and args.concurrent > 1:
            out["concurrent_records"] = leg("concurrent_records", concurrent_records, pkg, s, scene, rec_len, n_code,
                                            local, args)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
        sys.stdout.flush()
    group.barrier()
    group.close()
    ctx.sync()


def acq_roofline(pkg, ctx, s, signal, local, n_code):
    """Roofline of the acquisition (configs[1]): SURVEY.md section 8(d) names FFT arithmetic and on-chip bandwidth,
    not HBM, as its roof.  Algorithmic flops = the deduplicated transform count x 5 N log2 N + pointwise + fine
    search; algorithmic bytes = the 2 ms searched + 10 ms per detection + the result rows."""
    a = pkg.AcquisitionResult(s, device=local)
    a.acquire(signal)
    ts = []
    for _ in range(5):
        a.acquire(signal)
        ts.append(ctx.timing())
    t_ms = float(np.mean([t["acquire_ms"] for t in ts]))
    n_det = int(np.sum(a.carrFreq > 0))
    bins = int(round(s.acqSearchBand * 2)) + 1
    fft_n = 5.0 * n_code * np.log2(n_code)
    flops = (2 * bins + 32 + 32 * 2 * bins) * fft_n + 32 * 2 * bins * n_code * 10.0 + n_det * 5.0 * 2 ** 22 * 22
    alg_bytes = 2.0 * n_code + n_det * 10.0 * n_code + 32 * 24
    pm = pmc_file("_pmc_acq.json", {"prns": 32, "blocks": 2})
    out = {"kernel": "acquisition (all kernels of one sgx_acquire call)", "bound": "fp64 valu", "acquire_ms": t_ms,
           # (the event between the coarse and the fine kernels holds the fine search back by 6-8 us and is recorded with
           # SGX_ACQ_SPLIT_EVENT=1 only: without it the split is not measured)
           "coarse_ms": float(np.mean([t["acq_coarse_ms"] for t in ts])) if ts[0]["acq_fine_ms"] > 0 else None,
           "fine_ms": float(np.mean([t["acq_fine_ms"] for t in ts])) if ts[0]["acq_fine_ms"] > 0 else None,
           "detections": n_det,
           "achieved": flops / (t_ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": flops / (t_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "algorithmic_flops_per_call": flops,
           "algorithmic_bytes_per_call": alg_bytes,
           "traffic": pm[0]["hbm_bytes_per_call"] if pm else None, "traffic_source": pm[1] if pm else None,
           "hbm_gbs_at_counter_traffic": (pm[0]["hbm_bytes_per_call"] / (t_ms * 1e-3) / 1e9) if pm else None}
    if pm:
        # the nearer roof since round 2: the correlation's intermediate crosses HBM once each way and dominates the call
        out["hbm_frac_at_counter_traffic"] = out["hbm_gbs_at_counter_traffic"] / HBM_PEAK_GBS
        out["nearer_roof"] = "hbm" if out["hbm_frac_at_counter_traffic"] > out["frac"] else "fp64 valu"
    return out


def many_channels_leg(pkg, ctx, rec, acq, args, read_gbs, n_code):
    """The throughput-mode kernel, where the 37 000-step dependency chain is not the limit: one workgroup per channel,
    two per CU.  The channels are replicas of the acquired ones; EVERY channel starts at its own offset into the record
    (whole code periods apart, so every replica stays locked; 3 072 channels: 11.9 ms from one to the next), so no two
    workgroups ever read the same bytes at the same time and the launch's working set is the whole 1.4 GB record.
    Timed as mean and minimum over five launches; the layout of earlier rounds (the eight channels i % 8 = 0..7 sharing
    an offset, 384 offsets) is timed next to it."""
    chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in acq.channels if c.PRN != 0]
    span_ms = args.ms - args.many_ms - 2
    nmany = args.many_channels

    def layout(distinct):
        out = []
        for i in range(nmany):
            prn, f, cp = chans[i % len(chans)]
            if span_ms <= 0:
                shift_ms = 0
            elif distinct:
                shift_ms = (i * span_ms) // nmany
            else:
                shift_ms = (i // len(chans)) * max(1, span_ms // max(1, nmany // len(chans)))
            out.append((prn, f, cp + shift_ms * n_code))
        return out

    def timed(many, reps):
        ctx.track(rec, many, 20)
        # two untimed launches at full length whose results are alive at the same time: the timed launches then alternate
        # between two pinned result buffers that have both been written before (the kernel writes its series straight
        # into pinned host memory; the first pass over a freshly pinned 160 MB costs ~2 ms of address translation)
        w1 = ctx.track(rec, many, args.many_ms)
        w2 = ctx.track(rec, many, args.many_ms)
        del w1, w2
        ts = []
        for _ in range(reps):                                # (kernel time by HIP events)
            ser, dn = ctx.track(rec, many, args.many_ms)
            ts.append(ctx.timing()["track_ms"])
        return ser, dn, ts

    many = layout(True)
    ser, dn, ts = timed(many, 5)
    _, dn_g, ts_g = timed(layout(False), 3)
    t_ms = float(np.mean(ts))
    first = np.array([c[2] for c in many])
    b_many = float(np.sum(ser[:, 0, -1] - first)) + len(many) * args.many_ms * 13 * 8.0
    lo, hi = float(np.min(first)), float(np.max(ser[:, 0, -1]))
    pm = pmc_file("_pmc_trk_tp.json", {"channels": len(many), "ms": args.many_ms})
    out = {"kernel": "trk_kernel_tp", "channels": len(many), "ms": args.many_ms,
           "achieved": b_many / (t_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": b_many / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": t_ms,
           "kernel_ms_min": float(np.min(ts)), "kernel_ms_all": [float(t) for t in ts],
           "frac_best_launch": b_many / (float(np.min(ts)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "distinct_start_offsets": len(set(c[2] for c in many)),
           "kernel_ms_grouped_offsets": float(np.mean(ts_g)), "locked_channels_grouped_offsets": int(np.sum(dn_g == args.many_ms)),
           "frac_of_measured_read": b_many / (t_ms * 1e-3) / 1e9 / read_gbs,
           "working_set_bytes": hi - lo, "locked_channels": int(np.sum(dn == args.many_ms)),
           "note": "throughput-mode kernel (one lane per prompt chip, split=1): channels x ms code periods of independent "
                   "work, every channel on its own window of the record; `frac` is the MEAN of five launches. `traffic` "
                   "is FETCH_SIZE x 2 + WRITE_SIZE: those counters sit between L2 and the fabric, so they count bytes that "
                   "the Infinity Cache serves as well as bytes from DRAM; the kernel is bound by vector-instruction issue "
                   "(`bound`), which is why the time is the same whether eight channels share a window "
                   "(kernel_ms_grouped_offsets) or none do"}
    if pm:
        d = pm[0]
        out["traffic"] = d.get("hbm_bytes_per_launch")
        out["valu_busy_frac"] = d.get("valu_busy_frac")
        out["valu_insts_per_sample"] = d.get("valu_insts_per_sample")
        out["traffic_source"] = pm[1]
        out["bound"] = d.get("bound", "valu")
    else:
        out["bound"] = "valu"
    return out


def many_typed_leg(pkg, ctx, rec, acq, n_code, nch=1024, ms=300):
    """The throughput-mode kernel on the other sample types it reads (Settings.dataType 'uint8' and 'int16',
    tracking.py:154): 1 024 channels x 300 ms of the first 310 ms of the record re-typed on the host, channel-seconds
    tracked per second of kernel time (best of three launches).  A side figure; never part of `value`."""
    chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in acq.channels if c.PRN != 0]
    x8 = rec.download(0, (ms + 10) * n_code)
    out = {"channels": nch, "ms": ms, "unit": "channel-s/s"}
    for name, arr, code in (("uint8", (x8.astype(np.int16) + 128).astype(np.uint8), pkg._native.DT_UINT8),
                            ("int16", x8.astype("<i2") * 129 - 5, pkg._native.DT_INT16)):
        isz = arr.dtype.itemsize
        dev = ctx.upload_bytes(np.ascontiguousarray(arr).view(np.int8))
        many = [(chans[j % len(chans)][0], chans[j % len(chans)][1],
                 (chans[j % len(chans)][2] + ((j // len(chans)) % 8) * n_code) * isz) for j in range(nch)]
        ts = []
        for _ in range(3):
            ser, dn = ctx.track(dev, many, ms, data_type=code)
            ts.append(ctx.timing()["track_ms"])
        dev.free()
        out[name] = {"kernel_ms": float(np.min(ts)), "channel_s_per_s": nch * ms / float(np.min(ts)),
                     "locked_channels": int(np.sum(dn == ms)), "track_kernel": int(ctx.timing()["track_kernel"])}
    return out


def float_records_leg(pkg, ctx, rec, acq, n_code, ms=2000):
    """Settings.dataType 'float32' / 'float64' records of ARBITRARY values (tracking.py:154 reads whatever numpy dtype the
    settings name): the acquired channels x `ms` code periods of the record's start re-typed on the host and scaled by
    a non-power of two, on the latency-mode kernel (sgx_trk2.hip <4,3> / <8,3>).  A side figure; never part of `value`."""
    chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in acq.channels if c.PRN != 0]
    x8 = rec.download(0, (ms + 4) * n_code)
    out = {"channels": len(chans), "ms": ms}
    for name, arr, code in (("float32", (x8.astype(np.float64) * 0.37 + 0.011).astype("<f4"), pkg._native.DT_FLOAT32),
                            ("float64", x8.astype(np.float64) * 1.2345e-3, pkg._native.DT_FLOAT64)):
        isz = arr.dtype.itemsize
        dev = ctx.upload_bytes(np.ascontiguousarray(arr).view(np.int8))
        many = [(p, f, cp * isz) for p, f, cp in chans]
        ts = []
        for _ in range(3):
            ser, dn = ctx.track(dev, many, ms, data_type=code)
            ts.append(ctx.timing()["track_ms"])
        dev.free()
        t = float(np.min(ts))
        out[name] = {"kernel_ms": t, "us_per_code_period": t * 1e3 / ms, "x_realtime": ms / t,
                     "locked_channels": int(np.sum(dn == ms)), "track_kernel": int(ctx.timing()["track_kernel"]),
                     "workgroups_per_channel": int(ctx.timing()["track_members"])}
    return out


def from_file_leg(pkg, ctx, s, rec, rec_len, n_code, local, args, resident_series, resident_step_ms):
    """The reference's own caller (initialize.py:466-506): open the record FILE, read 11 ms, acquire, preRun, and hand
    the open file to track().  The resident record is written to a file once (outside every timed region); each timed
    step then opens it, reads the acquisition window with np.fromfile and calls TrackingResult.track(fid), which streams
    the file into HBM (sgx_if_open_file) while the tracking kernel follows the watermark.  Never part of `value`."""
    import tempfile
    path = os.path.join(tempfile.gettempdir(), "sgx_bench_record_%d.bin" % os.getpid())
    rec.download().tofile(path)
    s_file = pkg.Settings()
    s_file.msToProcess = s.msToProcess
    s_file.numberOfChannels = s.numberOfChannels
    s_file.fileName = path
    info = {}

    def one():
        with open(path, "rb") as fid:
            fid.seek(int(s_file.skipNumberOfBytes), 0)                    # initialize.py:472
            data = np.fromfile(fid, s_file.dataType, 11 * n_code)         # initialize.py:481
            acq = pkg.AcquisitionResult(s_file, device=local)
            acq.acquire(data)
            info["acquire_ms"] = ctx.timing()["acquire_ms"]
            acq.preRun()
            trk = pkg.TrackingResult(acq, device=local)
            trk.track(fid)
            if trk.series is None:
                raise RuntimeError("tracking ran out of record")
        t = ctx.timing()
        info["track_kernel_ms"] = t["track_ms"]
        info["streamed"] = bool(t["track_streamed"])
        return trk

    try:
        one()
        ctx.sync()
        k_ms, streamed = [], []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            trk = one()
            k_ms.append(info["track_kernel_ms"])
            streamed.append(info["streamed"])
        ctx.sync()
        dt = (time.perf_counter() - t0) / args.steps
    finally:
        try:
            os.remove(path)
        except OSError:
            pass
    value = float(rec_len) / dt / 1e6
    return {"workload": "the same step from a %.3f GB record FILE (page cache): open, np.fromfile of 11 ms, acquire, preRun, "
                        "TrackingResult.track(fid) - file -> pinned ring -> HBM while the kernel runs" % (rec_len / 1e9),
            "steps": args.steps, "ms_per_step": dt * 1e3, "value": value, "unit": "Msamples/s",
            "x_realtime": value / REALTIME_MSPS, "track_kernel_ms": float(np.mean(k_ms)),
            "acquire_ms": info["acquire_ms"], "streamed": bool(all(streamed)),
            "bit_identical_to_resident": bool(np.array_equal(trk.series, resident_series)),
            "vs_resident_step": dt * 1e3 / resident_step_ms,
            "note": "PCIe-inclusive: 1.4 GB cross the host link inside every step; never part of `value`"}


if __name__ == "__main__":
    main()
