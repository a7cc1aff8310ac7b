#!/usr/bin/env python3
"""bench.py - IF Msamples/s (and x real-time) through acquisition + tracking on MI355X.

One "step" = the reference's postProcessing hot path on one synthetic 37.0 s int8 record resident
in HBM: AcquisitionResult.acquire (32 PRNs, 2 x 1 ms coherent blocks, 29 Doppler bins, fine search)
-> preRun -> TrackingResult.track (8 channels x 37 000 ms).  BASELINE.json configs[1] + configs[2].

    python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1 is launched one process per GPU by `python -m torch.distributed.run`; every rank tracks its
own 8 channels on its own copy of the record (weak scaling, BASELINE.json config 5) and searches
32/N of the PRNs, the peaks being all-gathered with RCCL (config 4's exchange).
Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)
REALTIME_MSPS = 38.192


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ms", type=int, default=37000, help="code periods tracked (default: full config)")
    ap.add_argument("--channels", type=int, default=8, help="tracking channels per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--many-channels", type=int, default=2048,
                    help="extra leg at N=1: channels of the many-channel (bandwidth-regime) tracking run, 0 = skip")
    ap.add_argument("--many-ms", type=int, default=500)
    ap.add_argument("--concurrent", type=int, default=3,
                    help="extra leg at N=1: this many independent records processed at once on one GPU (0 = skip)")
    ap.add_argument("--cpu-trk-ms", type=int, default=4000, help="ms of 1-channel oracle tracking timed")
    ap.add_argument("--cpu-acq-prns", type=int, default=16, help="PRNs of oracle acquisition timed")
    return ap.parse_args()


def _cpu_acq(job):
    from oracle import softgnss_oracle as orc
    host, n_prn = job
    t0 = time.perf_counter()
    r = orc.acquire(orc.OracleSettings(acqSatelliteList=list(range(1, n_prn + 1))), host, as_written=True)
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
    """Time the numpy oracle (a port of the reference's algorithm, as written) on a bounded sample of the same
    workload, one process per host core up to 8 (PRNs and channels are independent, numpy's FFT and ufuncs are
    single-threaded), and extrapolate linearly to the full step."""
    from concurrent.futures import ProcessPoolExecutor
    from oracle import softgnss_oracle as orc
    workers = max(1, min(8, os.cpu_count() or 1))
    host = pkg.synth.generate(scene, pkg.synth.record_length(n_code, args.cpu_trk_ms))
    k_prn = max(1, args.cpu_acq_prns // workers)
    with ProcessPoolExecutor(max_workers=workers) as ex:
        t0 = time.perf_counter()
        res = list(ex.map(_cpu_acq, [(host[:11 * n_code], k_prn)] * workers))
        t_acq = time.perf_counter() - t0                       # `workers` x k_prn PRN searches in parallel
        ch = orc.pre_run(orc.OracleSettings(numberOfChannels=1), res[0][1])   # PRN 1 is in the scene
        t0 = time.perf_counter()
        t_each = list(ex.map(_cpu_trk, [(host, ch, args.cpu_trk_ms)] * workers))
        t_trk = time.perf_counter() - t0                       # `workers` channels x cpu_trk_ms in parallel
    full = t_acq * (32.0 / (workers * k_prn)) + t_trk * (n_ch * ms / float(workers * args.cpu_trk_ms))
    one = np.mean([r[0] for r in res]) * (32.0 / k_prn) + np.mean(t_each) * (n_ch * ms / float(args.cpu_trk_ms))
    return {"value": total_samples / full / 1e6, "unit": "Msamples/s", "cores": workers, "kind": "port",
            "sample": "numpy oracle in %d processes: as-written acquisition of %d PRNs each on 11 ms (%.2f s) + one "
                      "channel x %d ms tracking each (%.2f s), scaled linearly to 32 PRNs + %d channels x %d ms"
                      % (workers, k_prn, t_acq, args.cpu_trk_ms, t_trk, n_ch, ms),
            "seconds_extrapolated": full, "single_core_value": total_samples / one / 1e6}


def guarded(label, seconds, fn, *a):
    """Run an optional leg with a deadline: the headline JSON line must come out whatever the extras do."""
    import threading
    box = {}

    def run():
        try:
            box["out"] = fn(*a)
        except Exception as e:   # noqa: BLE001 - reported, never fatal
            box["out"] = {"error": "%s: %r" % (label, e)}

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return {"error": "%s did not finish within %d s" % (label, seconds)}
    return box["out"]


def concurrent_records(pkg, s, scene, rec_len, n_code, local, args):
    """One receiver keeps 80 of the 256 CUs busy (8 channels x 10 cooperating workgroups), so a GPU can serve several
    independent records at once: each thread below owns a context (stream, scratch, record) and runs the same
    step as the headline measurement; the aggregate is reported next to it, never instead of it."""
    import threading
    n = args.concurrent
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
                ready.wait()
                go.wait()
                for _ in range(args.steps):
                    one()
                ctx.sync()
                fin.wait()
                rec.free()
        except Exception as e:   # noqa: BLE001 - reported in the JSON line
            errors.append(repr(e))
            for b in (ready, go, fin):
                b.abort()

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(n)]
    for t in threads:
        t.start()
    try:
        ready.wait()
        t0 = time.perf_counter()
        go.wait()
        fin.wait()
        dt = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        dt = None
    for t in threads:
        t.join()
    if dt is None or errors:
        return {"records": n, "error": "; ".join(errors) or "barrier broken"}
    value = n * float(rec_len) * args.steps / dt / 1e6
    return {"records": n, "steps_each": args.steps, "value": value, "unit": "Msamples/s",
            "x_realtime_aggregate": value / REALTIME_MSPS, "ms_per_step_each": dt / args.steps * 1e3,
            "note": "independent records on one GPU at once (one context, stream and 1.4 GB record per thread); "
                    "every record is still processed at its own latency-bound rate"}


def pmc_traffic(channels, ms):
    """HBM bytes per trk_kernel launch from the committed rocprofv3 PMC passes (profiles/), if they
    were taken on this workload; None otherwise.  Counters cannot be read from inside the process."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_pmc_trk_kernel.json"):
            try:
                with open(os.path.join(pdir, name)) as f:
                    d = json.load(f)
            except (OSError, ValueError):
                continue
            w = d.get("workload", {})
            if w.get("channels") == channels and w.get("ms") == ms:
                best = (d["hbm_bytes_per_launch"], "profiles/" + name)
    return best


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("SGX_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # one process per GPU
    if world != args.gpus and world > 1:
        args.gpus = world
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    pkg = importlib.import_module("softgnss-python_amd")
    shard = importlib.import_module("softgnss-python_amd.shard")
    if pkg._native.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libsgx has no CPU path")

    s = pkg.Settings()
    s.msToProcess = float(args.ms)
    s.numberOfChannels = args.channels
    n_code = s.samplesPerCode
    ctx = pkg.engine.get_context(s, local)

    # ---- peak gather transport ----------------------------------------------------------------
    gather = shard.LocalGather()
    if world > 1:
        try:
            uid = [pkg._native.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            gather = shard.RcclGather(pkg._native.Comm(ctx, world, rank, uid[0]))
            gather.allgather(shard.pack_peaks([], dict(), 1))   # warm the communicator
        except Exception as e:   # flagged, never silent
            sys.stderr.write("[bench] rank %d: RCCL gather unavailable (%s); using host gather\n" % (rank, e))
            gather = shard.HostGather(dist)

    # ---- synthetic record, generated in HBM (bit-identical to softgnss-python_amd/synth.py) --------
    scene = pkg.synth.Scene.default()
    rec_len = pkg.synth.record_length(n_code, args.ms)
    rec = ctx.synth(scene, rec_len)
    signal = pkg.DeviceSignal(rec, 0, 11 * n_code)

    def device_sync():
        ctx.sync()
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except ImportError:
            pass

    def barrier():
        if dist is not None:
            dist.barrier()

    # result buffers live in pinned host memory (the kernel writes its per-millisecond records straight into
    # them); pinning is slow, so the two buffers the steady state alternates between are created during setup
    warm = [pkg._native.pinned_empty((args.channels, 13, args.ms)) for _ in range(2)]
    del warm

    last = {}

    def step():
        acq = pkg.AcquisitionResult(s, device=local)
        shard.acquire_sharded(acq, signal, rank, world, gather)
        t = ctx.timing()
        last["acquire_ms"] = t["acquire_ms"]
        acq.preRun()
        trk = pkg.TrackingResult(acq, device=local)
        trk.track(pkg.DeviceFile(rec))
        if trk.series is None:
            raise RuntimeError("tracking ran out of record")
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
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

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
        traffic = pmc_traffic(args.channels, args.ms)
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
            "us_per_code_period": k_ms * 1e3 / args.ms,
            "roofline": {"kernel": "trk_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic[0] if traffic else None,
                         "traffic_source": (traffic[1] + " (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes of this "
                                            "command; FETCH_SIZE x2, gfx950 correction)") if traffic else None,
                         "algorithmic_bytes_per_launch": b_trk,
                         "measured_stream_read_gbs": read_gbs, "measured_stream_copy_gbs": copy_gbs,
                         "frac_of_measured_read": achieved / read_gbs,
                         "note": "37 000 dependent steps per channel; 8 channels x 10 cooperating CUs = 80 of 256 "
                                 "CUs busy: latency-bound, not bandwidth-bound (DESIGN.md section 5)"},
        }
        if world == 1 and args.many_channels > 0:
            # the throughput-mode kernel, where bandwidth and not the 37 000-step dependency chain is the limit: one
            # CU per channel (two channels per CU), replicas of the acquired channels, HIP-event kernel time
            def many_leg():
                chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in acq.channels if c.PRN != 0]
                many = [chans[i % len(chans)] for i in range(args.many_channels)]
                ctx.track(rec, many, 20)
                ser, dn = ctx.track(rec, many, args.many_ms)
                t_ms = ctx.timing()["track_ms"]
                b_many = float(np.sum(ser[:, 0, -1] - np.array([c[2] for c in many]))) + \
                    len(many) * args.many_ms * 13 * 8.0
                return {"kernel": "trk_kernel_tp", "bound": "hbm", "channels": len(many), "ms": args.many_ms,
                        "achieved": b_many / (t_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": b_many / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": t_ms,
                        "frac_of_measured_read": b_many / (t_ms * 1e-3) / 1e9 / read_gbs,
                        "note": "throughput-mode kernel (one lane per prompt chip, split=1): channels x ms code "
                                "periods of independent work; fp64 VALU-bound at >= 4 instructions per sample"}
            out["roofline_many_channels"] = guarded("roofline_many_channels", 180, many_leg)
        if world == 1 and args.concurrent > 1:
            out["concurrent_records"] = guarded("concurrent_records", 180, concurrent_records, pkg, s, scene, rec_len,
                                                n_code, local, args)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = guarded("cpu_baseline", 600, cpu_baseline, pkg, scene, n_code, args, samples_per_step,
                                          args.channels, args.ms)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
