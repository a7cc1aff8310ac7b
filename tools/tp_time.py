"""(diagnosis) kernel time of the throughput-mode tracking kernel: N channels staggered over the 37-s record x 500 ms.
Usage (GPU box): [SGX_LIB=...] python tools/tp_time.py [channels ...]"""
import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
m = importlib.import_module("softgnss-python_amd")
s = m.Settings(); ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, 37000))
a = m.AcquisitionResult(s, device=0); a.acquire(m.DeviceSignal(rec, 0, 11 * n)); a.preRun()
chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels if c.PRN != 0]
for nch in [int(x) for x in sys.argv[1:]] or [2048]:
    step = max(1, (37000 - 502) // max(1, nch // len(chans)))
    many = [(chans[i % len(chans)][0], chans[i % len(chans)][1], chans[i % len(chans)][2] + (i // len(chans)) * step * n) for i in range(nch)]
    ctx.track(rec, many, 20)
    ts = []
    for _ in range(3):
        ser, dn = ctx.track(rec, many, 500); ts.append(ctx.timing()["track_ms"])
    byts = float(sum(ser[i, 0, -1] - many[i][2] for i in range(nch))) + nch * 500 * 104.0
    print(os.environ.get("SGX_LIB", "default").split("/")[-1], "%d x 500 ms: %.3f ms  %.3f TB/s  %.1f %% of 8 TB/s  locked %d"
          % (nch, min(ts), byts / min(ts) / 1e9, byts / min(ts) / 1e9 / 8.0 * 100.0, int((dn == 500).sum())))
