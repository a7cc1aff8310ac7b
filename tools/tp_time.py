import importlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
m = importlib.import_module("softgnss-python_amd")
s = m.Settings(); ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, 37000))
a = m.AcquisitionResult(s, device=0); a.acquire(m.DeviceSignal(rec, 0, 11 * n)); a.preRun()
chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels if c.PRN != 0]
many = [(chans[i % len(chans)][0], chans[i % len(chans)][1], chans[i % len(chans)][2] + (i // len(chans)) * 140 * n) for i in range(2048)]
ctx.track(rec, many, 20)
ts = []
for _ in range(3):
    ser, dn = ctx.track(rec, many, 500); ts.append(ctx.timing()["track_ms"])
print(os.environ.get("SGX_LIB", "default").split("/")[-1], "2048 x 500 ms:", min(ts), "ms")
