"""Per-member phase times of the latency-mode tracking kernel (SGX_TRK_PROFILE=1): GPU box."""
import importlib, os, sys
os.environ["SGX_TRK_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module('softgnss-python_amd')
ms = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
s = m.Settings(); ctx = m.engine.get_context(s, 0)
n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, ms))
a = m.AcquisitionResult(s, device=0); a.acquire(m.DeviceSignal(rec, 0, 11 * n)); a.preRun()
chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels]
ctx.track(rec, chans, ms)
series, done = ctx.track(rec, chans, ms)
print("kernel ms", ctx.timing()["track_ms"], "us/block", ctx.timing()["track_ms"] * 1e3 / ms)
