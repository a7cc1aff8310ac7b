"""N acquisitions of BASELINE configs[3] (32 PRNs, 10 ms non-coherent) and nothing else. GPU box."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("softgnss-python_amd")
n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3
s = m.Settings(); ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), 21 * n)
sig = m.DeviceSignal(rec, 0, 20 * n)
for _ in range(n_calls):
    a = m.AcquisitionResult(s, device=0); a.acquire(sig, n_blocks=10, noncoh=True)
print("acquire_ms", ctx.timing()["acquire_ms"], "detections", int((a.carrFreq > 0).sum()))
