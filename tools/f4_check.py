"""(diagnosis) peakMetric of all 32 PRNs: four-step kernels against the pass-per-radix path (SGX_ACQ_V1=1). GPU box."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("softgnss-python_amd")
s = m.Settings(); ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), 12 * n)
sig = m.DeviceSignal(rec, 0, 11 * n)
a = m.AcquisitionResult(s, device=0); a.acquire(sig)
os.environ["SGX_ACQ_V1"] = "1"
b = m.AcquisitionResult(s, device=0); b.acquire(sig)
pa, pb = a.internals["peakMetricAll"] if "peakMetricAll" in a.internals else a.peakMetric, b.internals["peakMetricAll"] if "peakMetricAll" in b.internals else b.peakMetric
print(os.environ.get("SGX_LIB", "default").split("/")[-1], "detections", int((a.carrFreq > 0).sum()), "ref", int((b.carrFreq > 0).sum()),
      "max rel diff", float(np.max(np.abs(pa - pb) / np.maximum(pb, 1e-30))), "freqBin equal", bool(np.array_equal(a.internals["freqBin"], b.internals["freqBin"])),
      "codePhase equal", bool(np.array_equal(a.codePhase, b.codePhase)))
