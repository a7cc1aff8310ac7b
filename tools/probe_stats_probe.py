"""Time sgx_probe_stats (Welch PSD + histogram of the first 10 code periods) against the scipy-based oracle.
Usage: python tools/probe_stats_probe.py   (GPU box)"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
m = importlib.import_module("softgnss-python_amd")
from oracle import softgnss_oracle as orc   # noqa: E402  (checker only)

s = m.Settings()
ctx = m.engine.get_context(s, 0)
n = 10 * s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), n)
data = rec.download()
ctx.probe_stats(rec, 0, n, 38.192)
t0 = time.perf_counter()
for _ in range(10):
    f, pxx, hist, nseg = ctx.probe_stats(rec, 0, n, 38.192)
dt = (time.perf_counter() - t0) / 10
t0 = time.perf_counter()
fo, po, ho = orc.probe_stats(orc.OracleSettings(), data)
dto = time.perf_counter() - t0
print("sgx_probe_stats: %.3f ms per call, oracle %.1f ms; max rel PSD error %.2e, hist equal %s"
      % (dt * 1e3, dto * 1e3,
         float(np.max(np.abs(pxx - po) / po)), np.array_equal(hist, ho)))
