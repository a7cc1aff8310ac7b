import importlib, sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden, pkg, scene_from_json
m = pkg()
print("devices", m._native.device_count(), m._native.lib().sgx_version())
g = load_golden("trk_default.npz")
t0 = time.time(); rec_h = m.synth.generate(scene_from_json(g["scene"]), int(g["n_samples"])); print("host gen s", time.time() - t0)
s = m.Settings(); ctx = m.engine.get_context(s, 0)
t0 = time.time(); rec = ctx.synth(scene_from_json(g["scene"]), int(g["n_samples"])); print("dev gen s", time.time() - t0, ctx.timing())
print("gen equal", np.array_equal(rec.download(), rec_h))
a = m.AcquisitionResult(s, device=0)
t0 = time.time(); a.acquire(m.DeviceSignal(rec, 0, 11 * 38192)); print("acq wall s", time.time() - t0, ctx.timing())
t0 = time.time(); a.acquire(m.DeviceSignal(rec, 0, 11 * 38192)); print("acq wall s (2nd)", time.time() - t0, ctx.timing())
ga = load_golden("acq_default.npz")
print("codePhase eq", np.array_equal(a.codePhase, ga["codePhase"]), "carr eq", np.array_equal(a.carrFreq, ga["carrFreq"]))
print("metric rel err", np.max(np.abs(a.peakMetric / ga["peakMetric"] - 1)))
chans = [(int(g["ch_PRN"][i]), float(g["ch_acquiredFreq"][i]), float(g["ch_codePhase"][i])) for i in range(4)]
t0 = time.time(); ser, done = ctx.track(rec, chans, 400); print("trk wall s", time.time() - t0, ctx.timing(), done)
w = g["series"]
print("abs eq", np.array_equal(ser[:, 0], w[:, 0]))
for c in range(4):
    sc = np.sqrt(np.mean(w[c, 3] ** 2 + w[c, 7] ** 2))
    print("ch", c, "rms", sc, "max corr err", np.max(np.abs(ser[c, 3:9] - w[c, 3:9])), "rel", np.max(np.abs(ser[c, 3:9] - w[c, 3:9])) / sc,
          "codeFreq err", np.max(np.abs(ser[c, 1] - w[c, 1])), "carrFreq err", np.max(np.abs(ser[c, 2] - w[c, 2])))
print("us per ms-step:", ctx.timing()["track_ms"] * 1e3 / 400)
