"""Diagnostics of the position-fix scene (tests/nav_scene.py): DOP, tracker vs generator timing, fix error."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nav_scene
from oracle import softgnss_oracle as orc
m = importlib.import_module("softgnss-python_amd")
sc, truth = nav_scene.build()
s = m.Settings()
s.samplingFreq, s.IF, s.msToProcess, s.numberOfChannels = 16368000.0, 4130400.0, 37000.0, len(truth['prns'])
ctx = m.engine.get_context(s, 0)
n = s.samplesPerCode
rec = ctx.synth(sc, m.synth.record_length(n, 37000))
a = m.AcquisitionResult(s, device=0); a.acquire(m.DeviceSignal(rec, 0, 11 * n)); a.preRun()
t = m.TrackingResult(a, device=0); t.track(m.DeviceFile(rec))
nav = m.NavigationResult(t, device=0); nav.postNavigate()
sol = nav.solutions[0]
first, active = nav.findPreambles()
print("first", first, "DOP (G,P,H,V,T) epoch 0:", sol.DOP[:, 0])
rows = [np.asarray(r.absoluteSample, dtype=np.float64) for r in t.results]
for c_ in active:
    prn = int(t.results[c_].PRN)
    sat_ = [q for q in sc.sats if q["prn"] == prn][0]
    arrival = truth["arrival_samples"][truth["prns"].index(prn)]
    per = 1023.0 * 2 ** 32 / sat_["code_fcw"]
    kk = np.arange(0, 37000 - int(first[c_]))
    d = rows[c_][int(first[c_]):] - (arrival + kk * per)
    print("PRN %2d: block start - true code boundary (samples): min %.3f max %.3f mean %.3f at epochs %s" % (
        prn, d.min(), d.max(), d.mean(), np.round(d[::500][:8], 2)))
xyz = np.stack([sol.X, sol.Y, sol.Z])[:, :63]
err = np.linalg.norm(xyz - truth["rx"][:, None], axis=0)
print("fix error:", np.round(err[:16], 1))
# the same least squares on ideal (unquantised) arrival times
so = orc.OracleSettings(samplingFreq=s.samplingFreq, IF=s.IF, numberOfChannels=len(truth['prns']), msToProcess=37000.0)
prn_act = [int(t.results[c_].PRN) for c_ in active]
for k in (0, 2, 10):
    tt = []
    for c_ in active:
        prn = prn_act[list(active).index(c_)]
        sat_ = [q for q in sc.sats if q["prn"] == prn][0]
        per = 1023.0 * 2 ** 32 / sat_["code_fcw"]
        tt.append((truth["arrival_samples"][truth["prns"].index(prn)] + 500 * k * per) / n)
    tt = np.array(tt); pr = (tt - np.floor(tt.min()) + so.startOffset) * so.c / 1000
    sat, clk = orc.satpos(truth["tow"] + 0.5 * k, prn_act, truth["eph_table"])
    p, el, az, dop = orc.least_square_pos(sat, pr + clk * so.c, so.c, True)
    print("epoch %d ideal-timing fix error %.2f m" % (k, np.linalg.norm(p[:3] - truth["rx"])))
