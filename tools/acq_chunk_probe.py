"""Acquisition time against the PRN chunk size of the correlation batch (SGX_ACQ_CHUNK_ROWS): GPU box."""
import importlib, os, sys
os.environ["SGX_ACQ_SPLIT_EVENT"] = "1"   # (the coarse / fine split is measured only with the event between them)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = importlib.import_module("softgnss-python_amd")
s = m.Settings(); ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
rec = ctx.synth(m.synth.Scene.default(), 21 * n)
for rows in (174, 232, 290, 348, 406, 2048):
    os.environ["SGX_ACQ_CHUNK_ROWS"] = str(rows)
    for nb, nc, label in ((2, False, "2x1ms"), (10, True, "10ms noncoh")):
        sig = m.DeviceSignal(rec, 0, (10 + nb) * n if nc else 11 * n)
        a = m.AcquisitionResult(s, device=0); a.acquire(sig, n_blocks=nb, noncoh=nc)
        ts = []
        for _ in range(4):
            a.acquire(sig, n_blocks=nb, noncoh=nc); ts.append(ctx.timing())
        print("chunk rows %4d %-12s acquire %.3f ms (coarse %.3f fine %.3f)" % (rows, label, min(t["acquire_ms"] for t in ts), min(t["acq_coarse_ms"] for t in ts), min(t["acq_fine_ms"] for t in ts)))
