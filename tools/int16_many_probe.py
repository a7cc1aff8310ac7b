"""(GPU box) Tracking of 8 .. 1024 channels of an int8 and of an int16 record (200 ms): kernel, member layout, channel-seconds per second."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
m = importlib.import_module("softgnss-python_amd")
s = m.Settings(); ctx = m.engine.get_context(s, 0); n = s.samplesPerCode
ms = 200
rec8 = m.synth.generate(m.synth.Scene.default(), m.synth.record_length(n, ms + 600))
a = m.AcquisitionResult(s, device=0); a.acquire(rec8[:11 * n]); a.preRun()
chans = [(int(c.PRN), float(c.acquiredFreq), float(c.codePhase)) for c in a.channels if c.PRN != 0]
r8 = ctx.upload(rec8)
r16 = ctx.upload_bytes((rec8.astype(np.int16) * 57).astype('<i2').view(np.int8))
for nch in (8, 64, 256, 1024):
    many8 = [(chans[i % 8][0], chans[i % 8][1], chans[i % 8][2] + (i // 8) * 2 * n) for i in range(nch)]
    many16 = [(p, f, 2 * (cp - 1) + 2 + 0) for (p, f, cp) in many8]   # byte offset of sample cp: codePhase bytes with skip 0 -> 2*cp
    for name, rec, ch, dt in (("int8", r8, many8, m._native.DT_INT8), ("int16", r16, [(p, f, 2 * cp) for (p, f, cp) in many8], m._native.DT_INT16)):
        ctx.track(rec, ch, 20, data_type=dt)
        ser, dn = ctx.track(rec, ch, ms, data_type=dt)
        tm = ctx.timing()
        print("%5d ch %-6s %8.2f ms  kernel %d members %d  locked %d  -> %.1f channel-seconds per second" % (nch, name, tm["track_ms"], tm["track_kernel"], tm["track_members"], int((dn == ms).sum()), nch * ms / tm["track_ms"]))
