"""(round 6 diagnosis) Clock audit: what an s_memtime tick is and what the SIMDs clock at - alone, beside an fp64 load on
every CU, and beside trk3_kernel at 1 and 8 channels (tools/ubench_clock.hip).  GPU box:
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench_clock.hip -o tools/bin/libclk.so   (done by the caller)
    python3 tools/clock_probe.py > gpurun_out/r06_clock.txt
"""
import ctypes, importlib, os, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
clk = ctypes.CDLL(os.path.join(ROOT, "tools", "bin", "libclk.so"))
clk.clk_probe.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
NAMES = ["v_fma_f64 (dependent)", "v_add_f32 (dependent)", "s_nop 15 (16 cycles each, per cycle)",
         "s_sleep 8 (512 cycles each, per cycle)", "v_fma_f64, long stretch"]


def probe(label, spin=20):
    res = (ctypes.c_ulonglong * 20)()
    ev = ctypes.c_double(0)
    rc = clk.clk_probe(res, spin, ctypes.byref(ev))
    assert rc == 0, rc
    print("== %s (probe kernel %.3f ms by HIP events)" % (label, ev.value))
    tot_t = tot_r = 0
    for k, name in enumerate(NAMES):
        t, r, n = res[4 * k], res[4 * k + 1], res[4 * k + 2]
        tot_t += t
        tot_r += r
        if n == 0 or r == 0:
            continue
        print("   %-40s %9d instr/cycles: %7.3f s_memtime ticks each, %7.3f ns each; s_memtime runs at %.4f GHz"
              % (name, n, t / n, 10.0 * r / n, t / (10.0 * r)))
    print("   all five chains: %.0f s_memtime ticks in %.1f us of s_memrealtime (%.4f GHz); HIP events see %.1f us"
          % (tot_t, tot_r * 0.01, tot_t / (10.0 * tot_r), ev.value * 1e3))
    sys.stdout.flush()
    return res


def main():
    for i in range(3):
        probe("alone, run %d" % i)
    # beside an fp64 load on every CU
    for n_wg in (256, 1024):
        clk.clk_load_start(n_wg, 60)
        time.sleep(0.002)
        probe("beside %d workgroups x 256 threads of dependent fp64 FMAs" % n_wg, spin=40)
        clk.clk_load_wait()
    # beside the production kernel
    m = importlib.import_module("softgnss-python_amd")
    for n_ch in (8, 1):
        s = m.Settings()
        s.msToProcess = 37000.0
        s.numberOfChannels = n_ch
        ctx = m.engine.get_context(s, 0)
        n = s.samplesPerCode
        rec = ctx.synth(m.synth.Scene.default(), m.synth.record_length(n, 37000))
        sig = m.DeviceSignal(rec, 0, 11 * n)
        a = m.AcquisitionResult(s, device=0)
        a.acquire(sig)
        a.preRun()
        out = {}

        def run():
            t = m.TrackingResult(a, device=0)
            t.track(m.DeviceFile(rec))
            out["ms"] = t.kernel_ms

        run()
        print("trk3_kernel, %d channel(s), alone: %.3f ms" % (n_ch, out["ms"]))
        for rep in range(2):
            th = threading.Thread(target=run)
            th.start()
            time.sleep(0.008)
            probe("beside trk3_kernel, %d channel(s), rep %d" % (n_ch, rep), spin=150)
            th.join()
            print("   trk3_kernel beside the probe: %.3f ms (37 000 blocks: %.4f us per block)" % (out["ms"], out["ms"] / 37.0))
    probe("alone again")


if __name__ == "__main__":
    main()
