"""(round 5) gpurun_out/ of tools/r5_profile.sh + the default bench run -> the tracked files under profiles/ (r05_*)."""
import json, os, re, shutil, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
for f in os.listdir(os.path.join(G, "prof_r05", "summary")):
    shutil.copy(os.path.join(G, "prof_r05", "summary", f), os.path.join(P, f))
shutil.copy(os.path.join(G, "r05_bench_line.json"), os.path.join(P, "r05_bench_line.json"))
L = open(os.path.join(G, "r05_phase.txt")).read().splitlines()
a, b, c = [], [], []
for l in L[:L.index(next(x for x in L if x.startswith("== the round-4")))]:
    m = re.search(r"member\s+(\d+) cycles/block: release->publish (\d+)\s+publish->sums (\d+)\s+sums->release (\d+)", l)
    if m and int(m.group(1)) != 0:
        a.append(int(m.group(2))); b.append(int(m.group(3))); c.append(int(m.group(4)))
fa, fb, fc = (int(statistics.median(v)) for v in (a, b, c))
step = next(l for l in L if l.startswith("step "))
kms = float(re.search(r"kernel ([\d.]+) ms", step).group(1))
bench = json.loads(open(os.path.join(G, "r05_bench_line.json")).read().strip().splitlines()[-1])
live = bench["track_kernel_ms"]
ticks = fa + fb + fc
out = ["""Round 5, tracking kernel trk3_kernel (speculative latency mode, speculation two blocks ahead: 8 channels x 20 units =
160 workgroups of 448 threads, one per CU; sgx_trk3.hip), MI355X.  Command (GPU box):  bash tools/r5_profile.sh  ->
SGX_TRK_PROFILE=1 python3 tools/step_profile.py 37000  plus diagnosis builds (tools/build_variant.sh).  Cycles are
s_memtime ticks (%.2f GHz in this run: %d ticks = %.4f us) per code period; SGX_TRK_PROFILE adds three time stamps per
period (~100 cycles of them inside "sums->release"; the profiled kernel takes %.2f ms against %.2f ms without the stamps).

1. Per-member phase times, channel 0 (first of four identical runs; medians over members 1..19 of all four runs:
   final pass %d, exchange %d, loop filter + barrier %d; round 4: 835 / 965 / 1010).  release->publish = the FINAL
   PASS of the member's map waves; publish->sums = exchange as seen by that member's PLL wave (includes waiting for the
   slowest member); sums->release = the PLL wave's loop filter + barrier.  Member = unit (member 0 also posts the records)."""
       % (ticks / (kms * 1e3 / 37000) / 1e3, ticks, kms * 1e3 / 37000, kms, live, fa, fb, fc)]
n = 0
for l in L:
    if l.startswith("[sgx trk2 profile]") and n < 20:
        out.append("   " + l); n += 1
out.append("   " + step)
names = {"== barrier": "2. ", "== the filter waves": "3. ", "== poll": "4. ", "== blocks off": "5. "}
sec = False
for l in L:
    k = next((v for p_, v in names.items() if l.startswith(p_)), None)
    if k:
        out.append("\n" + k + l[3:]); sec = True
    elif l.startswith("== figures") or l.startswith("== -DT3_PROF_DLL"):
        out.append("   " + l[3:])
    elif l.startswith("=="):
        sec = False
    elif sec and l.startswith(("[waveprof]", "[t3 ", "pp0", "pp1", "pd0", "pd1")):
        out.append("   " + l)
rounds = []
for r in (2, 3, 4):
    d = json.loads(open(os.path.join(P, "r0%d_bench_line.json" % r)).read().strip().splitlines()[-1])
    rounds.append((r, d["track_kernel_ms"], d["x_realtime"]))
out.append("""
6. Round by round (same workload, 37 000 blocks x 8 channels; from profiles/rNN_bench_line.json: kernel ms, x real time
   of the whole step incl. acquisition and host):
   round 2  trk2 (10 members, no speculation)               %.2f ms  %.0fx
   round 3  trk3 (speculation one block ahead)              %.2f ms  %.0fx
   round 4  trk3 + exchange through same-XCD L2             %.2f ms  %.0fx
   round 5  trk3 two blocks ahead, part A/B, 3 looks        %.2f ms  %.0fx   (%.3f us per code period)"""
           % (rounds[0][1], rounds[0][2], rounds[1][1], rounds[1][2], rounds[2][1], rounds[2][2], live, bench["x_realtime"],
              bench["us_per_code_period"]))
open(os.path.join(P, "r05_trk_phase_profile.txt"), "w").write("\n".join(out) + "\n")
json.dump({"workload": {"channels": 8, "ms": 37000}, "final_pass": fa, "exchange": fb, "loop_filter": fc, "block": ticks,
           "unit": "s_memtime ticks per code period, medians over members 1..19 of four runs of the SGX_TRK_PROFILE build "
                   "(three stamps per block, ~100 ticks inside loop_filter; %.2f ms against %.2f ms unprofiled = %d ticks)"
                   % (kms, live, round(ticks * live / kms)),
           "block_unprofiled": round(ticks * live / kms), "source": "profiles/r05_trk_phase_profile.txt"},
          open(os.path.join(P, "r05_trk_chain.json"), "w"), indent=1)
print(fa, fb, fc, ticks, kms, live)
