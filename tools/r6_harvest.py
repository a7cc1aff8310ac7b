"""(round 6) gpurun_out/ of tools/r6_profile.sh -> the tracked files under profiles/ (r06_*: trace stats, PMC summaries, the
stamped phases with the TRUE cycle count of a block, the bench line)."""
import json, os, re, shutil, statistics
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
for f in os.listdir(os.path.join(G, "prof_r06", "summary")):
    shutil.copy(os.path.join(G, "prof_r06", "summary", f), os.path.join(P, f))
shutil.copy(os.path.join(G, "r06_bench_line.json"), os.path.join(P, "r06_bench_line.json"))
bench = json.loads(open(os.path.join(G, "r06_bench_line.json")).read().strip().splitlines()[-1])
L = open(os.path.join(G, "r06_phase.txt")).read().splitlines()
a, b, c, steps = [], [], [], []
for l in L:
    m = re.search(r"member\s+(\d+) cycles/block: release->publish (\d+)\s+publish->sums (\d+)\s+sums->release (\d+)", l)
    if m and int(m.group(1)) != 0:
        a.append(int(m.group(2))); b.append(int(m.group(3))); c.append(int(m.group(4)))
    if l.startswith("step "):
        steps.append(float(re.search(r"kernel ([\d.]+) ms", l).group(1)))
fa, fb, fc = (int(statistics.median(v)) for v in (a, b, c))
live = bench["track_kernel_ms"]
true_block = live * 1e-3 / 37000 * 2.4e9
chain = {"workload": {"channels": 8, "ms": 37000}, "final_pass": fa, "exchange": fb, "loop_filter": fc,
         "block": int(round(true_block)),
         "unit": "shader cycles (an s_memtime tick is one, the SIMDs run at 2.40 GHz: profiles/r06_clock.txt). final_pass / exchange / "
                 "loop_filter: medians over members 1..19 of three runs of the SGX_TRK_PROFILE build (three stamps per block on the PLL "
                 "wave; %.2f ms against %.2f ms unprofiled) - true cycle counts of the intervals they bracket, which do NOT tile the "
                 "period: ~250-350 cycles per block lie between the stamp behind the barrier and the next one at the loop's top. "
                 "block: the unprofiled kernel's HIP-event time / 37 000 x 2.40 GHz. profiles/r06_timeline.txt has the events of "
                 "every wave." % (statistics.mean(steps), live),
         "source": "profiles/r06_trk_phase_profile.txt, profiles/r06_timeline.txt"}
json.dump(chain, open(os.path.join(P, "r06_trk_chain.json"), "w"), indent=1)
out = ["Round 6, tracking kernel trk3_kernel (unchanged on the chain since round 5: 8 channels x 20 units = 160 workgroups of 448",
       "threads, one per CU), MI355X.  bash tools/r6_profile.sh: SGX_TRK_PROFILE=1 python3 tools/step_profile.py 37000, three runs.",
       "Cycles are shader cycles (2.40 GHz: profiles/r06_clock.txt); the profiled kernel takes %.2f ms against %.2f ms without the" % (statistics.mean(steps), live),
       "stamps.  Medians over members 1..19: final pass %d, exchange %d, loop filter + barrier %d - sum %d of the profiled period's" % (fa, fb, fc, fa + fb + fc),
       "%d cycles (%.3f us x 2.4 GHz): the three intervals do not tile the period, see profiles/r06_timeline.txt for all events." % (statistics.mean(steps) * 1e-3 / 37000 * 2.4e9, statistics.mean(steps) * 1e3 / 37000),
       "One block of the shipped kernel: %.4f us = %d cycles." % (live * 1e3 / 37000, round(true_block)), ""]
out += ["   " + l for l in L if l.startswith("[sgx trk2 profile]") or l.startswith("step ")][:24]
open(os.path.join(P, "r06_trk_phase_profile.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:8]))
print({k: bench[k] for k in ("x_realtime", "ms_per_step", "acquire_ms", "track_kernel_ms", "host_glue_ms")})
