"""(round 6 diagnosis) Where a block's period goes: mean event times of every wave of channel 0, relative to the barrier
release of the wave's own member, from the -DT3_TIMELINE variants of sgx_trk3.hip (tools/r6_timeline.sh builds them: one
event of each role per build).  GPU box:  python3 tools/r6_timeline.py > gpurun_out/r06_timeline.txt
"""
import os, re, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = re.compile(r"\[t3 tl\] unit (\d+) wave (\d+) par (\d+) ev (\d+) mean ([\d.]+)")
MAP_EV = {1: "parameters in registers", 2: "tested, group phasor", 3: "arms", 4: "reduced in rows", 5: "publish issued / handed on", 6: "at the barrier"}
FLT_EV = {1: "poll entered", 2: "sums found", 3: "nco", 4: "at the barrier", 6: "member's publish stamp"}
res = {}      # (role, ev, par) -> list over units of (t - release of the unit)
periods = []
for e in range(1, 7):
    lib = os.path.join(ROOT, "softgnss-python_amd", "lib", "variants", "libsgx_tl%d.so" % e)
    if not os.path.exists(lib):
        continue
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_profile.py"), "37000"], env=dict(os.environ, SGX_LIB=lib),
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "r06_timeline_raw_tl%d.txt" % e), "w").write(out)
    T = {}
    for m in pat.finditer(out):
        T[(int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)))] = float(m.group(5))
    ktime = [l for l in out.splitlines() if l.startswith("step ")]
    units = sorted({k[0] for k in T})
    if not units:
        print("variant tl%d: no stamps\n%s" % (e, out[-1500:]))
        continue
    per = np.mean([T[(u, 4, 1, 0)] - T[(u, 4, 0, 0)] for u in units])
    periods.append(per)
    print("variant tl%d: %s; period %.0f cycles" % (e, ktime[0] if ktime else "?", per))
    for par in (0, 1):
        rel = np.array([T[(u, 4, par, 0)] for u in units])
        line = "   %s blocks: members' releases after the earliest one: %s" % ("even" if par == 0 else "odd", " ".join("%.0f" % (r - rel.min()) for r in rel))
        if (units[0], 4, par, 6) in T:
            pub = np.array([T[(u, 4, par, 6)] for u in units])
            fnd = np.array([T[(u, 4, par, 2)] for u in units])
            line += "\n      publishes after the earliest release: %s\n      last publish -> PLL waves' finds: %s" % (
                " ".join("%.0f" % (x - rel.min()) for x in pub), " ".join("%.0f" % (x - pub.max()) for x in fnd))
        print(line)
    for (u, w, par, ev), t in T.items():
        if ev == 0:
            continue
        role = "pll" if w == 4 else ("dll" if w == 5 else "map")
        key = (role, ev, par, "final" if (role != "map" or (w >> 1) == par) else "spec")
        res.setdefault(key, []).append((u, w, t - T[(u, 4, par, 0)]))
print()
for par in (0, 1):
    print("==== %s blocks: cycles after the member's barrier release (mean over members 1..19 | member 0 | min .. max over members)" % ("even" if par == 0 else "odd"))
    for role, names in (("map", MAP_EV), ("pll", FLT_EV), ("dll", FLT_EV)):
        for kind in ("final", "spec"):
            for ev in sorted(names):
                v = res.get((role, ev, par, kind))
                if not v:
                    continue
                v = [x for x in v if abs(x[2]) < 1e5]   # (a printf line cut short by another workgroup's)
                a = np.array([x[2] for x in v if x[0] != 0])
                a0 = np.array([x[2] for x in v if x[0] == 0])
                label = ("map waves, final pass" if kind == "final" else "map waves, the other set") if role == "map" else role + " wave"
                print("  %-26s %-28s %7.0f | %7.0f | %7.0f .. %.0f" % (label, names[ev], a.mean(), a0.mean() if len(a0) else float("nan"), a.min(), a.max()))
    print()
print("mean period over the variants: %.0f cycles (the shipped build: kernel ms x 2.4e6 / 37 000)" % np.mean(periods))
