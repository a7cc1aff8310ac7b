"""Build recipe for libsgx.so (hipcc, gfx950 only, in-tree so the .so travels with the repo)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libsgx.so")
SOURCES = ["sgx_host.cpp", "sgx_synth.hip", "sgx_fft.hip", "sgx_acq.hip", "sgx_trk.hip", "sgx_trk_f32.hip", "sgx_trk_tp.hip", "sgx_trk2.hip", "sgx_trk3.hip", "sgx_trk_multi.hip", "sgx_trk_any.hip", "sgx_nav.hip", "sgx_navhost.cpp", "sgx_probe.hip", "sgx_geo.cpp"]
HEADERS = [os.path.join(CSRC, "sgx_trk_kernel.inc"), os.path.join(CSRC, "sgx_trk_math.h"), os.path.join(CSRC, "sgx_trk_common.h"), os.path.join(CSRC, "sgx_trk2_parts.h"), os.path.join(CSRC, "sgx_internal.h"), os.path.join(ROOT, "include", "sgx.h")]
# -ffp-contract=off: chip-boundary index math must round exactly like the reference's numpy
# expressions; fused multiply-adds are written explicitly where wanted.
EXTRA = os.environ.get("SGX_EXTRA_FLAGS", "").split()   # e.g. -DTRK_FINEPROF for the diagnosis build
FLAGS = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wall", "-Wno-unused-result", "-Wno-unused-value"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every translation unit for gfx950 and link libsgx.so. Returns the library path."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    cc = hipcc()
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src + ".o")
        objs.append(obj)
        cmd = [cc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
