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


def check_trk3_registers(extra=(), cc=None):
    """Build gate for csrc/sgx_trk3.hip: the filter waves' polls leave loads in flight whose destinations are the PHYSICAL
    registers v[244:255]; nothing the compiler allocates may touch them.  Compiles the file to assembly (with `extra`
    flags in front of the build's own) and returns (ok, message): not ok if v244..v255 appear outside an inline-asm
    statement, or if the kernel spills vector registers or uses AGPRs.  Run by build(), tools/build_variant.sh (through
    tools/check_trk3_regs.py) and tests/test_cabi_and_host.py."""
    import re
    import tempfile
    cc = cc or hipcc()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "trk3.s")
        flags = [f for f in FLAGS if f != "-Wall"]
        r = subprocess.run([cc] + list(extra) + flags + ["-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "sgx_trk3.hip")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            return False, r.stderr.decode(errors="replace")[-2000:]
        reserved = re.compile(r"\bv(24[4-9]|25[0-5])\b|\bv\[(\d+):(\d+)\]")
        in_asm, hits_in, hits_out, spills, agprs = False, 0, [], None, None
        for line in open(out):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif t.startswith(".vgpr_spill_count:"):
                spills = int(t.split(":")[1])
            elif t.startswith(".agpr_count:"):
                agprs = int(t.split(":")[1])
            elif t and not t.startswith((";", ".")):
                for mm in reserved.finditer(t):
                    if mm.group(1) or (int(mm.group(3)) >= 244 and int(mm.group(2)) <= 255):
                        if in_asm:
                            hits_in += 1
                        else:
                            hits_out.append(t)
                        break
    if hits_in < 10:
        return False, "the polls' asm statements were not found (%d lines on v[244:255])" % hits_in
    if hits_out:
        return False, "v[244:255] are used outside the polls' asm statements, e.g. '%s' (%d lines)" % (hits_out[0], len(hits_out))
    if spills != 0:
        return False, "trk3_kernel spills %s vector registers" % spills
    if agprs not in (0, None):
        return False, "trk3_kernel uses %s AGPRs" % agprs
    return True, "v[244:255] untouched outside the polls (%d lines inside), no spills" % hits_in


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
    # (beside the compiles: the register gate of the speculative tracking kernel)
    import concurrent.futures
    gate = concurrent.futures.ThreadPoolExecutor(max_workers=1).submit(check_trk3_registers, tuple(EXTRA), cc)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    ok, msg = gate.result()
    if not ok:
        raise RuntimeError("sgx_trk3.hip failed its register gate: %s" % msg)
    if verbose:
        print("[check_trk3_registers] " + msg)
    link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
