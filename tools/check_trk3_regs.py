"""Build gate for csrc/sgx_trk3.hip: the filter waves' polls leave loads in flight whose destinations are the PHYSICAL
registers v[244:255]; nothing the compiler allocates may touch them.  Compiles the file to assembly with the given extra
flags and fails (exit 1) if v244..v255 appear outside an inline-asm statement, or if the kernel spills vector registers
or uses AGPRs.  Used by softgnss-python_amd/build.py, tools/build_variant.sh and tests/test_cabi_and_host.py.
    python3 tools/check_trk3_regs.py [extra hipcc flags ...]
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(extra=(), hipcc="/opt/rocm/bin/hipcc"):
    """-> (ok, message)"""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "trk3.s")
        r = subprocess.run([hipcc] + list(extra) + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                            "-x", "hip", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "softgnss-python_amd", "csrc"),
                            "-S", "--cuda-device-only", "-o", out,
                            os.path.join(ROOT, "softgnss-python_amd", "csrc", "sgx_trk3.hip")],
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


if __name__ == "__main__":
    ok, msg = check(sys.argv[1:])
    print("[check_trk3_regs] " + ("ok: " if ok else "FAILED: ") + msg)
    sys.exit(0 if ok else 1)
