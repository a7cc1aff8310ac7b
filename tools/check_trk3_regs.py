"""Build gate for csrc/sgx_trk3.hip (softgnss-python_amd/build.py: check_trk3_registers): fails (exit 1) if the compiler
touches the polls' reserved registers v[244:255], spills vector registers or uses AGPRs, with the given extra flags.
    python3 tools/check_trk3_regs.py [extra hipcc flags ...]
"""
import importlib.util, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("sgx_build", os.path.join(ROOT, "softgnss-python_amd", "build.py"))
build = importlib.util.module_from_spec(spec)
spec.loader.exec_module(build)

if __name__ == "__main__":
    ok, msg = build.check_trk3_registers(sys.argv[1:])
    print("[check_trk3_regs] " + ("ok: " if ok else "FAILED: ") + msg)
    sys.exit(0 if ok else 1)
