"""TEST INFRASTRUCTURE: builds tests/hipsim/_build/libghicp_sim.so = the library's own .hip sources compiled with g++ against the
host SIMT interpreter (tests/hipsim/include).  Never used by the package: gh-icp_amd/api.py loads libghicp_hip.so or raises."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "gh-icp_amd", "csrc")
ASAN = os.environ.get("HIPSIM_ASAN") == "1"  # LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0
TSAN = os.environ.get("HIPSIM_TSAN") == "1"  # LD_PRELOAD=$(gcc -print-file-name=libtsan.so): lanes are TSan fibers, barriers / collectives are its sync points
OUT_DIR = os.path.join(HERE, "_build_asan" if ASAN else ("_build_tsan" if TSAN else "_build"))
LIB = os.path.join(OUT_DIR, "libghicp_sim.so")
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-g1", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-pthread", "-w", "-fno-extern-tls-init",
         "-I", os.path.join(HERE, "include"), "-I", CSRC] + (["-fsanitize=address", "-fno-omit-frame-pointer", "-g"] if ASAN else []) + (["-fsanitize=thread", "-fno-omit-frame-pointer", "-g", "-O1"] if TSAN else [])


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "ghicp_c.h")]
    for d, _, fs in os.walk(os.path.join(HERE, "include")):
        hdrs += [os.path.join(d, f) for f in fs]
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip")) + [os.path.join(HERE, "hipsim.cpp")]
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OUT_DIR, os.path.basename(s) + ".o")
        objs.append(o)
        if _newer(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        r = subprocess.run(["g++"] + FLAGS + ["-c", s, "-o", o], capture_output=True, text=True)
        return s, r.returncode, r.stderr

    failed = False
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for s, rc, err in ex.map(cc, jobs):
            if rc != 0:
                failed = True
                sys.stderr.write("hipsim build: %s failed\n%s\n" % (os.path.basename(s), err[-6000:]))
            elif verbose:
                print("compiled", os.path.basename(s))
    if failed:
        raise RuntimeError("hipsim build failed")
    if jobs or not os.path.exists(LIB):
        subprocess.run(["g++", "-shared", "-pthread"] + (["-fsanitize=address"] if ASAN else []) + (["-fsanitize=thread"] if TSAN else []) + ["-o", LIB] + objs, check=True)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
