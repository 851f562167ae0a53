"""Build the gfx950 shared library libpic1dp_hip.so in-tree with hipcc.

    python pic1dp_amd/build.py [--force]      (run as a script: importing the
                                               package needs the built library)

hipcc cross-compiles for gfx950 without a GPU.  The library is the whole
product: HIP kernels + C ABI + native host loader.  It is git-ignored but
travels to the GPU box with gpurun.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpic1dp_hip.so")
SOURCES = ["kernels.hip", "capi.cpp", "loader.cpp", "multirand.cpp", "optimize.cpp"]
HEADERS = ["kernels.hpp", "loader.hpp", "multirand.hpp", "optimize.hpp", "rccl_dyn.hpp",
           os.path.join("..", "..", "include", "pic1dp_hip.h")]

# -ffp-contract=off : products and sums round separately, like the reference's
#                     plain -O3 x86-64 build (no FMA) -- needed for bit-exact
#                     positions / cell indices
# -munsafe-fp-atomics: native ds_add_f64 / global_atomic_add_f64, no CAS loops
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-result",
         "-x", "hip"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libpic1dp_hip.so")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    # tuning builds: PIC1DP_EXTRA_FLAGS="-DPIC1DP_NT=0" PIC1DP_LIB_OUT=/path/variant.so
    # (load one with PIC1DP_LIB=/path/variant.so)
    out = os.environ.get("PIC1DP_LIB_OUT") or LIB
    extra = os.environ.get("PIC1DP_EXTRA_FLAGS", "").split()
    tmp = out + ".tmp.%d" % os.getpid()
    cmd = [hipcc()] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + [
        "-o", tmp, "-ldl", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
