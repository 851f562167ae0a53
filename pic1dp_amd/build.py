"""Build the gfx950 shared libraries in-tree with hipcc.

    python pic1dp_amd/build.py [--force]      (run as a script: importing the
                                               package needs the built library)

hipcc cross-compiles for gfx950 without a GPU.  lib/libpic1dp_hip.so is the whole
product: HIP kernels + C ABI + native host loader; lib/libpic1dp_probe.so holds the
measurement / test-support kernels (include/pic1dp_probe.h), built from the same
device headers.  Both are git-ignored but travel to the GPU box with gpurun.

The translation units are compiled in parallel (objects under csrc/build/);
kernels_step.hip once per distribution (-DPIC1DP_STEP_DIST=0..5).
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpic1dp_hip.so")
PROBE_LIB = os.path.join(LIBDIR, "libpic1dp_probe.so")
STEP_DISTS = (0, 1, 2, 3, 4, 5)
# (source, extra flags, object stem)
PRODUCT_UNITS = [("kernels_step.hip", ["-DPIC1DP_STEP_DIST=%d" % d], "kernels_step_d%d" % d) for d in STEP_DISTS] + [
    ("kernels_push.hip", [], "kernels_push"), ("kernels_field.hip", [], "kernels_field"),
    ("kernels_diag.hip", [], "kernels_diag"), ("kernels_opt.hip", [], "kernels_opt"), ("step_dispatch.cpp", [], "step_dispatch"),
    ("capi.cpp", [], "capi"), ("capi_step.cpp", [], "capi_step"), ("capi_comm.cpp", [], "capi_comm"), ("capi_diag.cpp", [], "capi_diag"),
    ("capi_optimize.cpp", [], "capi_optimize"), ("loader.cpp", [], "loader"), ("multirand.cpp", [], "multirand"),
    ("optimize.cpp", [], "optimize"), ("species.cpp", [], "species"), ("hostcheck.cpp", [], "hostcheck")]
PROBE_UNITS = [("probe.hip", [], "probe"), ("optcheck.cpp", [], "optcheck")]
PROBE_SHARED = ["species", "hostcheck", "optimize", "multirand"]      # objects of the product the probe library links as well
HEADERS = ["kernels.hpp", "device_math.hpp", "device_field.hpp", "device_diag.hpp", "device_xchg.hpp", "step_args.hpp", "check_values.hpp", "loader.hpp",
           "multirand.hpp", "optimize.hpp", "rccl_dyn.hpp", "ctx.hpp",
           os.path.join("..", "..", "include", "pic1dp_hip.h"), os.path.join("..", "..", "include", "pic1dp_probe.h")]

# -ffp-contract=off : products and sums round separately, like the reference's
#                     plain -O3 x86-64 build (no FMA) -- needed for bit-exact
#                     positions / cell indices
# -munsafe-fp-atomics: native ds_add_f64 / global_atomic_add_f64, no CAS loops
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-result", "-x", "hip"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libpic1dp_hip.so")


def stale():
    for lib in (LIB, PROBE_LIB):
        if not os.path.exists(lib):
            return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(PROBE_LIB))
    srcs = sorted({u[0] for u in PRODUCT_UNITS + PROBE_UNITS})
    deps = [os.path.join(CSRC, f) for f in srcs + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(cc, unit, extra, objdir, verbose):
    src, flags, stem = unit
    obj = os.path.join(objdir, stem + ".o")
    cmd = [cc] + FLAGS + extra + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return obj


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    # one build at a time per tree: the objects of the translation units are shared files
    import fcntl
    lock = open(os.path.join(LIBDIR, ".build.lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not stale():      # another process built it while this one waited
            return LIB
        return _build_locked(verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(verbose):
    # tuning builds: PIC1DP_EXTRA_FLAGS="-DPIC1DP_NT=0" PIC1DP_LIB_OUT=/path/variant.so
    # (load one with PIC1DP_LIB=/path/variant.so); their objects go to a directory of their own
    out = os.path.abspath(os.environ.get("PIC1DP_LIB_OUT") or LIB)   # (the link runs in csrc/)
    extra = os.environ.get("PIC1DP_EXTRA_FLAGS", "").split()
    variant = out != LIB
    objdir = OBJDIR if not variant else OBJDIR + "." + os.path.basename(out)
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    units = PRODUCT_UNITS + ([] if variant else PROBE_UNITS)
    jobs = int(os.environ.get("PIC1DP_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(lambda u: _compile(cc, u, extra, objdir, verbose), units))
    by_stem = {u[2]: o for u, o in zip(units, objs)}

    def link(target, stems):
        tmp = target + ".tmp.%d" % os.getpid()
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [by_stem[s] for s in stems] + [
            "-o", tmp, "-ldl", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(tmp, target)

    link(out, [u[2] for u in PRODUCT_UNITS])
    if not variant:
        link(PROBE_LIB, [u[2] for u in PROBE_UNITS] + PROBE_SHARED)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
