#!/usr/bin/env python3
"""kernel_resources.py -- VGPRs / scratch / occupancy of every kernel of the marker / field translation units as hipcc
reports them (-Rpass-analysis=kernel-resource-usage); runs on the CPU (cross-compile).
    python tools/kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the translation unit and, for kernels_step.hip, the distribution it is compiled for (pic1dp_amd/build.py):
#   PIC1DP_TU=kernels_step.hip PIC1DP_STEP_DIST=5 (default: the one-exp bump-on-tail unit of the default input)
src = os.path.join(ROOT, "pic1dp_amd", "csrc", os.environ.get("PIC1DP_TU", "kernels_step.hip"))
DIST_FLAG = ["-DPIC1DP_STEP_DIST=" + os.environ.get("PIC1DP_STEP_DIST", "5")] if src.endswith("kernels_step.hip") else []
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-munsafe-fp-atomics", "-c", "-x", "hip", src, "-o", "/tmp/kernel_resources.o",
                    "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("PIC1DP_EXTRA_FLAGS", "").split() + DIST_FLAG,
                   capture_output=True, text=True)
blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
if r.returncode != 0 or not blocks:
    sys.exit("compile failed:\n" + r.stderr[-3000:])
names = [b.split()[0] for b in blocks]
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(re.escape(k) + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    d = d.replace("pic1dp::(anonymous namespace)::", "").replace("void ", "")
    d = re.sub(r"\(.*", "", d)
    if flt in d:
        print("%-60s VGPR %3d  SGPR %3d  scratch %4d  waves/SIMD %d  spill %d" % (
            d[:60], g("VGPRs"), g("TotalSGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), g("VGPRs Spill")))
