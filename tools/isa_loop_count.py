#!/usr/bin/env python3
"""isa_loop_count.py -- static instruction mix of the marker loop of one kernel instantiation (cross-compiled,
no GPU): python tools/isa_loop_count.py 'k_step_one<5, 0, 2, true, 0>'   (PIC1DP_ISA_OPS=1: the loop's opcode histogram too)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the translation unit and, for kernels_step.hip, the distribution it is compiled for (pic1dp_amd/build.py):
#   PIC1DP_TU=kernels_step.hip PIC1DP_STEP_DIST=5 (default: the one-exp bump-on-tail unit of the default input)
src = os.path.join(ROOT, "pic1dp_amd", "csrc", os.environ.get("PIC1DP_TU", "kernels_step.hip"))
DIST_FLAG = ["-DPIC1DP_STEP_DIST=" + os.environ.get("PIC1DP_STEP_DIST", "5")] if src.endswith("kernels_step.hip") else []
out = "/tmp/isa_loop_count.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                "-munsafe-fp-atomics", "-x", "hip", "-S", "--cuda-device-only", src, "-o", out] +
               os.environ.get("PIC1DP_EXTRA_FLAGS", "").split() + DIST_FLAG, check=True, capture_output=True)
s = open(out).read()
want = sys.argv[1]
names = re.findall(r'^(_ZN6pic1dp[^\n:]*):', s, re.M)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
target = [n for n, d in zip(names, dem) if want in d][0]
i = s.index(target + ":")
j = s.index("s_endpgm", i)
ins, labels = [], {}
for l in (x.strip() for x in s[i:j].split("\n")):
    if not l or l.startswith((";", "//")):
        continue
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if l.startswith(".") or l.endswith(":"):
        continue
    ins.append(l)
best = None
for k, l in enumerate(ins):
    m = re.match(r'^s_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < k:
        a = labels[m.group(1)]
        nv = sum(1 for x in ins[a:k + 1] if x.startswith("v_"))
        if best is None or nv > best[0]:
            best = (nv, a, k)
nv, a, k = best
loop = ins[a:k + 1]
g = collections.Counter()
for x in loop:
    op = x.split()[0]
    g["VALU" if op.startswith("v_") else "LDS" if op.startswith("ds_") else "SALU" if op.startswith("s_") else
      "VMEM" if op.startswith(("global_", "buffer_", "flat_")) else "scratch" if op.startswith("scratch_") else "other"] += 1
print(want, "kernel", len(ins), "loop", len(loop), dict(g))
if os.environ.get("PIC1DP_ISA_OPS"):   # the opcode histogram of the loop, most frequent first
    ops = collections.Counter(x.split()[0] for x in loop)
    print("  ".join("%s %d" % kv for kv in ops.most_common(40)))
