#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into
HBM bytes per launch of the particle kernels, with the gfx950 correction of
MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 64 B per 128-B request for
wide (16 B/lane) coalesced streaming reads, so it is doubled; WRITE_SIZE is
exact.  Both counters are in KiB.

    python profiles/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> \
        <particles_per_gpu> <nx> [out.json]

Several (fetch, write) pairs may be merged: pass "a.csv,b.csv" for either.
"""
import collections
import csv
import json
import re
import sys

PARTICLE_KERNELS = ("k_push", "k_step_half", "k_step_full", "k_step_one", "k_step_sums", "k_deposit")


def short_name(full):
    for k in PARTICLE_KERNELS:
        if k in full:
            if k == "k_push":
                m = re.search(r"k_push<\d+, \d+, \w+, (\w+), (\w+)>", full)
                if m:
                    irk = "irk2" if m.group(1) in ("true", "1") else "irk1"
                    fused = "fused" if m.group(2) in ("true", "1") else "push_only"
                    return "k_push_%s_%s" % (fused, irk)
            return k
    return None


def mean_by_kernel(paths, counter):
    acc = collections.defaultdict(list)
    for path in paths.split(","):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = short_name(r["Kernel_Name"])
                if name and r["Counter_Name"] == counter:
                    acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch_csv, write_csv, n, nx = sys.argv[1], sys.argv[2], int(float(sys.argv[3])), int(sys.argv[4])
    out = sys.argv[5] if len(sys.argv) > 5 else None
    fetch, nf = mean_by_kernel(fetch_csv, "FETCH_SIZE")
    write, _ = mean_by_kernel(write_csv, "WRITE_SIZE")
    rows, by_kernel = [], {}
    for k in sorted(fetch):
        rd = fetch[k] * 1024 * 2.0      # KiB -> B, x2 gfx950 wide-read correction
        wr = write.get(k, 0.0) * 1024
        rows.append(dict(kernel=k, launches=nf[k], fetch_size_raw_kib=fetch[k], write_size_raw_kib=write.get(k),
                         read_bytes=rd, write_bytes=wr, hbm_bytes=rd + wr,
                         read_bytes_per_particle=rd / n, write_bytes_per_particle=wr / n))
        by_kernel[k] = rd + wr
    # the fused sub-step kernel has two instantiations (irk 1 / irk 2): bench.py
    # reports their mean launch as "k_push"
    fused = [r for r in rows if r["kernel"].startswith("k_push_fused")]
    if fused:
        by_kernel["k_push"] = sum(r["hbm_bytes"] * r["launches"] for r in fused) / sum(r["launches"] for r in fused)
    res = dict(particles_per_gpu=n, nx=nx, hbm_bytes_per_launch_by_kernel=by_kernel,
               compulsory_bytes_per_marker=dict(k_step_half=32.0, k_step_full=56.0, k_step_one=56.0, k_step_sums=56.0),
               compulsory_note="32 B read (x, v, w, p) + 24 B written (x, v, w) per marker and launch of a whole-step "
                               "kernel; a carry of -f0'/f0 (PIC1DP_CARRY=1, or the reference-order form: 8 B read + 8 B "
                               "written) is traffic the kernel chooses, reported apart by pic1dp_hip_kernel_bytes",
               reference_priced_bytes_per_update=80.0,
               correction="FETCH_SIZE x2 (gfx950 wide coalesced reads), WRITE_SIZE exact; KiB units",
               kernels=rows)
    print(json.dumps(res, indent=1))
    if out:
        with open(out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
