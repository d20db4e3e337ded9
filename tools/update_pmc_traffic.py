#!/usr/bin/env python3
"""Re-derive profiles/pmc_traffic.json's per-launch HBM-side bytes from a generation of PMC summaries (tools/show_pmc.py output):
   python tools/update_pmc_traffic.py profiles/r06/pmc a
bytes per launch = 2 x raw FETCH_SIZE + WRITE_SIZE (the factor 2: see the file's _comment).  Only the workloads that have a summary in the
generation are touched; every touched entry's `source` is re-pointed at that summary."""
import json
import os
import re
import sys

MIB = 1 << 20
PICK = {   # workload -> substrings of the kernel names whose launches are averaged (weighted by launch count)
    "plume3d_slab_jacobi": ["jacobi3d_march2_kernel<false, false, 3"],
    "plume3d_256_jacobi": ["jacobi3d_march2_kernel<false, false, 3"],
    "plume3d_hbm_jacobi": ["jacobi3d_march2_kernel<false, false, 3"],
    "rt2d_2048_jacobi": ["jacobi2d_wg_kernel<8, 8>"],
    "plume2d_1024_jacobi": ["jacobi2d_wg_kernel<8, 8>"],
    "plume2d_1024_cnn": ["conv3_wino4_kernel<false", "conv3_wino3_kernel<2, 2, false>", "conv3_wino3_kernel<1, 2, false>"],
    "plume3d_256_cnn": ["conv3_wino4_kernel<true", "conv3_wino3_kernel<2, 2, true>", "conv3_wino3_kernel<1, 2, true>"],
}


def rows(path):
    out = []
    for line in open(path):
        m = re.match(r"(.*?)\s+n=\s*(\d+)\s+FETCH_SIZE=\s*([\d.]+) MiB\s+WRITE_SIZE=\s*([\d.]+) MiB", line)
        if m:
            out.append((m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4))))
    return out


def main():
    pdir, gen = sys.argv[1], sys.argv[2]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tfile = os.path.join(repo, "profiles", "pmc_traffic.json")
    T = json.load(open(tfile))
    for w, keys in PICK.items():
        f = os.path.join(pdir, f"{gen}_{w}_pmc_summary.txt")
        if not os.path.exists(f) or w not in T:
            continue
        sel = [r for r in rows(f) if any(k in r[0] for k in keys)]
        n = sum(r[1] for r in sel)
        if not n:
            continue
        fetch = sum(r[1] * r[2] for r in sel) / n
        write = sum(r[1] * r[3] for r in sel) / n
        old = T[w].get("bytes_per_launch")
        T[w]["bytes_per_launch"] = int(round((2 * fetch + write) * MIB))
        T[w]["fetch_size_kib_raw"] = int(round(fetch * 1024)); T[w]["write_size_kib_raw"] = int(round(write * 1024))
        T[w]["source"] = os.path.relpath(f, repo)
        print(f"{w}: {old} -> {T[w]['bytes_per_launch']} bytes per launch ({n} launches)")
    f = os.path.join(pdir, f"{gen}_plume3d_slab_jacobi_pmc_summary.txt")
    if os.path.exists(f) and "advection_3d_512x512x64" in T:
        A = T["advection_3d_512x512x64"]
        for k in [k for k in A if k.startswith("advect3d_")]:
            del A[k]
        for name, n, fe, wr in rows(f):
            m = re.search(r"(advect3d_\w+_tile_kernel)", name)
            if m:
                A[m.group(1)] = int(round((2 * fe + wr) * MIB))
        A["source"] = os.path.relpath(f, repo)
        print("advection_3d_512x512x64:", {k: v for k, v in A.items() if k.startswith("advect3d_")})
    json.dump(T, open(tfile, "w"), indent=1)


if __name__ == "__main__":
    main()
