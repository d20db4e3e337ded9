#!/usr/bin/env python3
"""A/B of the 3D advection pair on a developed plume: fnx_advect_step with its fused backward march (plan 'tiles') against the two
separate backward marches ('tiles_split') and one thread per cell ('cells'); stand-alone advect_scalar + advect_vel beside them.
Same bits required.  python tools/advect_ab.py [res D]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402
from fluidnet_cxx_amd import simulate  # noqa: E402
from fluidnet_cxx_amd._ext import ext  # noqa: E402


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device("cuda:0")
    w = dict(res=res, D=D, method="jacobi", iters=40, kind="plume")
    m = bench.mconf_for(w)
    bd = bench.build_state(w, dev)
    for _ in range(100):
        simulate(m, bd, None, "jacobi")
    torch.cuda.synchronize()
    rho, U, f = bd["density"], bd["U"], bd["flags"]
    print(f"grid {D}x{res}x{res}, max CFL {float(U.abs().max()) * m['dt']:.3f}")
    outs = {}

    ro, uo = torch.empty_like(rho), torch.empty_like(U)
    M = "maccormackFluidNet"
    cases = {
        "advect_step tiles": lambda: ext.advect_step(m["dt"], rho, U, f, False, 0.6, ro, uo, None, "tiles"),
        "advect_step tiles_split": lambda: ext.advect_step(m["dt"], rho, U, f, False, 0.6, ro, uo, None, "tiles_split"),
        "advect_step cells": lambda: ext.advect_step(m["dt"], rho, U, f, False, 0.6, ro, uo, None, "cells"),
        "advect_scalar alone (tiles)": lambda: ext.advect_scalar(m["dt"], rho, U, f, M, 1, False, 0.6, ro),
        "advect_vel alone (tiles)": lambda: ext.advect_vel(m["dt"], U, U, f, M, 1, 0.6, uo),
        "advect_scalar alone (cells)": lambda: ext.advect_scalar(m["dt"], rho, U, f, M, 1, False, 0.6, ro, None, "cells"),
        "advect_vel alone (cells)": lambda: ext.advect_vel(m["dt"], U, U, f, M, 1, 0.6, uo, None, "cells"),
    }
    # the order of measurement matters on a box that has just woken up (the first case timed used to come out 3-5 % slow): a long
    # warm-up of everything, then ROUNDS alternating passes over all cases, the mean and the spread per case
    for fn in cases.values():
        for _ in range(10):
            fn()
    torch.cuda.synchronize()
    ROUNDS, REPS = 5, 20
    times = {k: [] for k in cases}
    for _ in range(ROUNDS):
        for k, fn in cases.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                fn()
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / REPS * 1e3)
    for k, v in times.items():
        print(f"{k:30s} {sum(v) / len(v):8.1f} us   (min {min(v):.1f}, max {max(v):.1f} over {ROUNDS} alternating rounds of {REPS} calls)")
    for k in ("advect_step tiles", "advect_step tiles_split", "advect_step cells"):
        cases[k]()
        outs[k] = (ro.clone(), uo.clone())
    a = outs["advect_step tiles"]
    for k in ("advect_step tiles_split", "advect_step cells"):
        b = outs[k]
        same = torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)) and torch.equal(a[1].view(torch.int32), b[1].view(torch.int32))
        print(f"tiles == {k.split()[-1]}: {same}")
    # per-kernel: HIP-event pairs of the library around the advection class
    ext.profile_enable(True)
    for _ in range(5):
        ext.advect_step(m["dt"], rho, U, f, False, 0.6, ro, uo, None, "tiles")
    torch.cuda.synchronize()
    print("advect class, plan tiles (ms total, launches):", ext.profile_read(2))
    ext.profile_enable(False)


if __name__ == "__main__":
    main()
