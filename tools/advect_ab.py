#!/usr/bin/env python3
"""A/B of the 3D advection pair on a developed plume: fnx_advect_step with its two backward marches (plan 'tiles') against the fused
backward march ('tiles_fused') and one thread per cell ('cells'); stand-alone advect_scalar + advect_vel beside them.
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

    def timed(name, fn, reps=20):
        for _ in range(3):
            r = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            r = fn()
        e1.record(); torch.cuda.synchronize()
        outs[name] = r
        print(f"{name:28s} {e0.elapsed_time(e1) / reps * 1e3:9.1f} us")
    ro, uo = torch.empty_like(rho), torch.empty_like(U)
    for plan in ("tiles", "tiles_fused", "cells"):
        timed(f"advect_step {plan}", lambda: ext.advect_step(m["dt"], rho, U, f, False, 0.6, ro, uo, None, plan))
        outs[f"advect_step {plan}"] = (ro.clone(), uo.clone())
    timed("advect_scalar alone (tiles)", lambda: ext.advect_scalar(m["dt"], rho, U, f, "maccormackFluidNet", 1, False, 0.6, ro))
    timed("advect_vel alone (tiles)", lambda: ext.advect_vel(m["dt"], U, U, f, "maccormackFluidNet", 1, 0.6, uo))
    timed("advect_scalar alone (cells)", lambda: ext.advect_scalar(m["dt"], rho, U, f, "maccormackFluidNet", 1, False, 0.6, ro, None, "cells"))
    timed("advect_vel alone (cells)", lambda: ext.advect_vel(m["dt"], U, U, f, "maccormackFluidNet", 1, 0.6, uo, None, "cells"))
    a = outs["advect_step tiles"]
    for k in ("advect_step tiles_fused", "advect_step cells"):
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
