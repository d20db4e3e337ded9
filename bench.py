#!/usr/bin/env python3
"""Benchmark of the fluid time step (BASELINE.json metric: steps/s + Mcells/s).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" is one pass of `simulate` over one synthetic plume state that is already resident in HBM.
Workloads (BASELINE.json configs):
  plume2d_1024_cnn     config[1]: 2D plume 1024^2, CNN pressure (ScaleNet, hash-seeded random-init weights)   [default, N=1]
  plume2d_1024_jacobi  2D plume 1024^2, Jacobi-28
  rt2d_2048_jacobi     config[2]: 2D Rayleigh-Taylor 2048^2, Jacobi-100
  plume2d_128_jacobi   config[0]: 2D plume 128^2, Jacobi-28
  plume3d_slab_jacobi  config[4] per-GPU share: 3D plume 512x512x(64 per rank), Jacobi-100  [default, N>1: independent
                       replicas of the slab until the RCCL halo exchange lands -- "scaling": "weak"]
Prints ONE JSON line (rank 0) with the contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TF = 157.3     # exact-f32 MFMA peak

# algorithmic bytes per cell, 2D / 3D
STEP_BYTES = {False: lambda n: 340 + 16 * n, True: lambda n: 452 + 16 * n}
CNN_FLOP_PER_CELL = {False: 484476, True: 1338929}
CNN_GLUE_BYTES = 16 + 8 + 12 + 24 + 24 + 20   # div, std, pack, velUpdate, scale, wallBcs (2D)

WORKLOADS = {
    "plume2d_1024_cnn": dict(res=1024, D=1, method="convnet", iters=0, kind="plume"),
    "plume2d_1024_jacobi": dict(res=1024, D=1, method="jacobi", iters=28, kind="plume"),
    "rt2d_2048_jacobi": dict(res=2048, D=1, method="jacobi", iters=100, kind="rt"),
    "plume2d_128_jacobi": dict(res=128, D=1, method="jacobi", iters=28, kind="plume"),
    "plume3d_256_jacobi": dict(res=256, D=256, method="jacobi", iters=100, kind="plume"),
    "plume3d_slab_jacobi": dict(res=512, D=64, method="jacobi", iters=100, kind="plume"),
}


def build_state(w, dev):
    import numpy as np
    import torch
    from util import plume_state
    res, D = w["res"], w["D"]
    if w["kind"] == "plume":
        st = plume_state(res, D)
        return {k: torch.from_numpy(v).to(dev) for k, v in st.items()}
    # Rayleigh-Taylor (reference init_conditions.py:121-125, rayleighTaylorConfig.yaml)
    from fluidnet_cxx_amd import fluid
    bd = dict(p=torch.zeros(1, 1, 1, res, res, device=dev), U=torch.zeros(1, 2, 1, res, res, device=dev),
              flags=torch.zeros(1, 1, 1, res, res, device=dev), density=torch.zeros(1, 1, 1, res, res, device=dev))
    fluid.emptyDomain(bd["flags"])
    fluid.createRayleighTaylorBCs(bd, dict(perturbThickness=100, perturbAmplitude=0.01, height=0.5), -0.01, 0.01)
    return bd


def mconf_for(w):
    from util import PLUME_CFG
    m = dict(PLUME_CFG)
    if w["kind"] == "rt":
        m.update(dt=0.5, buoyancyScale=1.0, gravityVec=dict(x=0.0, y=1.0, z=0.0))
    m["jacobiIter"] = max(w["iters"], 1)
    m.update(model="ScaleNet", inputChannels=dict(div=True, pDiv=False, UDiv=False), normalizeInput=True,
             normalizeInputChan="UDiv", is3D=w["D"] > 1)
    return m


def cpu_baseline(w, budget_s=20.0):
    """The oracle ("port") timed on the host cores on a bounded sample of the same workload."""
    import numpy as np
    from oracle import oracle as O
    from util import plume_state
    O.build()
    threads = os.cpu_count() or 1
    is3d = w["D"] > 1
    # sample: a smaller grid of the same configuration, scaled per cell
    res = min(w["res"], 256 if not is3d else 64)
    D = 1 if not is3d else min(w["D"], 32)
    st = plume_state(res, D)
    m = mconf_for(w)
    blob = None
    if w["method"] == "convnet":
        from fluidnet_cxx_amd.weights import make_scalenet_weights
        blob = O.pack_weights(make_scalenet_weights(0, ndim=3 if is3d else 2), 3 if is3d else 2)
        res_c = 128
        st = plume_state(res_c, 1); res = res_c
    st = O.simulate_step(st, m, w["method"], blob)      # warm-up
    t0 = time.time(); n = 0
    while True:
        st = O.simulate_step(st, m, w["method"], blob); n += 1
        if time.time() - t0 > budget_s or n >= 50:
            break
    dt = (time.time() - t0) / n
    cells = res * res * D
    return dict(value=cells / dt / 1e6, unit="Mcells/s", cores=threads, kind="port",
                sample=f"{n} steps of the same step on a {D}x{res}x{res} grid ({w['method']}), OpenMP {threads} threads, scaled per cell")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    name = a.workload or ("plume2d_1024_cnn" if a.gpus == 1 else "plume3d_slab_jacobi")
    w = WORKLOADS[name]
    is3d = w["D"] > 1

    from fluidnet_cxx_amd import FluidNet, simulate
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    m = mconf_for(w)
    bd = build_state(w, dev)
    net = FluidNet(m, make_scalenet_weights(0, ndim=3 if is3d else 2), dev) if w["method"] == "convnet" else None
    ws = torch.empty(ext.step_workspace_bytes(1, w["D"], w["res"], w["res"], is3d), dtype=torch.uint8, device=dev)

    def step():
        simulate(m, bd, net, w["method"], workspace=ws)

    for _ in range(a.warmup):
        step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    cells = w["res"] * w["res"] * w["D"]
    ms = elapsed / a.steps * 1e3
    mcells = cells * world * a.steps / elapsed / 1e6

    # ---- dominant kernel, timed with HIP events on the launch stream (torch's current stream) ----
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(3, min(a.steps, 20))
    if w["method"] == "convnet":
        x = torch.randn(1, 2, w["res"], w["res"], device=dev) if not is3d else torch.randn(1, 2, w["D"], w["res"], w["res"], device=dev)
        net.multiScale(x); torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            net.multiScale(x)
        e1.record(); torch.cuda.synchronize()
        kms = e0.elapsed_time(e1) / reps
        flops = CNN_FLOP_PER_CELL[is3d] * cells
        ach = flops / (kms * 1e-3) / 1e12
        roof = dict(bound="mfma", kernel="MultiScaleNet conv stack (17 conv launches)", achieved=ach, peak=MFMA_F32_PEAK_TF,
                    unit="TFLOP/s", frac=ach / MFMA_F32_PEAK_TF, traffic=None, ms_per_launch=kms,
                    algorithmic=f"{CNN_FLOP_PER_CELL[is3d]} FLOP/cell x {cells} cells per forward")
    else:
        from fluidnet_cxx_amd import fluid
        div = fluid.velocityDivergence(bd["U"], bd["flags"])
        fluid.solveLinearSystemJacobi(bd["flags"], div, is3d, 0.0, w["iters"]); torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fluid.solveLinearSystemJacobi(bd["flags"], div, is3d, 0.0, w["iters"])
        e1.record(); torch.cuda.synchronize()
        kms = e0.elapsed_time(e1) / reps
        byts = 16.0 * w["iters"] * cells
        ach = byts / (kms * 1e-3) / 1e9
        roof = dict(bound="hbm", kernel=f"Jacobi solve ({w['iters']} sweeps)", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=ach / HBM_PEAK_GBS, traffic=None, ms_per_launch=kms,
                    algorithmic=f"16 B/cell/sweep x {w['iters']} sweeps x {cells} cells per solve")
    step_bytes = (STEP_BYTES[is3d](w["iters"]) if w["method"] == "jacobi" else STEP_BYTES[is3d](0) - 44 + CNN_GLUE_BYTES) * cells
    out = dict(metric="fluid time-step throughput (steps/s; Mcells/s = cells*steps/s/1e6)", value=mcells, unit="Mcells/s",
               steps_per_s=a.steps * world / elapsed if world == 1 else a.steps / elapsed,
               n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=ms, higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype="f32", data="synthetic",
               config=dict(workload=name, grid=[w["D"], w["res"], w["res"]], cells_per_gpu=cells, method=w["method"],
                           jacobi_iters=w["iters"], parallelism="1 GPU" if world == 1 else f"{world} independent z-slab replicas",
                           weights="hash-seeded random init (pretrained blob absent from the reference)" if net else None),
               step_hbm_frac=step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               roofline=roof)
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
