#!/usr/bin/env python3
"""Benchmark of the fluid time step (BASELINE.json metric: steps/s + Mcells/s, 2D 1024^2 CNN plume & 3D Jacobi, 1-8 GPUs).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-graph] [--no-cpu-baseline] [--no-also]

A "step" is one pass of `simulate` over one synthetic plume state that is already resident in HBM.

Launch.  With WORLD_SIZE in the environment (the driver's `python -m torch.distributed.run ... bench.py --gpus N`) this
process is one rank.  Without it, `--gpus N` (N > 1) re-executes itself under `torch.distributed.run` with N ranks on
127.0.0.1; `--gpus` that disagrees with WORLD_SIZE, or fewer visible devices than ranks, is an error.  `--dry-run`
replaces the GPU work by a gloo rendezvous + all-reduce (proves the launcher on a box without GPUs).

Headline workload.  N = 1: `plume3d_256_jacobi` -- 3D plume 256^3, Jacobi-100, the metric's own 3D configuration.  N > 1:
`plume3d_slab_jacobi` -- 512 x 512 x 64 cells per GPU (as many cells as 256^3), Jacobi-100, one z-slab of configs[4] per rank (at
N = 8 exactly configs[4], 512^3): the SAME per-GPU work at every N (weak scaling); its N = 1 row is `config.weak_n1` of the N = 1 line.
At N = 1 the same JSON line carries the other configurations the metric and the north star name:
  plume3d_slab_jacobi  one z-slab of configs[4] on one GPU (also through the C++ z-slab driver + the middle-rank link model)
  plume2d_1024_cnn     configs[1]: 2D plume 1024^2, CNN pressure (the metric's "2D 1024^2 CNN plume")
  plume2d_1024_jacobi  2D plume 1024^2, Jacobi-28               (north star: >= 60 % HBM roofline on advection+Jacobi at 1024^2)
  rt2d_2048_jacobi     configs[2]: 2D Rayleigh-Taylor 2048^2, Jacobi-100
  plume3d_256_cnn      configs[3]: 3D plume 256^3, CNN pressure (Conv3d analogue of ScaleNet on the MFMA)
  plume3d_hbm_jacobi   3D plume 512 x 512 x 256, Jacobi-100: the solver's working set (1.3 GiB) exceeds the 256 MiB
                       Infinity Cache, so its roofline numbers are against HBM proper
  plume2d_128_jacobi   configs[0]: 2D plume 128^2, Jacobi-28 -- the one configuration the reference itself runs (CPU, plumeConfig.yaml)
  plume2d_128_b32_cnn  the training-shaped call: 32 samples of 128^2 through one `simulate(..., 'convnet')` (the long-term rollout
                       of fluid_net_train.py:349-373 issues such calls under no_grad); also reports samples/s
  plume2d_128_b32_jacobi  configs[0]'s step on 32 samples at once
`config.metric_2d` in the printed line holds the metric's 2D configuration (1024^2 CNN) with its own roofline block and CPU baseline;
`config.dropin` the reference's own call pattern beside the tuned figure, per configuration: [tuned ms, `simulate(mconf, batch_dict,
net, method)` with four arguments and eager launches (plume.py:237), the same step operator by operator (`fused=False`)].
Other names for --workload: plume3d_128_cnn, plume2d_1024_cnn_f2 / plume3d_256_cnn_f2 (F(2x2) Winograd everywhere: the default of rounds 2-5).

State.  Every workload is first advanced by >= 100 untimed steps (`config.developed_steps`) so that a plume exists
(advection cost is data dependent: zero-velocity cells leave the line trace at once); the CNN workloads are developed with
the Jacobi projection (a physical plume) and then timed with the CNN.  W more untimed warm-up steps follow the HIP-graph
capture, then EXACTLY K timed steps between barrier + synchronize pairs, max over ranks.

The slab workload is timed through BOTH z-slab drivers, one after the other on the same state: the Python one
(fluidnet_cxx_amd/slab.py over torch.distributed P2P) and the C++ one (fnx_slab_step: launches and RCCL ncclSend/ncclRecv
issued from C++; `native_driver`, under a watchdog).  They issue the same kernels and exchanges and produce the same bits
(tests/test_slab.py).  At N = 1 the Python-driver time is the headline (one GPU: the step is GPU-bound either way); at N > 1 the
C++ driver is (`config.driver` says so; if its leg fails the Python driver's number stands) and the other is kept beside it
(`python_driver` / `native_driver`), together with `comm`: bytes posted per neighbour and step, the time the compute stream
waited for exchanges, and a send/recv probe of the neighbour links.

Prints ONE JSON line (rank 0) of < 4 kB: the contract fields; `config` (the headline workload, `metric_2d`, `dropin`, `weak_n1`,
`weights`); `roofline`; `cpu_baseline`; `configs` -- one row (ms, Mcells/s, util, of: which utilisation, legend `util_of`) per
BASELINE.json configuration, configs[0] (the reference's own 128^2 CPU case) to configs[4]; `other` -- the remaining workloads;
`kernel_ms_per_step`, `advect`.  Everything else (per-configuration config / roofline / kernel times, prose, PMC detail) goes to the
side file named in `detail_file` (gpurun_out/bench_detail.json).
  roofline.achieved / frac  PHYSICAL: stencil kernels -- HBM-side bytes of one launch (`traffic`) / the launch time measured in this run
                            with HIP events / 8 TB/s; convolutions -- FLOPs issued to the matrix cores / time / 157.3 TF (a Winograd
                            launch issues 16/36 of the direct convolution's FLOPs).  `frac_of` says which
  roofline.traffic          a RECORDED PMC figure (rocprofv3 --pmc passes of an earlier run of the same kernel; `traffic_source`
                            names file and commit), not measured in this run: HIP events and PMC passes cannot share a run
  roofline.achieved_model / frac_model
                            SURVEY 8d's model: algorithmic bytes (16 B per cell and SWEEP) or direct-convolution FLOPs / launch time / peak.
                            The solvers run several sweeps per pass over HBM and the 3x3 layers run in the Winograd domain, so this
                            exceeds 1 by design; it is not a utilisation
  roofline.frac_compulsory  the bytes one launch cannot avoid at its sweeps per pass (3D: p in, div, p out, mask byte = 13 B/cell per
                            two-sweep pass; 2D: 16 B/cell per launch) / launch time / peak
  roofline.frac_out_of_cache  the headline kernel's physical fraction on the 512 x 512 x 256 grid (`plume3d_hbm_jacobi`), whose working
                            set does not fit the 256 MiB Infinity Cache the PMC counters cannot tell from HBM
  kernel_ms_per_step        HIP-event pairs around every launch of a class: ~2 us per launch above the kernels' own time, so the
                            classes can add up to more than ms_per_step
  advect                    the 3D advection launches: ms, fraction of the 120 B/cell model, and the recorded VALU-issue fraction (the
                            kernels are issue-bound: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x busy cycles))
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL P2P fails with hipIpcGetMemHandle otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec; 6.3 TB/s is what a plain copy reaches)
MFMA_F32_PEAK_TF = 157.3     # exact-f32 MFMA peak (v_mfma_f32_32x32x2_f32)
DEVELOP_STEPS = 100          # "warm-up 100 steps so a plume exists"

# algorithmic bytes per cell, 2D / 3D
STEP_BYTES = {False: lambda n: 340 + 16 * n, True: lambda n: 452 + 16 * n}
PROF = dict(jacobi=0, conv_mfma=1, advect=2, stage=3, conv_direct=4, conv_mfma16=5, conv_bf16=6)
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)

# plumeConfig.yaml:29-76 with BASELINE.json's overrides (jacobiIter per workload, pTol 0)
PLUME_CFG = dict(dt=0.1, maccormackStrength=0.6, sampleOutsideFluid=False, buoyancyScale=0.25, gravityScale=0,
                 viscosity=0, correctScalar=False, gravityVec=dict(x=0.0, y=-1.0, z=0.0), operatingDensity=0.0,
                 pTol=0.0, jacobiIter=28, normalizeInputThreshold=1e-5)

WORKLOADS = {
    "plume2d_1024_cnn": dict(res=1024, D=1, method="convnet", iters=0, kind="plume"),
    "plume2d_1024_jacobi": dict(res=1024, D=1, method="jacobi", iters=28, kind="plume"),
    "rt2d_2048_jacobi": dict(res=2048, D=1, method="jacobi", iters=100, kind="rt"),
    "plume2d_128_jacobi": dict(res=128, D=1, method="jacobi", iters=28, kind="plume"),
    "plume3d_256_jacobi": dict(res=256, D=256, method="jacobi", iters=100, kind="plume"),
    "plume3d_hbm_jacobi": dict(res=512, D=256, method="jacobi", iters=100, kind="plume"),
    "plume3d_256_cnn": dict(res=256, D=256, method="convnet", iters=0, kind="plume"),      # configs[3]
    "plume3d_128_cnn": dict(res=128, D=128, method="convnet", iters=0, kind="plume"),
    "plume3d_slab_jacobi": dict(res=512, D=64, method="jacobi", iters=100, kind="plume", slab=True),
    # the training-shaped call (fluid_net_train.py:349-373, trainConfig.yaml batchSize 32/64 at 128^2): the long-term rollout runs
    # `simulate(..., 'convnet')` on the whole batch under no_grad; a step here is one such call on 32 samples
    "plume2d_128_b32_cnn": dict(res=128, D=1, method="convnet", iters=0, kind="plume", batch=32),
    "plume2d_128_b32_jacobi": dict(res=128, D=1, method="jacobi", iters=28, kind="plume", batch=32),   # configs[0]'s step on 32 samples: the lever a 36-us step has
    # round 6: Winograd F(4x4,3x3) for the 64/128-output-channel 3x3(x3) layers is the default; ..._cnn_f2: F(2x2) everywhere (the default of
    # rounds 2-5) for comparison
    "plume2d_1024_cnn_f2": dict(res=1024, D=1, method="convnet", iters=0, kind="plume", precision="fp32_f2"),
    "plume3d_256_cnn_f2": dict(res=256, D=256, method="convnet", iters=0, kind="plume", precision="fp32_f2"),
    # OPT-IN precision mode, never the headline: the 64/128-output-channel Winograd layers as six bf16 MFMA products per fp32
    # product (FNX_PRECISION_BF16X6; same 1e-5 |ref|max tolerance against the oracle as the exact-fp32 modes, tests/)
    "plume2d_1024_cnn_bf16x6": dict(res=1024, D=1, method="convnet", iters=0, kind="plume", precision="bf16x6"),
    "plume3d_256_cnn_bf16x6": dict(res=256, D=256, method="convnet", iters=0, kind="plume", precision="bf16x6"),
    # OPT-IN "accurate bf16" mode (FNX_PRECISION_BF16X3: the three products without a low piece), its own tolerance: 1e-4 |ref|max
    "plume2d_1024_cnn_bf16x3": dict(res=1024, D=1, method="convnet", iters=0, kind="plume", precision="bf16x3"),
    "plume3d_256_cnn_bf16x3": dict(res=256, D=256, method="convnet", iters=0, kind="plume", precision="bf16x3"),
}
ALSO = ["plume3d_slab_jacobi", "plume2d_1024_cnn", "plume2d_128_jacobi", "plume2d_1024_jacobi", "rt2d_2048_jacobi", "plume3d_256_cnn",
        "plume3d_hbm_jacobi", "plume2d_128_b32_cnn", "plume2d_128_b32_jacobi", "plume2d_1024_cnn_bf16x6", "plume3d_256_cnn_bf16x6",
        "plume2d_1024_cnn_bf16x3", "plume3d_256_cnn_bf16x3"]
# the reference's own call pattern timed beside the tuned figure (`dropin`): the metric's two configurations and configs[1..4]
DROPIN = ["plume3d_256_jacobi", "plume2d_1024_cnn", "rt2d_2048_jacobi", "plume3d_256_cnn", "plume3d_slab_jacobi"]
N1_HEADLINE = "plume3d_256_jacobi"   # N = 1: the metric's own 3D configuration; N > 1: plume3d_slab_jacobi (weak scaling, 512^3 at N = 8)
# BASELINE.json's configs[0..4] -> the workload that measures each (configs[4]: one z-slab of it per GPU)
BASELINE_CONFIGS = ["plume2d_128_jacobi", "plume2d_1024_cnn", "rt2d_2048_jacobi", "plume3d_256_cnn", "plume3d_slab_jacobi"]
BF16_MODES = ("bf16x6", "bf16x3")
JOB_DOG = []                 # the N > 1 job's watchdog timer (main)


class Fallback:
    """Rank 0's insurance at N > 1: the supplementary legs (peer-store transport, RCCL inside a HIP graph) have never met a multi-GPU
    box, and a GPU fault in one of them kills the rank before any Python runs -- the job would end without its line although the legs
    that set `value` were through.  Before each such leg rank 0 hands the line AS IT STANDS to ONE small monitor process (own session:
    the launcher's SIGTERM to the workers does not reach it) that reads a pipe of records to its end.  Once started, the monitor is the
    ONLY printer: emit() sends the final line down the pipe (`publish`) and the monitor prints it; if the pipe closes without a
    complete final record -- the rank died, or the launcher killed it because another rank died -- the monitor prints the last line it
    was armed with (marked "fallback").  Records are only taken whole (a rank can die in the middle of a write): one line, whatever the
    moment the rank dies."""
    CODE = ("import sys\nd = sys.stdin.read().split('\\nEOR\\n')\nfb = fin = None\n"
            "for r in d[:-1]:\n"
            "    k, _, v = r.partition('\\n')\n"
            "    if k == 'ARM': fb = v\n"
            "    elif k == 'FINAL': fin = v\n"
            "out = fin if fin is not None else fb\n"
            "if out: sys.stdout.write(out + '\\n'); sys.stdout.flush()\n")

    def __init__(self):
        self.p = None

    def _send(self, kind, line):
        self.p.stdin.write(kind + "\n" + line.replace("\n", " ") + "\nEOR\n")
        self.p.stdin.flush()

    def arm(self, line):
        try:
            if self.p is None:
                self.p = subprocess.Popen([sys.executable, "-c", self.CODE], stdin=subprocess.PIPE, text=True, start_new_session=True)
            self._send("ARM", line)
        except Exception as e:  # noqa: BLE001  (no insurance is not a reason to stop)
            sys.stderr.write(f"bench: fallback monitor not armed ({e})\n")

    def publish(self, line):
        """print the job's line: through the monitor when one runs, directly otherwise"""
        if self.p is None:
            print(line, flush=True)
            return
        try:
            self._send("FINAL", line)
            self.p.stdin.close()
            self.p.wait(timeout=15)
        except Exception:  # noqa: BLE001  (a broken pipe: the monitor is gone -- print here)
            print(line, flush=True)
        self.p = None


FALLBACK = Fallback()
METRIC_CONFIGS = ["plume3d_256_jacobi", "plume2d_1024_cnn"]      # the two configurations BASELINE.json's metric is quoted on


def mfma_flops_per_cell(is3d):
    """Direct-convolution FLOPs of the layers that run on the matrix cores (3x3 convs with 32/64/128 channels), per
    full-resolution cell."""
    taps = 27 if is3d else 9
    t1 = 2 * taps * (32 * 64 + 64 * 128 + 128 * 64 + 64 * 32)
    t4 = 2 * taps * (32 * 64 + 64 * 32)
    s = 8 if is3d else 4
    return t1 * (1 + 1.0 / s) + t4 / (s * s)


def plume_state_torch(res, D_local, dev, z_offset=0, D_global=None):
    """Plume initial state + BC masks (reference plume.py:131-163 / createPlumeBCs; 3D: inlet disc) for the planes
    [z_offset, z_offset + D_local) of a D_global-deep domain, built on `dev`."""
    import torch
    is3d = (D_global or D_local) > 1
    Dg = D_global or D_local
    nc = 3 if is3d else 2
    shp = (1, 1, D_local, res, res)
    z = torch.arange(z_offset, z_offset + D_local, device=dev).view(D_local, 1, 1)
    y = torch.arange(res, device=dev).view(1, res, 1)
    x = torch.arange(res, device=dev).view(1, 1, res)
    border = (x < 1) | (x > res - 2) | (y < 1) | (y > res - 2)
    if is3d:
        border = border | (z < 1) | (z > Dg - 2)
    flags = torch.where(border, 2.0, 1.0).to(torch.float32).expand(D_local, res, res).reshape(shp).contiguous()
    rad = math.floor(res * 0.145)
    r2 = (x - res // 2) ** 2
    if is3d:
        r2 = r2 + (z - Dg // 2) ** 2
    inside = (r2 <= rad * rad).expand(D_local, 4, res)
    st = dict(p=torch.zeros(shp, device=dev), U=torch.zeros((1, nc, D_local, res, res), device=dev),
              density=torch.zeros(shp, device=dev), flags=flags)
    UBC = torch.zeros_like(st["U"]); UBC[0, 1, :, 0:4] = inside.float() * 2.0
    UBCInvMask = torch.ones_like(st["U"]); UBCInvMask[:, :, :, 0:4] = 0
    dBC = torch.zeros(shp, device=dev); dBC[0, 0, :, 0:4] = inside.float() * 0.1
    dMask = torch.ones(shp, device=dev); dMask[0, 0, :, 0:4] = (~inside).float()
    st.update(UBC=UBC, UBCInvMask=UBCInvMask, densityBC=dBC, densityBCInvMask=dMask)
    return st


def build_state(w, dev):
    import torch
    res, D = w["res"], w["D"]
    if w["kind"] == "plume":
        st = plume_state_torch(res, D, dev)
        if w.get("batch", 1) > 1:
            st = {k: v.repeat(w["batch"], 1, 1, 1, 1).contiguous() for k, v in st.items()}
        return st
    # Rayleigh-Taylor (reference init_conditions.py:121-125, rayleighTaylorConfig.yaml)
    from fluidnet_cxx_amd import fluid
    bd = dict(p=torch.zeros(1, 1, 1, res, res, device=dev), U=torch.zeros(1, 2, 1, res, res, device=dev),
              flags=torch.zeros(1, 1, 1, res, res, device=dev), density=torch.zeros(1, 1, 1, res, res, device=dev))
    fluid.emptyDomain(bd["flags"])
    fluid.createRayleighTaylorBCs(bd, dict(perturbThickness=100, perturbAmplitude=0.01, height=0.5), -0.01, 0.01)
    return bd


def mconf_for(w):
    m = dict(PLUME_CFG)
    if w["kind"] == "rt":
        m.update(dt=0.5, buoyancyScale=1.0, gravityVec=dict(x=0.0, y=1.0, z=0.0))
    m["jacobiIter"] = max(w["iters"], 1)
    m.update(model="ScaleNet", inputChannels=dict(div=True, pDiv=False, UDiv=False), normalizeInput=True,
             normalizeInputChan="UDiv", is3D=w["D"] > 1)
    if w.get("precision"):
        m["precisionMode"] = w["precision"]
    return m


def cpu_baseline(w, budget_s=12.0):
    """The oracle ("port": plain-C restatement of the reference, OpenMP) timed on this box's host cores on a bounded
    sample of the same workload: the same step on a smaller grid of the same configuration, scaled per cell.  The only
    place bench.py touches oracle/ -- as the thing compared WITH, outside every timed GPU region."""
    import torch
    from oracle import oracle as O
    O.build()
    threads = min(os.cpu_count() or 1, 64)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    is3d = w["D"] > 1
    m = mconf_for(w)
    blob = None
    if w["method"] == "convnet":
        from fluidnet_cxx_amd.weights import make_scalenet_weights
        blob = O.pack_weights(make_scalenet_weights(0, ndim=3 if is3d else 2), 3 if is3d else 2)
        res, D = 128, 1
    else:
        res, D = (min(w["res"], 512), 1) if not is3d else (128, 64)
    st = {k: v.numpy() for k, v in plume_state_torch(res, D, torch.device("cpu")).items()}
    st = O.simulate_step(st, m, w["method"], blob)      # warm-up
    t0 = time.time(); n = 0
    while True:
        st = O.simulate_step(st, m, w["method"], blob); n += 1
        if time.time() - t0 > budget_s or n >= 400:
            break
    dt = (time.time() - t0) / n
    cells = res * res * D
    return dict(value=cells / dt / 1e6, unit="Mcells/s", cores=threads, kind="port",
                sample=f"{n} {w['method']} steps on a {D}x{res}x{res} grid, per-cell rate")


def jacobi_replay_launch_ms(ext, bd, is3d, iters, launches_per_solve):
    """ms per launch of the solver's pass with its launches back to back in a replayed graph (see the call)"""
    import torch
    flags = bd["flags"]
    div = ext.velocity_divergence(bd["U"], flags, None)

    def solve_ms(n_iter, reps=10):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            keep = ext.solve_linear_system(flags, div, is3d, 0.0, n_iter, False, None)   # noqa: F841  (outputs live in the graph's pool)
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):                                   # (the least disturbed of three timings of `reps` replays)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / reps
            best = t if best is None or t < best else best
        return best
    return (solve_ms(2 * iters) - solve_ms(iters)) / launches_per_solve


def jacobi_roofline(name, w, cells, prof_steps, jac, traffic, traffic_src, traffic_detail, replay_ms=None):
    """roofline block of the solver's pass kernel.  `jac` = (ms, launches) of the HIP-event pairs around every launch over `prof_steps`
    eager steps of this rank's `cells`; `replay_ms`: ms per launch with the launches back to back in a replayed graph -- the figure the
    fractions use when it is there (how the kernel runs in the timed step; the pairs' figure stays beside it as avg_launch_ms_each)"""
    is3d = w["D"] > 1
    tms, nl = jac
    each_ms = tms / max(nl, 1)
    if replay_ms and replay_ms > 0:
        tms = replay_ms * nl
    byts = 16.0 * w["iters"] * cells * prof_steps
    ach = byts / (tms * 1e-3) / 1e9 if tms > 0 else 0.0
    kname = ("jacobi3d_march2_kernel<false,false,3> (the steady-state instantiation: z-marching, TWO sweeps per pass, p handed from "
             "pass to pass in the row-quad layout; the first pass of a solve is <true,..>, the last one writes rows)" if is3d
             else "jacobi2d_wg_kernel<8,8> (register/DPP temporal blocking, 64x64-cell workgroup tiles, 7-10 sweeps per launch)")
    avg_ms = tms / max(nl, 1)
    # what one launch MUST move at its sweeps per pass: 3D, two sweeps per pass: p in + div + p out (4 B each) + the mask byte;
    # 2D, 7-10 sweeps per launch: p in + div + flags + p out.  frac_model (SURVEY 8d's 16 B per cell and SWEEP) exceeds 1 by design of
    # the temporal blocking; frac_compulsory is the fraction of the HBM peak the kernel needs for the bytes it cannot avoid.
    comp = (13.0 if is3d else 16.0) * cells
    return dict(bound="hbm", kernel=kname, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                compulsory_bytes_per_launch=comp,
                frac_compulsory=(comp / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if avg_ms > 0 else None,
                traffic=traffic, traffic_source=traffic_src,
                frac_traffic=(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and avg_ms > 0) else None,
                traffic_detail=traffic_detail, launches_per_step=nl / prof_steps,
                avg_launch_ms=avg_ms, avg_launch_ms_each=each_ms,
                timing=("HIP events around 10 replays of a graph holding one solve, (T(2 x iters) - T(iters)) / launches; _each: an event pair around "
                        "every eager launch") if (replay_ms and replay_ms > 0) else "an event pair around every eager launch",
                algorithmic=f"16 B/cell/sweep x {w['iters']} sweeps x {cells} owned cells per step")


def recorded_traffic(name):
    """(bytes per launch, source, detail) of the dominant kernel of workload `name` from the committed PMC table"""
    tfile = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if not os.path.exists(tfile):
        return None, None, None
    d = json.load(open(tfile)).get(name)
    return (d.get("bytes_per_launch") if d else None), (d or {}).get("source"), d


def run_workload(name, steps, warmup, use_graph, world, rank, dev, schedule="deep_first", dropin=False):
    """Develop the state, warm up, time `steps` steps (barrier + synchronize on both sides, max over ranks), then profile
    the dominant kernel class with HIP events.  Returns the JSON-able result dict (without cpu_baseline)."""
    import torch
    import torch.distributed as dist
    from fluidnet_cxx_amd import FluidNet, simulate
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    w = WORKLOADS[name]
    is3d = w["D"] > 1
    slab = bool(w.get("slab"))
    m = mconf_for(w)
    graph_used = False
    layout = None
    develop = max(DEVELOP_STEPS, warmup)
    if slab:
        # weak scaling: every GPU owns 64 planes of a 512 x 512 x (64*world) plume
        from fluidnet_cxx_amd.slab import SlabLayout, SlabSimulator
        layout = SlabLayout(w["D"] * world, world, rank, halo=6)
        bd = plume_state_torch(w["res"], layout.D_local, dev, layout.z_offset, layout.D_global)
        # 6 sweeps per exchange = halo: 17 ghost exchanges of p per 100 sweeps; the plume's flags never change
        sim = SlabSimulator(layout, m, sweeps_per_exchange=6, schedule=schedule, static_flags=True)
        net = None

        def eager_step(method=None):
            sim.step(bd)
        cells = w["res"] * w["res"] * layout.owned
        static_desc = "flags + BC arrays promised static from step 2 on (solver obstacle mask and BC class map reused)"
    else:
        bd = build_state(w, dev)
        net = FluidNet.from_weights(m, make_scalenet_weights(0, ndim=3 if is3d else 2), dev) if w["method"] == "convnet" else None
        ws = torch.empty(ext.step_workspace_bytes(w.get("batch", 1), w["D"], w["res"], w["res"], is3d), dtype=torch.uint8, device=dev)
        seen = []

        def eager_step(method=None):
            # flags and BC arrays never change here: static_flags 0 for the first step, then 3 (solver keeps its mask, the
            # BC stages build their class map), then 7 (and reuse it)
            meth = method or w["method"]
            simulate(m, bd, net if meth == "convnet" else None, meth, workspace=ws, static_flags=(0, 3, 7)[min(len(seen), 2)])
            seen.append(1)
        cells = w["res"] * w["res"] * w["D"] * w.get("batch", 1)
        static_desc = "static_flags 0, 3, then 7: flags + BC arrays promised static (solver obstacle mask and BC class map reused)"
    step = eager_step

    # ---- develop the state (untimed): a physical plume via the Jacobi projection, whatever method is timed afterwards
    timed_iters = m["jacobiIter"]
    if w["method"] == "convnet":
        m["jacobiIter"] = 28 if not is3d else 40
    for _ in range(develop):
        eager_step(None if slab else "jacobi")
    m["jacobiIter"] = timed_iters
    torch.cuda.synchronize()
    umax = float(bd["U"].abs().max()) * float(m["dt"])
    if not slab and use_graph:
        # the step is a fixed launch sequence on fixed buffers: capture it once, replay it per step
        try:
            eager_step()                       # (first step of the timed method outside the capture: lazy packing etc.)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                eager_step()
            g.replay(); torch.cuda.synchronize()
            step = g.replay
            graph_used = True
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"bench: graph capture failed ({e}); running eagerly\n")
            step = eager_step
    for _ in range(warmup):
        step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = elapsed / steps * 1e3
    mcells = cells * world * steps / elapsed / 1e6
    finite = bool(torch.isfinite(bd["U"]).all()) and bool(torch.isfinite(bd["p"]).all())

    # ---- the reference's own call pattern on the same state (single GPU): `simulate(mconf, batch_dict, net, method)` with four
    # arguments, eager -- plume.py:237; workspace and static inputs are the layer's business (_simulate.py) -- and the same step
    # operator by operator (`fused=False`: advectScalar / advectVelocity / ... one call each, cpp/advection.py:64,115)
    drop = None
    if dropin and world == 1:
        from fluidnet_cxx_amd import _simulate
        mm, nn = w["method"], (net if w["method"] == "convnet" else None)

        def loop(fn, n):
            for _ in range(3):
                fn()                             # (three calls: static inputs detected, class map built, steady state)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / n * 1e3
        try:
            four = loop(lambda: simulate(m, bd, nn, mm), steps)
            ops = loop(lambda: simulate(m, bd, nn, mm, fused=False), steps)
            drop = dict(tuned_ms=ms, four_arg_ms=four, operators_ms=ops, four_arg_over_tuned=four / ms, operators_over_tuned=ops / ms,
                        state_finite=bool(torch.isfinite(bd["U"]).all()) and bool(torch.isfinite(bd["p"]).all()))
        except Exception as e:  # noqa: BLE001
            drop = dict(error=f"{type(e).__name__}: {e}"[:200])
        _simulate.release_workspaces()

    # ---- dominant kernel: HIP events around every launch of its class, on the launch stream, over more steps of the
    # same workload (eager launches: events cannot be recorded inside a captured graph) ----
    ext.profile_enable(True)
    prof_steps = max(2, min(steps, 10))
    for _ in range(prof_steps):
        eager_step()
    torch.cuda.synchronize()
    times = {k: ext.profile_read(v) for k, v in PROF.items()}
    issued = {k: ext.profile_read_work(v) for k, v in PROF.items()}
    ext.profile_enable(False)
    # The solver's pass as it runs in the timed step -- launches back to back in a replayed graph: HIP events around ten replays of a graph
    # that holds ONE solve, at `iters` and at 2 x `iters` sweeps; the difference is the time of the extra launches alone (no mask build,
    # no first pass).  A pair around every eager launch (above) makes each kernel wait for the one before it to drain and reads ~2 us
    # long; rocprofv3's average over the same command agrees with the replay figure (profiles/r06/a_*_kernel_stats.csv).
    replay_ms = None
    if w["method"] == "jacobi" and world == 1 and times["jacobi"][1] > 0:
        try:
            replay_ms = jacobi_replay_launch_ms(ext, bd, is3d, w["iters"], times["jacobi"][1] / prof_steps)
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"bench: no replayed-solve timing ({type(e).__name__}: {e})\n")
    # HBM bytes per launch of the roofline kernel (PMC FETCH_SIZE/WRITE_SIZE passes, profiles/)
    traffic, traffic_src, traffic_detail = recorded_traffic(name)
    tfile = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if w["method"] == "convnet":
        tms, nl = times["conv_mfma"]
        if w.get("precision") in BF16_MODES:      # achieved/frac: all the 3x3(x3) MFMA layers, whichever kernel ran them
            tms, nl = tms + times["conv_bf16"][0], nl + times["conv_bf16"][1]
        flops = mfma_flops_per_cell(is3d) * cells * prof_steps
        ach = flops / (tms * 1e-3) / 1e12 if tms > 0 else 0.0
        util = issued["conv_mfma"] / (tms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF if tms > 0 else 0.0
        if w.get("precision") in BF16_MODES:
            util = None                          # (two instruction kinds in one figure would mean nothing: see roofline.bf16x6)
        kname = ("conv3_wino3_kernel<2,2," + ("true" if is3d else "false") + "> (+ <1,2,.> for the 32-channel outputs; persistent software pipeline; 3x3" + ("x3" if is3d else "") + " conv in the Winograd F(2x2,3x3) domain" +
                 (" in x,y, the three z taps in the contraction" if is3d else "") + ": 16 multiplies per 4 outputs "
                 "instead of 36, v_mfma_f32_32x32x2_f32); achieved/frac count DIRECT-convolution FLOPs and can exceed the "
                 "MFMA peak, mfma_util counts the FLOPs actually issued to the matrix cores")
        if w.get("precision", "fp32") in ("fp32", "fp32_f4"):
            t3 = "true" if is3d else "false"
            kname = (f"conv3_wino4_kernel<{t3}> (round 6: the 64/128-output-channel 3x3{'x3' if is3d else ''} layers whose launch fills the chip -- 6 of the 10 MFMA launches of a "
                     "forward, 0.9 of their time -- in the Winograd F(4x4,3x3) domain" + (" in (y, x), the three z taps as stages" if is3d else "") + ": 36 multiplies per 16 "
                     "outputs instead of 144, v_mfma_f32_16x16x4_f32, one wave = 16 output channels x 16 blocks x all 36 positions) + "
                     f"conv3_wino3_kernel<1,2,{t3}> / <2,2,{t3}> (F(2x2,3x3), v_mfma_f32_32x32x2_f32) for the 32-channel outputs and the quarter-resolution layers; "
                     "achieved/frac count DIRECT-convolution FLOPs and can exceed the MFMA peak, mfma_util counts the FLOPs actually issued to the matrix cores "
                     "(F(4x4) issues 0.5625 of F(2x2)'s)")
        avg_ms = tms / max(nl, 1)
        if w.get("precision") in BF16_MODES:
            # the opt-in mode: its own kernel, priced against the bf16 MFMA peak on the bf16 FLOPs it issues (six per fp32 product)
            tb, nb = times["conv_bf16"]
            ub = issued["conv_bf16"] / (tb * 1e-3) / 1e12 if tb > 0 else 0.0
            roof_bf16 = dict(kernel="conv3_wbf_kernel<" + ("true" if is3d else "false") + (",6" if w["precision"] == "bf16x6" else ",3") + "> (FNX_PRECISION_" + w["precision"].upper() + ": Winograd-domain GEMMs as six / three "
                             "v_mfma_f32_32x32x16_bf16 per fp32 product; the 32-output-channel layers stay on conv3_wino3_kernel)",
                             achieved=ub, peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s (bf16 issued)", frac=ub / MFMA_BF16_PEAK_TF,
                             launches_per_step=nb / prof_steps, ms_per_step=tb / prof_steps)
        else:
            roof_bf16 = None
        roof = dict(bound="mfma", kernel=kname, achieved=ach,
                    peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=ach / MFMA_F32_PEAK_TF, mfma_util=util, bf16x6=roof_bf16,
                    achieved_issued=(util * MFMA_F32_PEAK_TF) if util is not None else None,
                    issued_tflop_per_step=issued["conv_mfma"] / prof_steps / 1e12, traffic=traffic, traffic_source=traffic_src,
                    frac_traffic=(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and avg_ms > 0) else None,
                    traffic_detail=traffic_detail,
                    launches_per_step=nl / prof_steps, avg_launch_ms=avg_ms,
                    algorithmic=f"{mfma_flops_per_cell(is3d):.0f} direct-convolution FLOP/cell in the MFMA conv launches x {cells} cells per step")
    else:
        roof = jacobi_roofline(name, w, cells, prof_steps, times["jacobi"], traffic, traffic_src, traffic_detail, replay_ms)
    if w["method"] == "jacobi":
        step_bytes = STEP_BYTES[is3d](w["iters"]) * cells
    else:
        step_bytes = (STEP_BYTES[is3d](0) - (44 if not is3d else 60) + 104) * cells   # advection + CNN glue
    advect = None
    if is3d and times.get("advect", (0, 0))[1] > 0:
        adv_ms = times["advect"][0] / prof_steps
        rec = (json.load(open(tfile)) if os.path.exists(tfile) else {}).get("advection_3d_512x512x64", {})
        advect = dict(ms_per_step=adv_ms, model_bytes_per_cell=120,
                      frac_of_model=(120.0 * cells / (adv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if adv_ms > 0 else None,
                      valu_issue_frac=rec.get("valu_issue_frac"), valu_source=rec.get("valu_source"),
                      note="issue-bound (~1 190 VALU instructions per cell); valu_issue_frac = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE), recorded PMC pass")
    if slab:
        run_workload.slab_state = (bd, m)
    return dict(advect=advect, metric="fluid time-step throughput, Mcells/s = cells*steps/s/1e6 (steps/s alongside)", value=mcells,
                unit="Mcells/s", steps_per_s=steps / elapsed, n_gpus=world, steps=steps, warmup=warmup, ms_per_step=ms,
                higher_is_better=True, scaling="weak", vs_baseline=None,
                dtype=f"f32 ({w['precision']} products: opt-in mode)" if w.get("precision") in BF16_MODES else "f32", data="synthetic",
                config=dict(workload=name, grid_per_gpu=[layout.owned if slab else w["D"], w["res"], w["res"]],
                            global_grid=[layout.D_global if slab else w["D"], w["res"], w["res"]], batch=w.get("batch", 1),
                            cells_per_gpu=cells, method=w["method"], jacobi_iters=w["iters"], precision=w.get("precision", "fp32"),
                            parallelism=("1 GPU" if world == 1 else
                                         f"{world} z-slabs, neighbour P2P ghost exchange (RCCL send/recv), halo 6, 6 sweeps per exchange, "
                                         f"schedule {schedule}"),
                            launch="hip-graph replay" if graph_used else "eager",
                            developed_steps=develop,
                            developed_with=("jacobi projection (physical plume), then timed with the CNN" if w["method"] == "convnet"
                                            else "the timed step itself"),
                            max_cfl_after_development=umax,
                            static_flags=static_desc,
                            state_finite_after_timing=finite,
                            weights="seeded (pretrained blob absent from the reference)" if net else None),
                samples_per_s=(w["batch"] * steps / elapsed) if w.get("batch", 1) > 1 else None,
                step_hbm_frac=step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                kernel_ms_per_step={k: v[0] / prof_steps for k, v in times.items() if v[1] > 0},
                roofline=roof, dropin=drop)


def run_native_slab(steps, warmup, world, rank, dev, bd, m, res, D, schedule="deep_first", transport="rccl", capture=None):
    """The SAME per-GPU slab step through the C++ z-slab driver (fnx_slab_step: launches, RCCL ncclSend/ncclRecv and their
    overlap issued from C++, csrc/fnx_slab.hip), continuing from the state the Python-driver run developed; same bits as the
    Python driver (tests/test_slab.py).  At N > 1 it also returns `comm`: a one-off probe of the communicator (ghost exchanges
    of 6 MiB and of 4 KiB with each neighbour: bandwidth and latency of the link as this job sees it), and from a few extra
    steps with the driver's statistics on, the bytes posted per neighbour and step, the number of exchanges, and the time the
    compute stream stood waiting for posted exchanges (HIP events around each stream wait)."""
    import torch
    import torch.distributed as dist
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout, peer_comm, rccl_comm
    layout = SlabLayout(D * world, world, rank, halo=6)
    # transport "peer": device stores into hipIpc-mapped mailboxes + flags (csrc/fnx_peer.hip) instead of RCCL send/recv
    comm = (peer_comm(rank, world, 16 << 20) if transport == "peer" else rccl_comm(rank, world)) if world > 1 else None
    sim = NativeSlabSimulator(layout, m, comm=comm, sweeps_per_exchange=6, static_flags=True, cfl_check_every=0, schedule=schedule)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    comm_info = None
    if world > 1:
        big, small, reps = 6 << 20, 4096, 20
        scratch = torch.zeros(4 * big, dtype=torch.uint8, device=dev)
        ms_big = ext.slab_comm_probe(comm, big, reps, scratch)
        ms_small = ext.slab_comm_probe(comm, small, reps, scratch)
        t = torch.tensor([ms_big, ms_small], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_big, ms_small = (float(x) for x in t.tolist())
        comm_info = dict(probe_6MiB_ms=ms_big, probe_6MiB_GBps_per_direction=(big / (ms_big * 1e-3) / 1e9) if ms_big > 0 else None,
                         probe_4KiB_us=ms_small * 1e3,
                         probe="max over ranks of the mean of 20 ghost exchanges with each z-neighbour, back to back (" +
                               ("peer-store launches" if transport == "peer" else "grouped ncclSend/ncclRecv") + ")")
        del scratch
    for _ in range(max(warmup, 3) + (DEVELOP_STEPS if getattr(run_workload, "fresh_state", False) else 0)):
        sim.step(bd)
    run_workload.fresh_state = False
    step, launch = (lambda: sim.step(bd)), "eager (C++ driver)"
    eager_ms = None
    if world == 1:
        # one rank: eager first (kept as `eager_ms`), then the replayed graph below -- the step is GPU-bound either way, and a replayed
        # node carries its dependency barrier where same-stream eager launches need none
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t1) / steps * 1e3
    if capture is None:
        capture = world == 1 or transport == "peer"
    if capture:
        # one rank: the step is a fixed launch sequence on fixed buffers -- capture it once, replay it per step.  The peer-store
        # transport too: its chunk counters live on the device, nothing in a captured step depends on how many exchanges came before
        # (tests/test_peer.py::test_peer_store_step_replays_as_hip_graph).  RCCL calls are eager in the leg that sets `value`; a
        # further leg (`native_driver_rccl_graph`, last, under its own watchdog) tries the same capture around RCCL's send / recv
        try:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                sim.step(bd)
            g.replay(); torch.cuda.synchronize()
            step, launch = g.replay, "hip-graph replay"
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"bench: native slab step not captured ({e}); running eagerly\n")
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # a few more steps with the driver's statistics on (event pairs around each wait: outside the timed region)
        n_stat = 5
        sim._drv.stats_enable(True)
        for _ in range(n_stat):
            sim.step(bd)
        torch.cuda.synchronize()
        st = sim._drv.stats_read()
        sim._drv.stats_enable(False)
        t = torch.tensor([st["wait_ms"] / n_stat, st["bytes_per_neighbour"] / n_stat], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        comm_info.update(wait_ms_per_step=float(t[0]), bytes_per_neighbour_per_step=float(t[1]), exchanges_per_step=st["exchanges"] / n_stat,
                         wait="max over ranks; time the compute stream stood in front of posted ghost exchanges, HIP events, 5 untimed steps")
    # the dominant kernel through THIS driver: HIP-event pairs around the solver's launches over a few eager steps (outside the timed
    # region: events cannot be recorded inside a replayed graph).  Partial edge / deep launches of the sweep blocks are in the mean.
    roof = None
    try:
        ext.profile_enable(True)
        n_prof = 3
        for _ in range(n_prof):
            sim.step(bd)
        torch.cuda.synchronize()
        jac = ext.profile_read(PROF["jacobi"])
        ext.profile_enable(False)
        tr, tsrc, tdet = recorded_traffic("plume3d_slab_jacobi")
        roof = jacobi_roofline("plume3d_slab_jacobi", WORKLOADS["plume3d_slab_jacobi"], res * res * layout.owned, n_prof, jac,
                               tr if world == 1 else None, tsrc if world == 1 else None, tdet if world == 1 else None)
        if world > 1:
            roof["note"] = "mean over the deep and the (shorter) edge launches of the sweep blocks of this rank"
    except Exception as e:  # noqa: BLE001
        roof = None
        sys.stderr.write(f"bench: no roofline from the native leg ({e})\n")
    model = None
    if world == 1 and hasattr(ext, "slab_comm_link_model"):
        # One GPU rehearses a MIDDLE rank (rank 1 of 3: 64 owned + 2 x 6 ghost planes) through the same C++ driver with the C ABI's
        # link-model communicator: every ghost exchange occupies the communication stream for latency + bytes / bandwidth and
        # fills the ghost planes from the slab's own edge planes.  A model of N >= 3, not a measurement of it.
        try:
            lm = SlabLayout(D * 3, 3, 1, 6)
            stm = plume_state_torch(res, lm.D_local, dev, lm.z_offset, lm.D_global)
            model = dict(what="middle rank of 3 z-slabs on ONE GPU, C++ driver, link-model communicator (a model of N >= 3, not a measurement)",
                         ghost_free_ms=elapsed / steps * 1e3)
            # (latency 9 us: the peer-store launch between two processes, tools/peer_probe.py; 20-25 us: a grouped RCCL send/recv.
            #  deep_beside takes the communicator's direct sends: the last edge part of a block stores into the neighbour's mailbox itself)
            def model_ms(sched, lat, gbps):
                simm = NativeSlabSimulator(lm, m, comm=ext.slab_comm_link_model(lat, gbps), sweeps_per_exchange=6, static_flags=True,
                                           cfl_check_every=0, schedule=sched)
                for _ in range(5):
                    simm.step(stm)
                stepm = lambda: simm.step(stm)      # noqa: E731
                try:                                 # the step is a fixed launch sequence: replay it as a HIP graph, as at N = 1
                    torch.cuda.synchronize()
                    gm = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gm):
                        simm.step(stm)
                    gm.replay(); torch.cuda.synchronize()
                    stepm = gm.replay
                except Exception:  # noqa: BLE001
                    pass
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(10):
                    stepm()
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / 10 * 1e3
            for sched, lat, gbps in ((schedule, 9.0, 75.0), (schedule, 9.0, 55.0), (schedule, 20.0, 75.0), (schedule, 25.0, 55.0),
                                     ("deep_beside", 9.0, 75.0), ("deep_beside", 9.0, 55.0)):
                msm = model_ms(sched, lat, gbps)
                tag = ("beside_" if sched == "deep_beside" and sched != schedule else "") + f"{int(gbps)}GBps_{int(lat)}us"
                model[f"ms_at_{tag}"] = msm
                model[f"modelled_efficiency_at_{tag}"] = (elapsed / steps * 1e3) / msm
        except Exception as e:  # noqa: BLE001
            model = dict(error=f"{type(e).__name__}: {e}")
    cells = res * res * layout.owned * world
    torch.cuda.synchronize()
    if comm is not None and comm.failed():
        # (peer-store exchanges are enqueued, not waited for: a replayed step that met a dead neighbour returns normally -- its time is not a result)
        raise RuntimeError("the peer-store communicator reports a timed-out / aborted exchange during the timed steps")
    out = dict(ms_per_step=elapsed / steps * 1e3, value=cells * steps / elapsed / 1e6, unit="Mcells/s", steps=steps, launch=launch,
               eager_ms_per_step=eager_ms, schedule=schedule,
               transport=(("peer-store: device stores into hipIpc-mapped mailboxes + flags (csrc/fnx_peer.hip), no RCCL on the data path"
                           if transport == "peer" else "RCCL ncclSend/ncclRecv issued from C++ (librccl resolved at run time)") if world > 1 else None),
               state_finite=bool(torch.isfinite(bd["U"]).all()) and bool(torch.isfinite(bd["p"]).all()))
    if model:
        out["middle_rank_model"] = model
    if roof:
        out["roofline"] = roof
    return out, comm_info


def _r(x, n=4):
    """round to n significant digits (the line is for reading; the side file keeps full precision)"""
    if isinstance(x, float) and x == x and x not in (float("inf"), float("-inf")) and x != 0.0:
        return float(f"{x:.{n}g}")
    return x


def _short(res):
    """one configuration's row of `configs` / `other`: [ms per step, Mcells/s, fraction, which fraction]"""
    rf = res.get("roofline", {})
    if rf.get("bound") == "mfma":
        if rf.get("bf16x6"):
            frac, what = rf["bf16x6"]["frac"], "B"
        else:
            frac, what = rf.get("mfma_util"), "M"
    else:
        frac, what = rf.get("frac_traffic"), "P"
        if frac is None:
            frac, what = rf.get("frac_compulsory"), "C"
    e = dict(ms=_r(res["ms_per_step"], 5), Mcells_s=_r(res["value"], 5), util=_r(frac, 3), of=what)
    if res.get("samples_per_s"):
        e["samples_s"] = _r(res["samples_per_s"], 5)
    return e


def pmc_source(summary_file=None):
    """where roofline.traffic comes from: the rocprofv3 PMC summary it was taken from (recorded in the committed PMC table) and the
    commit that last changed the table"""
    src = "profiles/pmc_traffic.json"
    if summary_file:
        return f"recorded PMC passes: {summary_file}"
    try:
        c = subprocess.run(["git", "log", "-1", "--format=%h", "--", src], cwd=REPO, capture_output=True, text=True, timeout=5).stdout.strip()
        return f"recorded rocprofv3 PMC passes, {src}" + (f" @ {c}" if c else "")
    except Exception:  # noqa: BLE001  (no git on the box)
        return f"recorded rocprofv3 PMC passes, {src}"


def _roof(rf, long=False):
    """a roofline block of the printed line.  `achieved` / `frac` are the PHYSICAL figures: stencils -- HBM-side bytes of the launch
    (recorded rocprofv3 PMC passes; the bytes a launch cannot avoid where no pass was recorded: `frac_of`) / the launch time measured in
    this run / 8 TB/s; convolutions -- FLOPs issued to the matrix cores / time / the fp32 MFMA peak.  `achieved_model` / `frac_model` are
    SURVEY 8d's figures (16 B per cell and SWEEP; direct-convolution FLOPs): they exceed 1 where a pass does several sweeps / the layer
    runs in the Winograd domain."""
    e = dict(bound=rf["bound"], kernel=rf["kernel"].split(" (")[0], peak=rf["peak"], unit=rf["unit"])
    if rf["bound"] == "mfma":
        if rf.get("mfma_util") is not None:
            e.update(achieved=_r(rf.get("achieved_issued"), 5), frac=_r(rf["mfma_util"]), frac_of="issued FLOPs")
        else:                                  # (opt-in bf16 modes: two instruction kinds -- the mode's own block says what it reaches)
            e.update(achieved=_r(rf["achieved"], 5), frac=_r(rf["frac"]), frac_of="direct-convolution FLOPs (model)")
        e.update(achieved_model=_r(rf["achieved"], 5), frac_model=_r(rf["frac"]), traffic=rf.get("traffic"), frac_traffic=_r(rf.get("frac_traffic")))
    else:
        if rf.get("frac_traffic") is not None:
            e.update(achieved=_r(rf["frac_traffic"] * rf["peak"], 5), frac=_r(rf["frac_traffic"]), frac_of="PMC bytes")
        else:
            e.update(achieved=_r((rf.get("frac_compulsory") or 0.0) * rf["peak"], 5), frac=_r(rf.get("frac_compulsory")), frac_of="compulsory bytes (no PMC pass recorded)")
        e.update(achieved_model=_r(rf["achieved"], 5), frac_model=_r(rf["frac"]), traffic=rf.get("traffic"), frac_compulsory=_r(rf.get("frac_compulsory")))
    if long:
        e.update(traffic_source=pmc_source(rf.get("traffic_source")) if rf.get("traffic") else None,
                 launches_per_step=_r(rf.get("launches_per_step")), avg_launch_ms=_r(rf.get("avg_launch_ms")),
                 avg_launch_ms_each=_r(rf.get("avg_launch_ms_each")), algorithmic=rf.get("algorithmic"))
    return e


def compact(out):
    """(the one printed line, the side file's content).  The line (4 kB budget).  The driver's parser keeps the contract fields and
    `config` / `roofline` / `cpu_baseline` whole, so what answers "what fraction of which roof, on the metric's configurations, and what
    does a reference-shaped call cost" lives inside those three:
      config      the headline workload; `metric_2d` -- the metric's 2D configuration (1024^2 CNN) with its own roofline and CPU baseline;
                  `dropin` -- per configuration [tuned ms, four-argument simulate() ms, operator-by-operator ms]; `weak_n1` -- the z-slab
                  workload at N = 1 (the base of the N > 1 series); `weights`
      roofline    the headline's dominant kernel (physical fraction; SURVEY 8d's model figure as frac_model; `frac_out_of_cache`: the same
                  kernel on the 512 x 512 x 256 grid, whose working set exceeds the Infinity Cache)
    Beside them: `configs` -- one row per BASELINE.json configuration; `other` -- the remaining workloads; kernel times; at N > 1 the
    drivers' numbers and `comm`."""
    name = out["config"]["workload"]
    line = {k: out[k] for k in ("metric",)}
    line.update(value=_r(out["value"], 6), unit=out["unit"], n_gpus=out["n_gpus"], steps=out["steps"], warmup=out["warmup"],
                ms_per_step=_r(out["ms_per_step"], 6), higher_is_better=True, scaling="weak", vs_baseline=None, dtype=out.get("dtype", "f32"),
                data="synthetic", steps_per_s=_r(out["steps_per_s"], 6))
    every = {name: out}
    every.update(out.get("also", {}))
    rows = {k: (_short(v) if "error" not in v else dict(error=v["error"][:100])) for k, v in every.items()}
    cfgs = {f"configs[{i}]": dict(workload=k, **rows[k]) for i, k in enumerate(BASELINE_CONFIGS) if k in rows}
    if cfgs:
        line["configs"] = cfgs
        line["util_of"] = "P: PMC bytes/HBM peak, C: compulsory bytes/HBM peak, M: issued FLOPs/fp32 MFMA peak"
    c = out["config"]
    line["config"] = {k: c.get(k) for k in ("workload", "grid_per_gpu", "global_grid", "method", "jacobi_iters", "parallelism",
                                            "launch", "driver", "developed_steps", "world_size", "backend") if c.get(k) is not None}
    line["config"]["static_flags"] = "tuned: flags+BCs promised static; dropin: detected by simulate()"
    line["config"]["state_finite"] = c.get("state_finite_after_timing")
    line["config"]["weights"] = "seeded (pretrained blob absent from the reference)"
    line["roofline"] = _roof(out["roofline"], long=True)
    hbm = every.get("plume3d_hbm_jacobi")
    if hbm and "error" not in hbm and out["roofline"]["bound"] == "hbm":
        line["roofline"]["frac_out_of_cache"] = _r(hbm["roofline"].get("frac_traffic"), 3)     # 512 x 512 x 256: beyond the 256 MiB Infinity Cache
    # the metric's 2D configuration with its own roofline and CPU baseline, inside `config` (which the driver's parser keeps)
    m2 = every.get("plume2d_1024_cnn")
    if m2 and "error" not in m2 and name != "plume2d_1024_cnn":
        e = dict(workload="plume2d_1024_cnn", ms=_r(m2["ms_per_step"], 5), steps_per_s=_r(m2["steps_per_s"], 5), Mcells_s=_r(m2["value"], 5),
                 roofline=_roof(m2["roofline"]))
        if "cpu_baseline_cnn" in out:
            e["cpu_baseline"] = {kk: (_r(vv) if isinstance(vv, float) else vv) for kk, vv in out["cpu_baseline_cnn"].items()}
        line["config"]["metric_2d"] = e
    m3 = every.get("plume3d_256_jacobi")
    if m3 and "error" not in m3 and name != "plume3d_256_jacobi":
        line["config"]["metric_3d"] = dict(workload="plume3d_256_jacobi", ms=_r(m3["ms_per_step"], 5), steps_per_s=_r(m3["steps_per_s"], 5),
                                           Mcells_s=_r(m3["value"], 5), roofline=_roof(m3["roofline"]))
    sl = every.get("plume3d_slab_jacobi")
    if sl and "error" not in sl and name != "plume3d_slab_jacobi":
        line["config"]["weak_n1"] = dict(workload="plume3d_slab_jacobi", ms=_r(sl["ms_per_step"], 5), Mcells_s=_r(sl["value"], 5),
                                         note="N > 1 runs this workload per GPU: the base of the weak-scaling series")
    drops = {k: v.get("dropin") for k, v in every.items() if isinstance(v, dict) and v.get("dropin")}
    if drops:
        d = dict(what="[tuned ms, 4-argument simulate(mconf,batch_dict,net,method) eager ms, fused=False operator-by-operator ms]")
        for k, v in drops.items():
            d[k] = [v.get("error")[:60]] if "error" in v else [_r(v["tuned_ms"], 4), _r(v["four_arg_ms"], 4), _r(v["operators_ms"], 4)]
        line["config"]["dropin"] = d
    # (the opt-in bf16 precision modes are never a headline: their rows are in the side file only)
    other = {k: v for k, v in rows.items() if k not in BASELINE_CONFIGS and k not in METRIC_CONFIGS and not any(k.endswith("_" + b) for b in BF16_MODES)}
    if other:
        line["other"] = other
    line["kernel_ms_per_step"] = dict({k: _r(v) for k, v in out.get("kernel_ms_per_step", {}).items()},
                                      note="HIP-event pairs: ~2 us per launch above kernel time")
    if out.get("advect"):
        ad = out["advect"]
        line["advect"] = dict(ms=_r(ad["ms_per_step"]), frac_of_120B_model=_r(ad["frac_of_model"], 3), valu_issue_frac=_r(ad.get("valu_issue_frac"), 3))
    if "cpu_baseline" in out:
        line["cpu_baseline"] = {kk: (_r(vv) if isinstance(vv, float) else vv) for kk, vv in out["cpu_baseline"].items()}
    for k in ("native_driver", "native_driver_peer", "native_driver_rccl", "native_driver_rccl_graph", "native_driver_rccl_eager", "python_driver",
              "comm", "comm_peer", "comm_rccl"):
        if isinstance(out.get(k), dict):
            line[k] = {kk: (_r(vv) if isinstance(vv, float) else vv) for kk, vv in out[k].items() if kk != "middle_rank_model"}
    mm = out.get("native_driver", {}).get("middle_rank_model")
    if mm:                                   # (short form: the sentence that says what it is stays in the side file)
        line["middle_rank_model"] = {kk.replace("modelled_efficiency", "eff"): _r(vv, 3) for kk, vv in mm.items() if kk != "what" and not kk.startswith("ms_at_")}
        line["middle_rank_model"]["note"] = "1 GPU as rank 1 of 3, link-model comm (9 us: peer-store, 20-25: RCCL; beside_: deep_beside + direct sends): a MODEL"
    line["detail_file"] = "gpurun_out/bench_detail.json"
    return line, out


def self_spawn(a):
    """`--gpus N` without a launcher: re-execute under torch.distributed.run with N ranks on this node."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write(f"bench: --gpus {a.gpus} without WORLD_SIZE: launching {a.gpus} ranks under torch.distributed.run\n")
    return subprocess.call(cmd)


def dry_run(a, rank, world):
    """Launcher check without GPUs: gloo rendezvous, an all-reduce over the ranks, one JSON line from rank 0."""
    import datetime
    import torch
    import torch.distributed as dist
    got = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
        t = torch.ones(1)
        dist.all_reduce(t)
        got = int(t.item())
        dist.barrier()
    if rank == 0:
        print(json.dumps(dict(metric="fluid time-step throughput, Mcells/s = cells*steps/s/1e6 (steps/s alongside)", value=None,
                              unit="Mcells/s", n_gpus=a.gpus, steps=a.steps, warmup=a.warmup, dry_run=True,
                              world_size=dist.get_world_size() if world > 1 else 1, ranks_counted=got, backend="gloo",
                              config=dict(workload=a.workload or "plume3d_slab_jacobi"))))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay of the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--schedule", default="deep_first", choices=["deep_first", "deep_beside", "edge_first", "last_pass"],
                    help="N > 1: how the slab driver orders a sweep block around its ghost exchange (slab.py)")
    ap.add_argument("--no-also", action="store_true", help="skip the other configurations reported under 'also' at N=1")
    ap.add_argument("--dry-run", action="store_true", help="launcher check only: gloo rendezvous, no GPU work")
    ap.add_argument("--no-native", action="store_true", help="skip the C++ z-slab driver leg reported as 'native_driver'")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 on a box with ONE GPU: every rank takes cuda:0 and the process group is gloo (RCCL refuses two ranks on one "
                         "device, so the Python driver's P2P leg fails and the RCCL legs are skipped -- which exercises the fall-back paths); "
                         "the C++ driver runs over the peer-store transport, which sets `value`.  The launcher, the self-spawn, the watchdogs "
                         "and the JSON line are the real job's; the timings mean nothing (N ranks share a device)")
    ap.add_argument("--no-rccl-graph", action="store_true", help="N > 1: skip the last leg (the RCCL step captured in a HIP graph and replayed)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the reference-shaped legs (4-argument simulate(), operator by operator) reported under config.dropin")
    ap.add_argument("--peer-schedule", default="deep_beside", choices=["deep_first", "deep_beside", "edge_first", "last_pass"],
                    help="N > 1: the sweep-block schedule of the peer-store leg (deep_beside takes the transport's direct sends)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "peer"],
                    help="N > 1: whose time becomes `value` -- the C++ driver over RCCL send/recv (what north_star names; the default) or over "
                         "the peer-store communicator; both legs are run and printed either way")
    a = ap.parse_args()
    assert a.gpus >= 1

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_spawn(a))
    rank = int(os.environ.get("RANK", 0)); world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        sys.exit(f"bench: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} (or without a "
                 f"launcher, which self-spawns)")
    if a.dry_run:
        return dry_run(a, rank, world)

    import torch
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    rehearse = a.rehearse_one_gpu and world > 1
    if rehearse:
        local = 0
        a.transport = "peer"
    if ndev < (1 if rehearse else world) or not torch.cuda.is_available():
        sys.exit(f"bench: {world} rank(s) requested but {ndev} HIP device(s) visible -- the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        if rehearse:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=600))   # fail, do not hang
        assert dist.get_world_size() == a.gpus
    name = a.workload or (N1_HEADLINE if world == 1 else "plume3d_slab_jacobi")
    if world > 1 and not WORKLOADS[name].get("slab"):
        sys.exit(f"bench: workload {name} does not shard (single-GPU configuration); N > 1 runs plume3d_slab_jacobi")
    if world > 1:
        # A job of more than one rank has never run before the driver's multi-GPU run: whatever happens, rank 0 prints a line.
        # (a leg that hangs -- a collective nobody joins -- is ended by this timer; legs that raise are caught one by one below)
        import threading

        def give_up():
            if rank == 0:
                FALLBACK.publish(json.dumps(dict(metric="fluid time-step throughput, Mcells/s = cells*steps/s/1e6 (steps/s alongside)", value=None,
                                      unit="Mcells/s", n_gpus=a.gpus, steps=a.steps, warmup=a.warmup, higher_is_better=True, scaling="weak",
                                      vs_baseline=None, dtype="f32", data="synthetic", config=dict(workload=name),
                                      error="no leg of the N > 1 job finished within 780 s"), separators=(",", ":")))
            os._exit(0)
        if rank == 0:
            # (and if rank 0 DIES before any leg is through -- a GPU fault in the first multi-GPU step of its life -- the monitor says so)
            FALLBACK.arm(json.dumps(dict(metric="fluid time-step throughput, Mcells/s = cells*steps/s/1e6 (steps/s alongside)", value=None,
                                         unit="Mcells/s", n_gpus=a.gpus, steps=a.steps, warmup=a.warmup, higher_is_better=True, scaling="weak",
                                         vs_baseline=None, dtype="f32", data="synthetic", config=dict(workload=name),
                                         fallback="printed by the monitor process: rank 0 ended before any leg of the N > 1 job had finished"),
                                    separators=(",", ":")))
        job_dog = threading.Timer(780.0, give_up)
        job_dog.daemon = True
        job_dog.start()
        JOB_DOG.append(job_dog)
    try:
        if rehearse:
            raise RuntimeError("skipped in the one-GPU rehearsal (gloo carries no GPU send / recv)")
        out = run_workload(name, a.steps, a.warmup, not a.no_graph, world, rank, dev, a.schedule,
                           dropin=not a.no_dropin and (a.workload is not None or name in DROPIN))
    except Exception as e:  # noqa: BLE001
        if world == 1:
            raise
        # the Python driver's leg failed: the C++ driver's legs below still run, from a fresh state they develop themselves
        w0 = WORKLOADS[name]
        from fluidnet_cxx_amd.slab import SlabLayout
        lay = SlabLayout(w0["D"] * world, world, rank, halo=6)
        run_workload.slab_state = (plume_state_torch(w0["res"], lay.D_local, dev, lay.z_offset, lay.D_global), mconf_for(w0))
        run_workload.fresh_state = True
        cells0 = w0["res"] * w0["res"] * w0["D"]
        out = dict(metric="fluid time-step throughput, Mcells/s = cells*steps/s/1e6 (steps/s alongside)", value=0.0, unit="Mcells/s",
                   steps_per_s=0.0, n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=None, dtype="f32", step_hbm_frac=0.0,
                   python_driver_error=f"{type(e).__name__}: {e}"[:300],
                   config=dict(workload=name, grid_per_gpu=[w0["D"], w0["res"], w0["res"]], global_grid=[w0["D"] * world, w0["res"], w0["res"]],
                               method="jacobi", jacobi_iters=w0["iters"], parallelism=f"{world} z-slabs", launch="eager", developed_steps=0,
                               state_finite_after_timing=None),
                   kernel_ms_per_step={},
                   roofline=dict(bound="hbm", kernel="jacobi3d_march2_kernel<false,false,3>", achieved=0.0, peak=HBM_PEAK_GBS, unit="GB/s", frac=0.0,
                                 algorithmic=f"16 B/cell/sweep x {w0['iters']} sweeps x {cells0} owned cells per step"))
    out["config"]["world_size"] = dist.get_world_size() if world > 1 else 1
    out["config"]["backend"] = ("gloo (REHEARSAL: all ranks on one GPU, timings meaningless)" if rehearse else "nccl (RCCL)") if world > 1 else None
    if world == 1 and a.workload is None and not a.no_also:
        # the other configurations the metric / north star name (single-GPU by definition), measured in the same run
        out["also"] = {}
        for other in ALSO:
            big = other in ("plume3d_256_cnn", "plume3d_hbm_jacobi", "plume3d_256_cnn_bf16x6", "plume3d_256_cnn_bf16x3")
            try:
                r = run_workload(other, min(a.steps, 5 if big else 20), min(a.warmup, 2 if big else 5), not a.no_graph, 1, 0, dev,
                                 dropin=not a.no_dropin and other in DROPIN)
                out["also"][other] = {k: r[k] for k in ("value", "unit", "steps_per_s", "samples_per_s", "ms_per_step", "step_hbm_frac", "steps",
                                                        "config", "roofline", "kernel_ms_per_step", "dtype", "advect", "dropin")}
            except Exception as e:  # noqa: BLE001  (an "also" line must not take the headline down)
                out["also"][other] = dict(error=f"{type(e).__name__}: {e}")
            torch.cuda.empty_cache()
    if rank == 0 and not a.no_cpu_baseline and world == 1:       # the host baseline is reported with the single-GPU line only
        out["cpu_baseline"] = cpu_baseline(WORKLOADS[name])
    if rank == 0 and not a.no_cpu_baseline and world == 1 and a.workload is None and not a.no_also:
        out["cpu_baseline_cnn"] = cpu_baseline(WORKLOADS["plume2d_1024_cnn"], budget_s=6.0)
    done = []

    def emit():
        for t in JOB_DOG:
            t.cancel()                       # (one line only: the job's watchdog must not add its own behind this one)
        if rank == 0 and not done:
            done.append(1)
            try:
                line, detail = compact(out)
            except Exception as e:  # noqa: BLE001  (whatever a leg left behind: the job still prints its line)
                line = dict(metric=out.get("metric"), value=out.get("value"), unit=out.get("unit"), n_gpus=out.get("n_gpus"), steps=out.get("steps"),
                            warmup=out.get("warmup"), ms_per_step=out.get("ms_per_step"), higher_is_better=True, scaling="weak", vs_baseline=None,
                            dtype="f32", data="synthetic", config=dict(workload=out.get("config", {}).get("workload")),
                            error=f"compact(): {type(e).__name__}: {e}"[:300])
                detail = dict(error=line["error"])
            try:
                os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
                with open(os.path.join(REPO, "gpurun_out", "bench_detail.json"), "w") as f:
                    json.dump(detail, f, indent=1)
            except OSError as e:
                line["detail_file"] = f"not written ({e})"
            FALLBACK.publish(json.dumps(line, separators=(",", ":")))

    def arm_fallback(before):
        """the line as it stands, with the legs run so far, in the monitor's hands (rank 0, N > 1)"""
        if rank != 0 or world == 1:
            return
        import copy
        snap = copy.deepcopy(out)
        try:
            finish()
            line, _ = compact(out)
            line["fallback"] = f"printed by the monitor process: rank 0 ended during the leg '{before}'; the legs before it are in this line"
            FALLBACK.arm(json.dumps(line, separators=(",", ":")))
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"bench: fallback line not armed ({e})\n")
        finally:
            out.clear(); out.update(snap)

    def finish():
        """N > 1: the C++ driver is the product path -- the leg `--transport` names sets `value` (RCCL eager unless its captured twin
        finished, finite and faster); the Python driver's time is kept beside it"""
        nd = out.get("native_driver")
        if nd is None:
            return
        g = out.get("native_driver_rccl_graph")
        if (g and "error" not in g and g.get("state_finite") and "error" not in nd and "RCCL" in (nd.get("transport") or "")
                and g["ms_per_step"] < nd["ms_per_step"]):
            out["native_driver_rccl_eager"], out["native_driver"] = nd, g
            nd = g
        if WORKLOADS[name].get("slab"):
            out["config"]["driver"] = "python (slab.py over torch.distributed P2P)"
        if world > 1 and "error" not in nd and nd.get("state_finite"):
            if "python_driver_error" in out:
                out["python_driver"] = dict(error=out.pop("python_driver_error"))
                out["ms_per_step"] = nd["ms_per_step"]
            # both drivers issue the same kernels and exchanges (same bits, tests/test_slab.py)
            if "python_driver" not in out:
                out["python_driver"] = dict(ms_per_step=out["ms_per_step"], value=out["value"], unit="Mcells/s", steps=out["steps"])
            pyms = out["python_driver"].get("ms_per_step", nd["ms_per_step"])
            out["value"], out["ms_per_step"] = nd["value"], nd["ms_per_step"]
            out["steps_per_s"] = 1e3 / nd["ms_per_step"]
            out["step_hbm_frac"] = out["step_hbm_frac"] * pyms / nd["ms_per_step"]
            out["config"]["driver"] = ("native (fnx_slab_step: C++ driver, " + ("peer-store transport" if "peer-store" in (nd.get("transport") or "")
                                       else "RCCL ncclSend/ncclRecv issued from C++") + ")")
            out["config"]["launch"] = nd["launch"]
            out["config"]["state_finite_after_timing"] = nd.get("state_finite")
            if nd.get("roofline"):
                out["roofline"] = nd["roofline"]       # the dominant kernel as THIS driver launches it
        for k in ("native_driver", "native_driver_peer", "native_driver_rccl", "native_driver_rccl_graph", "native_driver_rccl_eager"):
            if isinstance(out.get(k), dict):
                out[k].pop("roofline", None)
        if out.get("native_driver") is out.get("native_driver_peer"):
            out.pop("native_driver_peer", None)        # (`--transport peer`: it IS native_driver)
        if out.get("comm") is out.get("comm_peer"):
            out.pop("comm_peer", None)

    # the z-slab workload through the C++ driver: the headline at N > 1; at N = 1 a row beside the headline (+ the middle-rank model)
    slab_name = name if WORKLOADS[name].get("slab") else ("plume3d_slab_jacobi" if "plume3d_slab_jacobi" in out.get("also", {})
                                                           and "error" not in out["also"]["plume3d_slab_jacobi"] else None)
    if slab_name and not a.no_native:
        # the same step through the C++ driver.  It has never met more than one GPU before the driver's multi-GPU run, so
        # it runs under a watchdog: if it is not through in time the Python-driver line is printed without it and the job ends
        import threading

        def bail():
            out["native_driver"] = dict(error="the native-driver leg did not finish within its time limit")
            finish()
            emit()
            os._exit(0)
        dog = threading.Timer(180.0, bail)
        dog.daemon = True
        dog.start()
        arm_fallback("C++ driver over RCCL")
        try:
            bd_s, m_s = run_workload.slab_state
            if rehearse:
                raise RuntimeError("skipped in the one-GPU rehearsal (RCCL refuses two ranks on one device)")
            out["native_driver"], comm_info = run_native_slab(a.steps, a.warmup, world, rank, dev, bd_s, m_s, WORKLOADS[slab_name]["res"],
                                                              WORKLOADS[slab_name]["D"], a.schedule)
            if comm_info:
                out["comm"] = comm_info
        except Exception as e:  # noqa: BLE001
            out["native_driver"] = dict(error=f"{type(e).__name__}: {e}")
        dog.cancel()
        if world > 1:
            # the same C++ driver over the peer-store transport (hipIpc-mapped mailboxes + flags, no RCCL on the data path), under its
            # own watchdog: this leg has never met more than one GPU either
            def bail_peer():
                out["native_driver_peer"] = dict(error="the peer-store leg did not finish within its time limit")
                finish()
                emit()
                os._exit(0)
            dog = threading.Timer(180.0, bail_peer)
            dog.daemon = True
            dog.start()
            arm_fallback("C++ driver over the peer-store transport")
            if os.environ.get("FNX_BENCH_TEST_CRASH") == "peer" and rank == world - 1:
                os.kill(os.getpid(), 9)                    # (tests: a rank dies the way a GPU fault kills it -- no Python runs)
            try:
                ndp, comm_p = run_native_slab(a.steps, a.warmup, world, rank, dev, bd_s, m_s, WORKLOADS[slab_name]["res"], WORKLOADS[slab_name]["D"],
                                              a.peer_schedule, transport="peer")
                out["native_driver_peer"] = ndp
                if comm_p:
                    out["comm_peer"] = comm_p
            except Exception as e:  # noqa: BLE001
                out["native_driver_peer"] = dict(error=f"{type(e).__name__}: {e}")
            dog.cancel()
            ndp = out["native_driver_peer"]
            if a.transport == "peer" and "error" not in ndp and ndp.get("state_finite"):
                out["native_driver_rccl"], out["native_driver"] = out["native_driver"], ndp
                if "comm_peer" in out:
                    if out.get("comm"):
                        out["comm_rccl"] = out["comm"]
                    out["comm"] = out["comm_peer"]
        rccl_leg = out.get("native_driver_rccl", out.get("native_driver")) or dict(error="none")
        if world > 1 and not rehearse and "error" not in rccl_leg and not a.no_rccl_graph:
            # LAST (a hang here ends the job with everything above already in the line): the RCCL leg once more with the step -- its
            # ncclSend / ncclRecv included -- captured in a HIP graph and replayed (RCCL >= 2.9 records its kernels into a capturing
            # stream).  The link model prices replay at ~0.25 ms per step; no multi-GPU box has run it yet, so it only ever REPLACES the
            # eager RCCL figure when it finishes, is finite and is faster.
            def bail_graph():
                out["native_driver_rccl_graph"] = dict(error="the captured-RCCL leg did not finish within its time limit")
                finish()
                emit()
                os._exit(0)
            dog = threading.Timer(150.0, bail_graph)
            dog.daemon = True
            dog.start()
            arm_fallback("C++ driver over RCCL captured in a HIP graph")
            try:
                ndg, _ = run_native_slab(a.steps, a.warmup, world, rank, dev, bd_s, m_s, WORKLOADS[slab_name]["res"], WORKLOADS[slab_name]["D"],
                                         a.schedule, capture=True)
                out["native_driver_rccl_graph"] = ndg
            except Exception as e:  # noqa: BLE001
                out["native_driver_rccl_graph"] = dict(error=f"{type(e).__name__}: {e}"[:200])
            dog.cancel()
        finish()
    emit()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
