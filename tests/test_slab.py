"""z-slab decomposition (config C5): the decomposed step must reproduce the single-domain step bit-for-bit on every
owned plane.  CPU: world_size-2 gloo processes (and an in-process lock-step run of 3 slabs) with the oracle plugged
in as the operator set -- this checks the decomposition logic itself (halo widths, global-z geometry, exchange
plumbing).  GPU: the same lock-step run on one device with the native HIP operators."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from util import PLUME_CFG, make_flags

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(PLUME_CFG, jacobiIter=11, gravityVec=dict(x=0.0, y=-1.0, z=0.3))


def global_state(D, H, W, seed=0):
    """A 3D state with obstacles, inflow BCs and CFL < 1 everywhere."""
    rng = np.random.default_rng(seed)
    flags = make_flags(1, D, H, W, boxes=True)
    st = dict(flags=flags, p=np.zeros((1, 1, D, H, W), np.float32),
              U=(rng.standard_normal((1, 3, D, H, W)) * 1.5).astype(np.float32),
              density=rng.random((1, 1, D, H, W)).astype(np.float32))
    UBC = np.zeros_like(st["U"]); UBCInvMask = np.ones_like(st["U"])
    UBC[0, 1, :, 0:3, 4:9] = 1.0; UBCInvMask[:, :, :, 0:3] = 0
    dBC = np.zeros_like(st["density"]); dMask = np.ones_like(st["density"])
    dBC[0, 0, :, 0:3, 4:9] = 0.5; dMask[0, 0, :, 0:3, 4:9] = 0
    st.update(UBC=UBC, UBCInvMask=UBCInvMask, densityBC=dBC, densityBCInvMask=dMask)
    return st


class OracleOps:
    """The CPU oracle behind the slab driver's operator interface (torch CPU tensors in/out)."""

    def __init__(self):
        from oracle import oracle as O
        self.O = O

    def set_slab(self, z, d):
        self.O.set_slab(z, d)

    def advect_scalar(self, dt, rho, U, flags, strength, so):
        return torch.from_numpy(self.O.advect_scalar(dt, rho.numpy(), U.numpy(), flags.numpy(), "maccormackFluidNet", 1, so, strength))

    # The oracle has no compute windows: with `windows=True` it accepts them and always produces whole fields, which
    # drives the driver's overlapped schedule (posted U/rho exchange, interior first, edges after the wait) over gloo.
    windows = False

    def __getattr__(self, name):
        if name == "set_window" and self.windows:
            return lambda a, b: None
        if name == "advect_both" and self.windows:
            return self._advect_both
        raise AttributeError(name)

    def _advect_both(self, dt, rho, U, flags, strength, so, out_rho=None, out_U=None):
        r = self.advect_scalar(dt, rho, U, flags, strength, so)
        u = self.advect_vel(dt, U, flags, strength)
        if out_rho is not None:
            out_rho.copy_(r); out_U.copy_(u)
            return out_rho, out_U
        return r, u

    def advect_vel(self, dt, U, flags, strength):
        return torch.from_numpy(self.O.advect_vel(dt, U.numpy(), U.numpy(), flags.numpy(), "maccormackFluidNet", 1, strength))

    def pre_projection(self, U_adv, rho_adv, st, cfg):
        O = self.O
        n = {k: v.numpy() for k, v in st.items()}
        U, rho = O.set_const_vals(U_adv.numpy(), n["UBC"], n["UBCInvMask"], rho_adv.numpy(), n["densityBC"], n["densityBCInvMask"])
        gv = cfg["gravityVec"]
        g = (np.array([gv["x"], gv["y"], gv["z"]], np.float32) * np.float32(-cfg["buoyancyScale"])).astype(np.float32)
        U = O.add_buoyancy(U, n["flags"], rho, g, cfg["operatingDensity"], cfg["dt"])
        U = O.set_wall_bcs(U, n["flags"])
        U, rho = O.set_const_vals(U, n["UBC"], n["UBCInvMask"], rho, n["densityBC"], n["densityBCInvMask"])
        st["U"].copy_(torch.from_numpy(U)); st["density"].copy_(torch.from_numpy(rho))
        return torch.from_numpy(O.velocity_divergence(U, n["flags"]))

    def jacobi_sweeps(self, flags, div, p, k):
        p.copy_(torch.from_numpy(self.O.jacobi_sweeps(flags.numpy(), div.numpy(), p.numpy(), True, k)))

    def jacobi_pass(self, flags, div, p_in, p_out, n, k_begin, k_end):
        full = torch.from_numpy(self.O.jacobi_sweeps(flags.numpy(), div.numpy(), p_in.numpy(), True, n))
        if k_end <= k_begin:
            p_out.copy_(full)
        else:
            p_out[:, :, k_begin:k_end].copy_(full[:, :, k_begin:k_end])

    def max_abs(self, x):
        return x.abs().max()

    # ---- the CNN projection's pieces (SlabSimulator._convnet_projection) ----
    def convnet_stage(self, U_adv, rho_adv, st, cfg):
        O = self.O
        n = {k: v.numpy() for k, v in st.items()}
        U, rho = O.set_const_vals(U_adv.numpy(), n["UBC"], n["UBCInvMask"], rho_adv.numpy(), n["densityBC"], n["densityBCInvMask"])
        gv = cfg["gravityVec"]
        g = (np.array([gv["x"], gv["y"], gv["z"]], np.float32) * np.float32(-cfg["buoyancyScale"])).astype(np.float32)
        U = O.add_buoyancy(U, n["flags"], rho, g, cfg["operatingDensity"], cfg["dt"])
        U, rho = O.set_const_vals(U, n["UBC"], n["UBCInvMask"], rho, n["densityBC"], n["densityBCInvMask"])
        st["U"].copy_(torch.from_numpy(U)); st["density"].copy_(torch.from_numpy(rho))

    def divergence(self, U, flags):
        return torch.from_numpy(self.O.velocity_divergence(U.numpy(), flags.numpy()))

    def occupancy(self, flags):
        return torch.from_numpy(self.O.flags_to_occupancy(flags.numpy()))

    def multiscale(self, net, x, trim=None):
        return torch.from_numpy(net(x.numpy(), trim) if trim is not None and any(trim) else net(x.numpy()))

    def convnet_post(self, pn, Un, s, st):
        O = self.O
        n = {k: v.numpy() for k, v in st.items()}
        U = O.velocity_update(pn.numpy(), Un.numpy(), n["flags"])
        sn = s.numpy()
        U = (U * sn).astype(np.float32)
        U = O.set_wall_bcs(U, n["flags"])
        U, rho = O.set_const_vals(U, n["UBC"], n["UBCInvMask"], n["density"], n["densityBC"], n["densityBCInvMask"])
        st["U"].copy_(torch.from_numpy(U)); st["density"].copy_(torch.from_numpy(rho))
        st["p"].copy_(torch.from_numpy((pn.numpy() * sn).astype(np.float32)))

    def post_projection(self, st, density_bc_applied=False):
        O = self.O
        n = {k: v.numpy() for k, v in st.items()}
        U = O.velocity_update(n["p"], n["U"], n["flags"])
        U = O.set_wall_bcs(U, n["flags"])
        U, rho = O.set_const_vals(U, n["UBC"], n["UBCInvMask"], n["density"], n["densityBC"], n["densityBCInvMask"])
        st["U"].copy_(torch.from_numpy(U)); st["density"].copy_(torch.from_numpy(rho))


def reference_steps(gs, nsteps):
    from oracle import oracle as O
    st = dict(gs)
    for _ in range(nsteps):
        st = O.simulate_step(st, CFG, "jacobi")
    return st


def local_state(gs, layout, dev="cpu"):
    return {k: layout.scatter(torch.from_numpy(v)).to(dev) for k, v in gs.items()}


def check_owned(st, ref, layout, what):
    for k in ("U", "density", "p"):
        a = st[k][:, :, layout.owned_slice].cpu().numpy()
        b = ref[k][:, :, layout.z_begin:layout.z_begin + layout.owned]
        bad = a.view(np.int32) != b.view(np.int32)
        assert not bad.any(), f"{what}: {k} differs on {int(bad.sum())} owned cells (rank {layout.rank}), max {np.abs(a - b).max():.3e}"


@pytest.mark.parametrize("world,halo,w,schedule", [(3, 6, 4, "last_pass"), (2, 5, 2, "last_pass"), (2, 8, 5, "last_pass"),
                                                   (2, 6, 3, "last_pass"), (2, 6, 6, "last_pass"),
                                                   (2, 6, 4, "edge_first"), (3, 6, 3, "edge_first"), (2, 5, 5, "edge_first"),
                                                   (2, 6, 4, "deep_first"), (3, 6, 3, "deep_first"), (2, 5, 5, "deep_first"),
                                                   (3, 6, 6, "deep_first"), (2, 6, 2, "deep_first"),
                                                   (2, 6, 4, "deep_beside"), (3, 6, 3, "deep_beside"), (2, 5, 5, "deep_beside"),
                                                   (3, 6, 6, "deep_beside"), (2, 6, 2, "deep_beside")])
def test_lockstep_slabs_match_single_domain_cpu(world, halo, w, schedule):
    from fluidnet_cxx_amd.slab import SlabLayout, SlabSimulator, lockstep_step
    D, H, W = 24 if (world == 3 or w == 6) else 20, 14, 18
    if schedule != "last_pass":
        D = 4 * w * world                     # the schedule needs 4w owned planes per rank
    gs = global_state(D, H, W)
    ref = reference_steps(gs, 2)
    ops = OracleOps()
    layouts = [SlabLayout(D, world, r, halo) for r in range(world)]
    sims = [SlabSimulator(l, CFG, ops=ops, sweeps_per_exchange=w, schedule=schedule) for l in layouts]
    states = [local_state(gs, l) for l in layouts]
    for n in range(2):
        lockstep_step(sims, states, defer=(n == 1 and schedule != "last_pass"))
    for l, st in zip(layouts, states):
        check_owned(st, ref, l, f"lockstep world={world}")


def test_cfl_guard_raises_on_every_rank():
    """max |U| dt > 1 breaks the assumption the ghost widths rest on: the step must refuse loudly (on all ranks: the
    number is all-reduced), not read stale ghost planes."""
    from fluidnet_cxx_amd.slab import SlabLayout, SlabSimulator, lockstep_step
    D, H, W, world = 24, 12, 16, 2
    gs = global_state(D, H, W)
    gs["U"] = gs["U"] * 20.0                           # CFL ~ 20 * 1.5 * 4 sigma * dt
    layouts = [SlabLayout(D, world, r, 6) for r in range(world)]
    sims = [SlabSimulator(l, CFG, ops=OracleOps(), sweeps_per_exchange=4, schedule="last_pass") for l in layouts]
    states = [local_state(gs, l) for l in layouts]
    with pytest.raises(RuntimeError, match="CFL <= 1"):
        lockstep_step(sims, states)
    # the guard is periodic: with cfl_check_every=0 nothing is checked (and nothing promised)
    sims = [SlabSimulator(l, CFG, ops=OracleOps(), sweeps_per_exchange=4, schedule="last_pass", cfl_check_every=0) for l in layouts]
    lockstep_step(sims, [local_state(gs, l) for l in layouts])


@pytest.mark.parametrize("tol", [1e9, 1e-30, "mid"])
def test_ptol_exit_matches_single_domain(tol):
    """pTol > 0 (the reference's per-sweep convergence test): the decomposed solve takes the same number of sweeps as the
    single-domain one -- the residual is all-reduced over the ranks -- and gives the same bits."""
    from fluidnet_cxx_amd.slab import SlabLayout, SlabSimulator, lockstep_step
    from oracle import oracle as O
    D, H, W, world = 24, 12, 16, 3
    gs = global_state(D, H, W)
    if tol == "mid":
        # a tolerance that stops the solve part-way: the first of a log grid whose result differs from both extremes
        lo = O.simulate_step(dict(gs), dict(CFG, pTol=1e9), "jacobi")["p"]
        hi = O.simulate_step(dict(gs), dict(CFG, pTol=1e-30), "jacobi")["p"]
        for cand in (3.0, 1.0, 0.3, 0.1, 0.03, 0.01, 3e-3, 1e-3):
            mid = O.simulate_step(dict(gs), dict(CFG, pTol=cand), "jacobi")["p"]
            if not np.array_equal(mid, lo) and not np.array_equal(mid, hi):
                tol = cand
                break
        assert tol != "mid", "no tolerance of the grid stops the solve part-way"
    cfg = dict(CFG, pTol=tol)
    ref = O.simulate_step(dict(gs), cfg, "jacobi")
    layouts = [SlabLayout(D, world, r, 6) for r in range(world)]
    sims = [SlabSimulator(l, cfg, ops=OracleOps(), sweeps_per_exchange=4, schedule="edge_first") for l in layouts]
    states = [local_state(gs, l) for l in layouts]
    lockstep_step(sims, states)
    for l, st in zip(layouts, states):
        check_owned(st, ref, l, f"pTol={tol}")


def _dist_worker(rank, world, port, D, H, W, halo, w, schedule, out_dir):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from fluidnet_cxx_amd.slab import SlabLayout, SlabSimulator
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        gs = global_state(D, H, W)
        layout = SlabLayout(D, world, rank, halo)
        ops = OracleOps(); ops.windows = True
        cfg = dict(CFG, pTol=0.05) if schedule == "ptol" else CFG
        sim = SlabSimulator(layout, cfg, ops=ops, sweeps_per_exchange=w, schedule="edge_first" if schedule == "ptol" else schedule)
        st = local_state(gs, layout)
        for _ in range(2):
            sim.step(st)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{k: st[k][:, :, layout.owned_slice].numpy() for k in ("U", "density", "p")})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("schedule", ["last_pass", "edge_first", "deep_first", "deep_beside", "ptol"])
def test_gloo_two_ranks_match_single_domain(tmp_path, schedule):
    """world_size 2 over gloo: real send/recv between two processes ("ptol": the per-sweep residual all-reduce of the
    pTol > 0 solve; every variant also runs the CFL guard's all-reduce)."""
    import torch.multiprocessing as mp
    from fluidnet_cxx_amd.slab import SlabLayout
    D, H, W, halo, w, world = (32 if schedule in ("edge_first", "deep_first", "deep_beside") else 24), 12, 16, 6, 4, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    os.environ.setdefault("OMP_NUM_THREADS", "2")
    mp.spawn(_dist_worker, args=(world, port, D, H, W, halo, w, schedule, str(tmp_path)), nprocs=world, join=True)
    if schedule == "ptol":
        from oracle import oracle as O
        ref = dict(global_state(D, H, W))
        for _ in range(2):
            ref = O.simulate_step(ref, dict(CFG, pTol=0.05), "jacobi")
    else:
        ref = reference_steps(global_state(D, H, W), 2)
    for r in range(world):
        l = SlabLayout(D, world, r, halo)
        z = np.load(tmp_path / f"rank{r}.npz")
        for k in ("U", "density", "p"):
            assert np.array_equal(z[k], ref[k][:, :, l.z_begin:l.z_begin + l.owned]), (k, r)


CNN_CFG = dict(CFG, model="ScaleNet", inputChannels=dict(div=True, pDiv=False, UDiv=False), normalizeInput=True,
               normalizeInputChan="UDiv", is3D=True, normalizeInputThreshold=1e-5)


def _cnn_reference(gs, nsteps):
    from oracle import oracle as O
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    blob = O.pack_weights(make_scalenet_weights(0, ndim=3), 3)
    st = dict(gs)
    for _ in range(nsteps):
        st = O.simulate_step(st, CNN_CFG, "convnet", blob)
    # (a rank's window: the nested crops of SlabSimulator._convnet_projection through the oracle's own towers)
    return st, (lambda x, trim=None: O.multiscale_forward_crop(blob, x, trim) if trim is not None else O.multiscale_forward(blob, x, True))


def test_nested_crop_margins_are_exact_and_tight_cpu():
    """SlabSimulator.NET_MARGIN / NET_MARGIN_HALF / NET_MARGIN_FULL (= FNX_SLAB_NET_MARGIN*): on a rank's window of owned +- 48 planes
    with the half- / full-resolution towers on owned +- 24 / 8, the oracle's towers give the single-domain pressure on every
    owned plane BIT FOR BIT (direct convolutions: same sums) -- and one step less of either margin does not."""
    from oracle import oracle as O
    from fluidnet_cxx_amd.slab import SlabSimulator as S
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    O.build()
    blob = O.pack_weights(make_scalenet_weights(0, ndim=3), 3)
    rng = np.random.default_rng(0)
    D, H, W, G = 144, 8, 12, S.NET_MARGIN
    x = rng.standard_normal((1, 2, D, H, W)).astype(np.float32)
    x[:, 1] = (rng.random((1, D, H, W)) < 0.1)                         # an occupancy channel
    full = O.multiscale_forward(blob, x, True)
    own = (56, 88)
    e0, e1 = own[0] - G, own[1] + G
    xw = np.ascontiguousarray(x[:, :, e0:e1])

    def owned_exact(mf, mh):
        trim = [G - mf, G - mf, G - mh, G - mh]
        pc = O.multiscale_forward_crop(blob, xw, trim)
        lo = e0 + trim[0]
        return np.array_equal(pc[:, :, own[0] - lo:own[1] - lo].view(np.int32), full[:, :, own[0]:own[1]].view(np.int32))
    assert owned_exact(S.NET_MARGIN_FULL, S.NET_MARGIN_HALF)
    assert not owned_exact(S.NET_MARGIN_FULL - 4, S.NET_MARGIN_HALF)
    assert not owned_exact(S.NET_MARGIN_FULL, S.NET_MARGIN_HALF - 4)
    # at a domain face nothing is trimmed on that side: rank 0's window [0, owned + 48)
    xw0 = np.ascontiguousarray(x[:, :, :32 + G])
    p0 = O.multiscale_forward_crop(blob, xw0, [0, G - S.NET_MARGIN_FULL, 0, G - S.NET_MARGIN_HALF])
    assert np.array_equal(p0[:, :, :32].view(np.int32), full[:, :, :32].view(np.int32))


def _check_cnn_owned(st, ref, layout, what, density_exact=True):
    """density bit for bit in a step from a common state (it does not pass through the net); p and U within 1e-5 of |ref|max (the
    std's summation order and the launch geometry of a crop differ from the single domain's)"""
    own = slice(layout.z_begin, layout.z_begin + layout.owned)
    a = st["density"][:, :, layout.owned_slice].cpu().numpy(); b = ref["density"][:, :, own]
    if density_exact:
        assert np.array_equal(a.view(np.int32), b.view(np.int32)), f"{what}: density differs on rank {layout.rank}"
    for k in ("p", "U") + (() if density_exact else ("density",)):
        a = st[k][:, :, layout.owned_slice].cpu().numpy().astype(np.float64); b = ref[k][:, :, own]
        scale = float(np.abs(ref[k]).max())
        d = float(np.abs(a - b).max())
        assert d <= 1e-5 * scale, f"{what}: {k} on rank {layout.rank}: max |d| = {d:.3e} > 1e-5 * {scale:.3e}"


def test_lockstep_convnet_slabs_match_single_domain_cpu():
    """The CNN projection on z-slabs (SlabSimulator(method='convnet')): two slabs of a 128-plane domain in lock-step with the
    oracle as operator set and as net, two steps, against oracle.simulate_step(..., 'convnet') on the whole domain."""
    from fluidnet_cxx_amd.slab import SlabLayout, SlabSimulator, lockstep_step
    D, H, W, world, halo = 128, 12, 16, 2, 52
    gs = global_state(D, H, W, seed=4)
    gs["U"] = (gs["U"] * 0.4).astype(np.float32)                      # CFL < 1
    ref1, net = _cnn_reference(gs, 1)
    ref2, _ = _cnn_reference(ref1, 1)
    layouts = [SlabLayout(D, world, r, halo) for r in range(world)]
    sims = [SlabSimulator(l, CNN_CFG, ops=OracleOps(), method="convnet", net=net) for l in layouts]
    states = [local_state(gs, l) for l in layouts]
    for n, ref in enumerate((ref1, ref2)):
        lockstep_step(sims, states)
        for l, st in zip(layouts, states):
            # (the second step starts from velocities that agree to rounding only: its density too is a tolerance statement)
            _check_cnn_owned(st, ref, l, f"cpu lockstep convnet, step {n + 1}", density_exact=(n == 0))


def _dist_cnn_worker(rank, world, port, D, H, W, halo, out_dir):
    import torch.distributed as dist
    from fluidnet_cxx_amd.slab import SlabLayout, SlabSimulator
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gs = global_state(D, H, W, seed=4)
        gs["U"] = (gs["U"] * 0.4).astype(np.float32)
        _, net = _cnn_reference(gs, 0)
        layout = SlabLayout(D, world, rank, halo)
        st = local_state(gs, layout)
        sim = SlabSimulator(layout, CNN_CFG, ops=OracleOps(), method="convnet", net=net)
        sim.step(st)
        np.savez(os.path.join(out_dir, f"cnn_rank{rank}.npz"), **{k: st[k][:, :, layout.owned_slice].numpy() for k in ("U", "density", "p")})
    finally:
        dist.destroy_process_group()


def test_gloo_two_ranks_convnet_match_single_domain(tmp_path):
    """world_size 2 over gloo: the (sum, sumsq) all-gather of the _ScaleNet std in rank order and the 49-plane exchange of the
    normalised velocity between two processes; one step against the single-domain oracle."""
    import torch.multiprocessing as mp
    from fluidnet_cxx_amd.slab import SlabLayout
    D, H, W, halo, world = 128, 12, 16, 52, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    os.environ.setdefault("OMP_NUM_THREADS", "2")
    mp.spawn(_dist_cnn_worker, args=(world, port, D, H, W, halo, str(tmp_path)), nprocs=world, join=True)
    gs = global_state(D, H, W, seed=4)
    gs["U"] = (gs["U"] * 0.4).astype(np.float32)
    ref, _ = _cnn_reference(gs, 1)
    for r in range(world):
        l = SlabLayout(D, world, r, halo)
        z = np.load(tmp_path / f"cnn_rank{r}.npz")
        _check_cnn_owned({k: torch.from_numpy(np.pad(z[k], [(0, 0), (0, 0), (l.lo, l.hi), (0, 0), (0, 0)])) for k in ("U", "density", "p")}, ref, l,
                         "gloo convnet")


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_lockstep_convnet_slabs_match_single_domain_gpu(world):
    """The CNN projection on z-slabs with the native operators: `world` slabs of 64 planes in lock-step on one device against the
    single-domain `simulate(..., 'convnet')`, two steps (p, U within 1e-5 of |ref|max; the first step's density bit for bit)."""
    from fluidnet_cxx_amd import FluidNet, simulate
    from fluidnet_cxx_amd.slab import SlabLayout, SlabSimulator, lockstep_step
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    dev = torch.device("cuda:0")
    D, H, W, halo = 64 * world, 40, 72, 52
    gs = global_state(D, H, W, seed=6)
    gs["U"] = (gs["U"] * 0.4).astype(np.float32)
    net = FluidNet.from_weights(CNN_CFG, make_scalenet_weights(0, ndim=3), dev)
    bd = {k: torch.from_numpy(v).to(dev) for k, v in gs.items()}
    layouts = [SlabLayout(D, world, r, halo) for r in range(world)]
    sims = [SlabSimulator(l, CNN_CFG, method="convnet", net=net, static_flags=True) for l in layouts]
    states = [local_state(gs, l, dev) for l in layouts]
    for n in range(2):
        simulate(CNN_CFG, bd, net, "convnet")
        ref = {k: bd[k].cpu().numpy() for k in ("U", "density", "p")}
        lockstep_step(sims, states)
        for l, st in zip(layouts, states):
            _check_cnn_owned(st, ref, l, f"gpu lockstep convnet world={world}, step {n + 1}", density_exact=(n == 0))


def test_layout_arithmetic():
    from fluidnet_cxx_amd.slab import SlabLayout
    ls = [SlabLayout(512, 8, r, 6) for r in range(8)]
    assert [l.z_begin for l in ls] == [64 * r for r in range(8)]
    assert ls[0].D_local == 70 and ls[3].D_local == 76 and ls[7].D_local == 70
    assert ls[0].z_offset == 0 and ls[1].z_offset == 58 and ls[7].z_offset + ls[7].D_local == 512
    assert all(l.owned == 64 for l in ls)


@pytest.mark.gpu
@pytest.mark.parametrize("world,halo,w,schedule,iters", [(2, 6, 4, "last_pass", 11), (4, 6, 3, "last_pass", 11), (2, 6, 6, "last_pass", 11),
                                                         (2, 6, 4, "edge_first", 11), (4, 6, 3, "edge_first", 11), (2, 6, 6, "edge_first", 11),
                                                         (2, 5, 5, "edge_first", 11), (3, 6, 5, "edge_first", 11),
                                                         (3, 6, 4, "edge_first", 14), (2, 6, 6, "edge_first", 20), (4, 6, 2, "edge_first", 6),
                                                         (2, 6, 4, "deep_first", 11), (4, 6, 3, "deep_first", 11), (2, 6, 6, "deep_first", 11),
                                                         (2, 5, 5, "deep_first", 11), (3, 6, 5, "deep_first", 11), (3, 6, 4, "deep_first", 14),
                                                         (3, 6, 6, "deep_first", 20), (4, 6, 2, "deep_first", 6),
                                                         (2, 6, 4, "deep_beside", 11), (4, 6, 3, "deep_beside", 11), (2, 6, 6, "deep_beside", 11),
                                                         (2, 5, 5, "deep_beside", 11), (3, 6, 5, "deep_beside", 11), (3, 6, 4, "deep_beside", 14),
                                                         (3, 6, 6, "deep_beside", 20), (4, 6, 2, "deep_beside", 6)])
def test_lockstep_slabs_match_single_domain_gpu(world, halo, w, schedule, iters):
    """Same decomposition check with the native HIP operators on one device (global-z geometry in the kernels).  Even
    sweep blocks with an even sweep count (H = 20): every pass is a two-sweep pass and they hand each other the pressure
    in the solver's row-quad layout, ghost planes included."""
    from fluidnet_cxx_amd import simulate
    from fluidnet_cxx_amd.slab import SlabLayout, SlabSimulator, lockstep_step
    dev = torch.device("cuda:0")
    CFG = dict(globals()["CFG"], jacobiIter=iters)
    D, H, W = (4 * w * world if schedule != "last_pass" else 32), 20, 70
    gs = global_state(D, H, W, seed=3)
    bd = {k: torch.from_numpy(v).to(dev) for k, v in gs.items()}
    for _ in range(2):
        simulate(CFG, bd, None, "jacobi")
    ref = {k: bd[k].cpu().numpy() for k in ("U", "density", "p")}
    layouts = [SlabLayout(D, world, r, halo) for r in range(world)]
    # (static_flags: the solver's neighbour mask of step 1 is reused in step 2, as bench.py runs it)
    sims = [SlabSimulator(l, CFG, sweeps_per_exchange=w, schedule=schedule, static_flags=(w == 6)) for l in layouts]
    states = [local_state(gs, l, dev) for l in layouts]
    for n in range(2):
        lockstep_step(sims, states, defer=(n == 1))          # both orders an asynchronous transfer can take
    for l, st in zip(layouts, states):
        check_owned(st, ref, l, f"gpu lockstep world={world}")
    # and against the CPU oracle (single domain)
    if iters == 11:
        oref = reference_steps(gs, 2)
        for k in ("U", "density", "p"):
            assert np.array_equal(ref[k], oref[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("world,halo,w,static,iters", [(2, 6, 4, False, 11), (4, 6, 3, False, 11), (2, 6, 6, True, 11), (3, 6, 5, True, 11),
                                                       (2, 5, 5, False, 11), (2, 6, 4, "thin", 11), (1, 6, 6, True, 11),
                                                       (3, 6, 4, True, 14), (2, 6, 6, False, 20), (1, 6, 6, True, 12)])
@pytest.mark.parametrize("schedule", ["deep_beside", "deep_first", "edge_first"])
def test_native_driver_threads_match_single_domain_gpu(world, halo, w, static, iters, schedule):
    """The C++ z-slab driver (fnx_slab_step) on `world` slabs of one domain, each driven by its own host thread and HIP
    stream on one device, ghost planes through the in-process communicator (event-ordered device copies): every owned
    plane bit-identical to the single-domain step, 3 steps (the third reuses the solver mask and the BC class map under
    the static promise).  "thin": slabs too thin for the edge-first schedule (the last-pass schedule of the driver)."""
    import threading
    from fluidnet_cxx_amd import simulate
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout
    dev = torch.device("cuda:0")
    thin = static == "thin"
    CFG = dict(globals()["CFG"], jacobiIter=iters)       # (even counts with even sweep blocks: row-quad hand-over between the passes)
    D = (2 * w * world if thin else 4 * w * world) if world > 1 else 24
    H, W = 20, 70
    gs = global_state(D, H, W, seed=5)
    bd = {k: torch.from_numpy(v).to(dev) for k, v in gs.items()}
    nsteps = 3
    for _ in range(nsteps):
        simulate(CFG, bd, None, "jacobi")
    ref = {k: bd[k].cpu().numpy() for k in ("U", "density", "p")}
    layouts = [SlabLayout(D, world, r, halo) for r in range(world)]
    states = [local_state(gs, l, dev) for l in layouts]
    group = ext.SlabLoopbackGroup(world)
    sims = [NativeSlabSimulator(l, CFG, comm=ext.slab_comm_loopback(group, l.rank) if world > 1 else None, sweeps_per_exchange=w,
                                static_flags=bool(static) and not thin, cfl_check_every=2, schedule=schedule) for l in layouts]
    torch.cuda.synchronize()
    errs = []

    def run(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                for _ in range(nsteps):
                    sims[r].step(states[r])
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append((r, e))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ts), "a rank's thread hangs"
    assert not errs, errs
    torch.cuda.synchronize()
    for l, st in zip(layouts, states):
        check_owned(st, ref, l, f"native driver world={world} w={w}")


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3])
def test_native_driver_convnet_threads_match_single_domain_gpu(world):
    """The C++ driver's CNN projection (fnx_slab_step, prm.method 1): `world` slabs of 64 planes, one host thread and stream
    each, the in-process communicator (the std's all-gather as an exact float all-reduce, the 49-plane exchange of U): two steps
    against the single-domain `simulate(..., 'convnet')` -- p, U within 1e-5 of |ref|max, the first step's density bit for bit."""
    import threading
    from fluidnet_cxx_amd import FluidNet, simulate
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    dev = torch.device("cuda:0")
    D, H, W, halo = 64 * world, 40, 72, 52
    gs = global_state(D, H, W, seed=6)
    gs["U"] = (gs["U"] * 0.4).astype(np.float32)
    net = FluidNet.from_weights(CNN_CFG, make_scalenet_weights(0, ndim=3), dev)
    bd = {k: torch.from_numpy(v).to(dev) for k, v in gs.items()}
    refs = []
    for _ in range(2):
        simulate(CNN_CFG, bd, net, "convnet")
        refs.append({k: bd[k].cpu().numpy() for k in ("U", "density", "p")})
    layouts = [SlabLayout(D, world, r, halo) for r in range(world)]
    states = [local_state(gs, l, dev) for l in layouts]
    group = ext.SlabLoopbackGroup(world)
    sims = [NativeSlabSimulator(l, CNN_CFG, comm=ext.slab_comm_loopback(group, l.rank) if world > 1 else None, static_flags=True,
                                cfl_check_every=2, method="convnet", net=net) for l in layouts]
    torch.cuda.synchronize()
    for n in range(2):
        errs = []

        def run(r):
            try:
                with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    sims[r].step(states[r])
                    torch.cuda.current_stream().synchronize()
            except Exception as e:  # noqa: BLE001
                errs.append((r, e))
        ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in ts), "a rank's thread hangs"
        assert not errs, errs
        torch.cuda.synchronize()
        for l, st in zip(layouts, states):
            _check_cnn_owned(st, refs[n], l, f"native convnet driver world={world}, step {n + 1}", density_exact=(n == 0))


@pytest.mark.gpu
def test_native_driver_failing_rank_releases_its_neighbour():
    """A rank whose step fails (here: its workspace is too small) never joins the exchange its neighbour is already waiting
    in; the driver aborts the communicator, and the neighbour returns an error instead of hanging."""
    import threading
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout
    dev = torch.device("cuda:0")
    world, w, D = 2, 4, 32
    gs = global_state(D, 20, 70, seed=2)
    layouts = [SlabLayout(D, world, r, 6) for r in range(world)]
    states = [local_state(gs, l, dev) for l in layouts]
    group = ext.SlabLoopbackGroup(world)
    sims = [NativeSlabSimulator(l, CFG, comm=ext.slab_comm_loopback(group, l.rank), sweeps_per_exchange=w, cfl_check_every=0) for l in layouts]
    sims[1]._driver(states[1])
    sims[1]._ws = torch.empty(1024, dtype=torch.uint8, device=dev)          # far too small
    errs = {}

    def run(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                sims[r].step(states[r])
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001
            errs[r] = str(e)

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in ts), "the neighbour of the failed rank hangs"
    assert "workspace too small" in errs.get(1, ""), errs
    assert "peer rank failed" in errs.get(0, ""), errs


@pytest.mark.gpu
def test_loopback_allreduce_timeout_then_retry_three_ranks():
    """The in-process communicator's all-reduce with a rank that arrives too late: the round in which a rank timed out is abandoned
    as a whole -- every rank of it fails -- and the group stays usable: the retry of all three ranks gives the exact sum (a resumed
    round would add the retrying ranks twice; set_timeout is what makes the wait short)."""
    import threading
    import time
    from fluidnet_cxx_amd._ext import ext
    dev = torch.device("cuda:0")
    world = 3
    group = ext.SlabLoopbackGroup(world)
    group.set_timeout(0.5)
    comms = [ext.slab_comm_loopback(group, r) for r in range(world)]
    xs = [torch.tensor([1.0 + r, 10.0 * (r + 1)], device=dev) for r in range(world)]
    errs, ok = {}, {}

    def run(r, delay):
        try:
            time.sleep(delay)
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                comms[r].allreduce_sum_(xs[r])
                torch.cuda.current_stream().synchronize()
            ok[r] = True
        except Exception as e:  # noqa: BLE001
            errs[r] = str(e)

    # round 1: ranks 0 and 1 arrive, rank 2 stays away longer than the timeout -> both fail, nothing is left folded in
    ts = [threading.Thread(target=run, args=(r, 0.0)) for r in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
    assert not any(t.is_alive() for t in ts)
    assert set(errs) == {0, 1} and all("timed out" in e for e in errs.values()), errs
    assert [x.tolist() for x in xs[:2]] == [[1.0, 10.0], [2.0, 20.0]], "a failed all-reduce leaves its operand alone"
    # round 2: all three retry (staggered, inside the timeout)
    errs.clear()
    ts = [threading.Thread(target=run, args=(r, 0.1 * r)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
    assert not any(t.is_alive() for t in ts) and not errs, errs
    for x in xs:
        assert x.tolist() == [6.0, 60.0], x.tolist()
    # max, for completeness
    ys = [torch.tensor([float(r)], device=dev) for r in range(world)]

    def runmax(r):
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            comms[r].allreduce_max_(ys[r])
            torch.cuda.current_stream().synchronize()
    ts = [threading.Thread(target=runmax, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
    assert [y.item() for y in ys] == [2.0, 2.0, 2.0]


@pytest.mark.gpu
def test_native_driver_timeout_aborts_group_and_reset_clears_it():
    """Through the DRIVER a peer timeout is a failed step: fnx_slab_step aborts the group (a step cannot be resumed half-way), every
    later call fails with the abort message until SlabLoopbackGroup.reset(), after which two fresh steps of both ranks match the single
    domain bit for bit."""
    import threading
    from fluidnet_cxx_amd import simulate
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout
    dev = torch.device("cuda:0")
    world, w, D = 2, 4, 32
    gs = global_state(D, 20, 70, seed=4)
    layouts = [SlabLayout(D, world, r, 6) for r in range(world)]
    states = [local_state(gs, l, dev) for l in layouts]
    group = ext.SlabLoopbackGroup(world)
    group.set_timeout(0.5)
    sims = [NativeSlabSimulator(l, CFG, comm=ext.slab_comm_loopback(group, l.rank), sweeps_per_exchange=w, cfl_check_every=0) for l in layouts]
    keep = {k: v.clone() for k, v in states[0].items()}
    with pytest.raises(RuntimeError, match="timed out"):       # rank 1 never shows up
        sims[0].step(states[0])
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="group aborted"):   # ... and the driver has aborted the group
        sims[0].step(states[0])
    torch.cuda.synchronize()
    group.reset()
    group.set_timeout(60.0)
    states[0] = keep                                            # (the failed steps had already advected rank 0's planes)
    sims = [NativeSlabSimulator(l, CFG, comm=ext.slab_comm_loopback(group, l.rank), sweeps_per_exchange=w, cfl_check_every=0) for l in layouts]
    bd = {k: torch.from_numpy(v).to(dev) for k, v in gs.items()}
    for _ in range(2):
        simulate(CFG, bd, None, "jacobi")
    ref = {k: bd[k].cpu().numpy() for k in ("U", "density", "p")}
    errs = []

    def run(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                for _ in range(2):
                    sims[r].step(states[r])
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append((r, e))
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ts) and not errs, errs
    torch.cuda.synchronize()
    for l, st in zip(layouts, states):
        check_owned(st, ref, l, "after timeout + reset")


@pytest.mark.gpu
def test_native_driver_cfl_guard_and_errors():
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout
    dev = torch.device("cuda:0")
    gs = global_state(24, 20, 70, seed=1)
    gs["U"] = gs["U"] * 40.0                                   # CFL >> 1
    l = SlabLayout(24, 1, 0, 6)
    st = local_state(gs, l, dev)
    sim = NativeSlabSimulator(l, CFG, sweeps_per_exchange=6, cfl_check_every=1)
    with pytest.raises(RuntimeError, match="max .U. dt"):
        sim.step(st)
    with pytest.raises(RuntimeError, match="needs a communicator"):
        ext.SlabDriver(1, 20, 70, 48, 0, 2, 6, 6)
    with pytest.raises(RuntimeError, match="5 valid ghost planes"):
        ext.SlabDriver(1, 20, 70, 48, 0, 2, 4, 4, comm=ext.slab_comm_loopback(ext.SlabLoopbackGroup(2), 0))


@pytest.mark.gpu
def test_native_driver_rccl_single_rank():
    """RCCL called directly from C++ (librccl resolved at run time): unique id, communicator of one rank (the box has one
    GPU), the all-reduce of the CFL guard through it, a step; same bits as the communicator-free driver."""
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout, rccl_comm
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    uid = ext.slab_rccl_unique_id()
    assert isinstance(uid, bytes) and len(uid) == 128 and any(uid)
    comm = rccl_comm(0, 1)
    gs = global_state(24, 20, 70, seed=2)
    l = SlabLayout(24, 1, 0, 6)
    a, b = local_state(gs, l, dev), local_state(gs, l, dev)
    NativeSlabSimulator(l, CFG, comm=comm, sweeps_per_exchange=6, cfl_check_every=1).step(a)
    NativeSlabSimulator(l, CFG, comm=None, sweeps_per_exchange=6, cfl_check_every=0).step(b)
    torch.cuda.synchronize()
    for k in ("U", "density", "p"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.gpu
def test_native_driver_step_is_graph_capturable():
    """The whole native step (no CFL check: that is the one host sync) captured in a HIP graph and replayed: the same
    bits as eager steps."""
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout
    dev = torch.device("cuda:0")
    gs = global_state(24, 20, 70, seed=4)
    l = SlabLayout(24, 1, 0, 6)
    a, b = local_state(gs, l, dev), local_state(gs, l, dev)
    sa = NativeSlabSimulator(l, CFG, sweeps_per_exchange=6, static_flags=True, cfl_check_every=0)
    sb = NativeSlabSimulator(l, CFG, sweeps_per_exchange=6, static_flags=True, cfl_check_every=0)
    for _ in range(2):                      # (the second step builds the BC class map the captured step reuses)
        sa.step(a); sb.step(b)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sa.step(a)
    for _ in range(3):                      # a: 2 eager steps + 3 replays of the captured one, b: 5 eager steps
        g.replay()
        sb.step(b)
    torch.cuda.synchronize()
    for k in ("U", "density", "p"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["deep_first", "deep_beside"])
def test_native_driver_middle_rank_step_is_graph_capturable(schedule):
    """A MIDDLE rank's step -- exchanges, sweep blocks with deep and edge parts, in deep_beside the second stream with the exchanges
    riding on it -- captured in a HIP graph and replayed: the same bits as eager steps.  The link-model communicator stands in for
    the neighbours (every exchange occupies its stream for latency + bytes / bandwidth and fills the ghost planes from the slab's
    own edge planes: deterministic, so eager and replayed steps must agree)."""
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout
    dev = torch.device("cuda:0")
    gs = global_state(96, 20, 70, seed=8)
    l = SlabLayout(96, 3, 1, 6)
    a, b = local_state(gs, l, dev), local_state(gs, l, dev)
    cfg = dict(CFG, jacobiIter=20)
    sa = NativeSlabSimulator(l, cfg, comm=ext.slab_comm_link_model(5.0, 100.0), sweeps_per_exchange=6, static_flags=True, cfl_check_every=0, schedule=schedule)
    sb = NativeSlabSimulator(l, cfg, comm=ext.slab_comm_link_model(5.0, 100.0), sweeps_per_exchange=6, static_flags=True, cfl_check_every=0, schedule=schedule)
    for _ in range(2):
        sa.step(a); sb.step(b)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sa.step(a)
    for _ in range(3):
        g.replay()
        sb.step(b)
    torch.cuda.synchronize()
    for k in ("U", "density", "p"):
        assert torch.equal(a[k], b[k]), k
    assert float(a["U"].abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("tol", [1e9, 1e-30, 0.3, 0.03])
def test_native_driver_ptol_matches_single_domain(tol):
    """pTol > 0 in the C++ driver: one sweep per ghost exchange, the residual summed over the ranks through the
    communicator -- the same number of sweeps and the same bits as the single-domain native solve."""
    import threading
    from fluidnet_cxx_amd import simulate
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout
    dev = torch.device("cuda:0")
    D, H, W, world = 36, 20, 70, 3
    gs = global_state(D, H, W, seed=6)
    cfg = dict(CFG, pTol=tol, jacobiIter=15)
    bd = {k: torch.from_numpy(v).to(dev) for k, v in gs.items()}
    simulate(cfg, bd, None, "jacobi")
    ref = {k: bd[k].cpu().numpy() for k in ("U", "density", "p")}
    layouts = [SlabLayout(D, world, r, 6) for r in range(world)]
    states = [local_state(gs, l, dev) for l in layouts]
    group = ext.SlabLoopbackGroup(world)
    sims = [NativeSlabSimulator(l, cfg, comm=ext.slab_comm_loopback(group, l.rank), sweeps_per_exchange=4, cfl_check_every=0)
            for l in layouts]
    torch.cuda.synchronize()
    errs = []

    def run(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                sims[r].step(states[r])
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append((r, e))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ts) and not errs, errs
    torch.cuda.synchronize()
    for l, st in zip(layouts, states):
        check_owned(st, ref, l, f"native pTol={tol}")
