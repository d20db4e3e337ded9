"""z-slab domain decomposition of the 3D time step across the GPUs of one node.

The reference has no multi-device support at all (single process, single device).  Here the 3D grid is cut along z
(the slowest axis: a ghost plane is one contiguous H*W block per channel); rank r owns `D/n` planes and keeps `halo`
ghost planes towards each neighbour.  All geometry runs in GLOBAL z coordinates (`set_slab(z_offset, D_global)` on
the native side), so every owned cell goes through bit-for-bit the same arithmetic as in a single-GPU run.

Per step (Jacobi method):
  1. exchange U, density ghosts (4 planes)                -> advection + BC/buoyancy/wall stage valid on owned +- 1
  2. exchange div ghosts (width w-1) once
  3. blocks of w Jacobi sweeps (temporal blocking in z: one message per w sweeps); the last pass of a block first
     produces the planes the neighbours need, posts the exchange, and computes the interior while it is in flight
  4. exchange p (width 1), velocity update + wall BCs + BCs on the owned planes
Communication is point-to-point between z-neighbours only (`torch.distributed` P2P = RCCL send/recv over xGMI on
GPUs, gloo in the CPU tests): each rank talks to at most two peers, there is no collective on the data path.

The driver is backend-agnostic: `ops` supplies the operators (native HIP ops on GPU tensors in production; the tests
plug in the CPU oracle on numpy-backed tensors to check the decomposition itself with gloo).
"""
from contextlib import nullcontext as _nullctx

import torch
import torch.distributed as dist


class SlabLayout:
    """Index arithmetic of one rank's slab."""

    def __init__(self, D_global, world, rank, halo):
        assert D_global % world == 0, "D must divide evenly across ranks"
        self.D_global, self.world, self.rank, self.halo = D_global, world, rank, halo
        self.owned = D_global // world
        assert world == 1 or self.owned >= halo, "slab thinner than its halo"
        self.z_begin = rank * self.owned                      # first owned global plane
        self.lo = halo if rank > 0 else 0                     # ghost planes below / above
        self.hi = halo if rank < world - 1 else 0
        self.z_offset = self.z_begin - self.lo                # global plane of local plane 0
        self.D_local = self.owned + self.lo + self.hi

    @property
    def owned_slice(self):
        return slice(self.lo, self.lo + self.owned)

    def scatter(self, full):
        """(1,C,D,H,W) global tensor -> this rank's local tensor (ghosts filled from the global data)."""
        return full[:, :, self.z_offset:self.z_offset + self.D_local].clone().contiguous()


class SlabComm:
    """Ghost-plane exchange with the two z-neighbours."""

    def __init__(self, layout, group=None):
        self.l = layout
        self.group = group

    def start(self, fields, width, sources=None):
        """Post the sends/receives of `width` ghost planes of every field; returns a handle for finish().  `sources`
        (default: the fields themselves) are the arrays whose owned edge planes are sent."""
        l = self.l
        if l.world == 1:
            return None
        assert width <= l.halo
        ops, recvs = [], []
        for f, src in zip(fields, sources if sources is not None else fields):
            if l.rank > 0:           # lower neighbour
                send = src[:, :, l.lo:l.lo + width]
                recv = f[:, :, l.lo - width:l.lo]
                direct = send.is_contiguous() and recv.is_contiguous()     # one plane block (B = C = 1): no staging copy
                sb = send if direct else send.contiguous()
                rb = recv if direct else torch.empty_like(sb)
                ops += [dist.P2POp(dist.isend, sb, l.rank - 1, self.group), dist.P2POp(dist.irecv, rb, l.rank - 1, self.group)]
                if not direct:
                    recvs.append((recv, rb))
            if l.rank < l.world - 1:  # upper neighbour
                top = l.lo + l.owned
                send = src[:, :, top - width:top]
                recv = f[:, :, top:top + width]
                direct = send.is_contiguous() and recv.is_contiguous()     # one plane block (B = C = 1): no staging copy
                sb = send if direct else send.contiguous()
                rb = recv if direct else torch.empty_like(sb)
                ops += [dist.P2POp(dist.isend, sb, l.rank + 1, self.group), dist.P2POp(dist.irecv, rb, l.rank + 1, self.group)]
                if not direct:
                    recvs.append((recv, rb))
        return dist.batch_isend_irecv(ops), recvs

    def finish(self, handle):
        if handle is None:
            return
        works, recvs = handle
        for w in works:
            w.wait()                 # NCCL: the current stream waits for the transfer; gloo: the host does
        for dst, buf in recvs:
            dst.copy_(buf)

    def exchange(self, fields, width):
        self.finish(self.start(fields, width))


class NativeOps:
    """The production operator set: hand-written HIP through the `fluidnet_cpp` extension."""

    zero_start = True            # jacobi_pass takes p_in=None for "the pressure is 0 everywhere"

    def __init__(self):
        from ._ext import ext
        self.ext = ext
        self._ws = None          # persistent Jacobi workspace: holds the 3D neighbour mask between sweep blocks
        self._ws_key = None
        self._mask_valid = False
        self.static_bcs = False  # set by the driver from the second step on when the caller promises static flags / BCs
        self._cls = None         # class map of the BC arrays (FnxState.bc_class), built once under that promise
        # z-slab view and compute window of THIS operator set: turned into an ext.Geom that travels with every call
        # (the extension itself keeps no state)
        self._slab = (0, 0)
        self._win = (0, 0)

    def begin_step(self):
        self._mask_valid = False
        self._cls = None         # (new flags / BC arrays: the class map of the BC stages goes too)

    def set_slab(self, z_offset, D_global):
        self._slab = (int(z_offset), int(D_global))

    def set_window(self, k_begin, k_end):
        self._win = (int(k_begin), int(k_end))

    def _geom(self, window=True):
        """Geometry of the next call: the slab view, plus the compute window for the plane-parallel operators (the Jacobi
        entry points take their own plane range and ignore it)."""
        return self.ext.Geom(z_offset=self._slab[0], D_global=self._slab[1], k_begin=self._win[0] if window else 0,
                             k_end=self._win[1] if window else 0)

    def advect_scalar(self, dt, rho, U, flags, strength, sample_outside, out=None):
        return self.ext.advect_scalar(dt, rho, U, flags, "maccormackFluidNet", 1, bool(sample_outside), strength, out, self._geom())

    def advect_vel(self, dt, U, flags, strength, out=None):
        return self.ext.advect_vel(dt, U, U, flags, "maccormackFluidNet", 1, strength, out, self._geom())

    def advect_both(self, dt, rho, U, flags, strength, sample_outside, out_rho=None, out_U=None):
        """density and velocity advection of one step as the fused pair of launches (same bits as the two calls)"""
        r, u = self.ext.advect_step(dt, rho, U, flags, bool(sample_outside), strength, out_rho, out_U, self._geom())
        return r, u

    def _bc_class(self, st):
        if not self.static_bcs or (st.get("UBC") is None and st.get("densityBC") is None):
            return None
        if self._cls is None:
            self._cls = self.ext.bc_classify(st["flags"], True, st.get("UBC"), st.get("UBCInvMask"), st.get("densityBC"),
                                             st.get("densityBCInvMask"))
        return self._cls

    def pre_projection(self, U_adv, rho_adv, st, cfg):
        gv = cfg["gravityVec"]
        return self.ext.pre_projection_(U_adv, rho_adv, st["p"], st["U"], st["flags"], st.get("density"), st.get("UBC"),
                                        st.get("UBCInvMask"), st.get("densityBC"), st.get("densityBCInvMask"),
                                        float(cfg["dt"]), float(cfg["buoyancyScale"]),
                                        [float(gv["x"]), float(gv["y"]), float(gv["z"])],
                                        float(cfg.get("operatingDensity", 0.0)), True, self._bc_class(st), self._geom())

    def jacobi_sweeps(self, flags, div, p, n, from_zero=False):
        """n sweeps in place; from_zero: the solve starts from p = 0 (p is not read)"""
        key = (tuple(flags.shape), flags.device)
        if self._ws is None or self._ws_key != key:
            B, _, D, H, W = flags.shape
            self._ws = torch.empty(self.ext.jacobi_workspace_bytes(B, D, H, W, True), dtype=torch.uint8, device=flags.device)
            self._ws_key, self._mask_valid = key, False
        self.ext.jacobi_sweeps_(flags, div, p, True, int(n), self._ws, self._mask_valid, self._geom(False), from_zero=bool(from_zero))
        self._mask_valid = True      # same flags for the rest of this step (begin_step resets)

    two_ranges = True            # jacobi_pass takes a second plane range of the same length (one launch for both faces)

    def quad_ok(self, flags):
        """may two-sweep passes hand each other the pressure in the solver's row-quad layout (jacobi_pass's `lay`)?"""
        B, _, D, H, W = flags.shape
        return bool(self.ext.jacobi_quad_ok(B, D, H, W))

    def jacobi_pass(self, flags, div, p_in, p_out, n, k_begin, k_end, k_begin2=-1, lay=0):
        """lay: bit 0 = p_in, bit 1 = p_out in the row-quad layout (fnx_jacobi_pass_layout)"""
        key = (tuple(flags.shape), flags.device)
        if self._ws is None or self._ws_key != key:
            B, _, D, H, W = flags.shape
            self._ws = torch.empty(self.ext.jacobi_workspace_bytes(B, D, H, W, True), dtype=torch.uint8, device=flags.device)
            self._ws_key, self._mask_valid = key, False
        self.ext.jacobi_pass_(flags, div, p_in, p_out, int(n), int(k_begin), int(k_end), self._ws, self._mask_valid,
                              int(k_begin2), self._geom(False), layout=int(lay) & (3 if p_in is not None else 2))
        self._mask_valid = True

    def max_abs(self, x):
        """max |x| as a 0-dim tensor on x's device (no host sync)"""
        return self.ext.max_abs(x)

    # ---- the CNN projection's pieces (SlabSimulator._convnet_projection) ----
    def convnet_stage(self, U_adv, rho_adv, st, cfg):
        """setConstVals, addBuoyancy, setConstVals on the window (simulate.py:96-133 of the convnet method: no wall BCs)"""
        gv = cfg["gravityVec"]
        self.ext.pre_projection_(U_adv, rho_adv, st["p"], st["U"], st["flags"], st.get("density"), st.get("UBC"),
                                 st.get("UBCInvMask"), st.get("densityBC"), st.get("densityBCInvMask"), float(cfg["dt"]),
                                 float(cfg["buoyancyScale"]), [float(gv["x"]), float(gv["y"]), float(gv["z"])],
                                 float(cfg.get("operatingDensity", 0.0)), False, self._bc_class(st), self._geom())

    def divergence(self, U, flags):
        return self.ext.velocity_divergence(U, flags, self._geom(False))

    def occupancy(self, flags):
        return self.ext.flags_to_occupancy(flags)

    def multiscale(self, net, x, trim=None):
        return net.multiScale(x, trim if trim is not None and any(trim) else None)

    def convnet_post(self, pn, Un, s, st):
        """model.py:190-227 + simulate.py:154-168 on the window: U = (Un - grad pn) * s, p = pn * s, setWallBcs, setConstVals"""
        g = self._geom()
        self.ext.velocity_update_(pn, Un, st["flags"], g)
        k0, k1 = self._win if self._win[1] > self._win[0] else (0, Un.shape[2])
        st["U"][:, :, k0:k1] = Un[:, :, k0:k1] * s
        st["p"][:, :, k0:k1] = pn[:, :, k0:k1] * s
        self.ext.set_wall_bcs_(st["U"], st["flags"], g)
        if st.get("UBC") is not None or st.get("densityBC") is not None:
            # (pointwise: the ghost planes it also touches are refreshed by the next step's exchange)
            self.ext.set_const_vals_(st["U"], st.get("UBC"), st.get("UBCInvMask"), st.get("density"), st.get("densityBC"),
                                     st.get("densityBCInvMask"))

    def post_projection(self, st, density_bc_applied=False):
        # density_bc_applied (FnxState.density_bc_applied): the caller's promise that pre_projection of THIS step has applied
        # the density BCs on the same planes with the same BC arrays -- the pass then leaves the density of identity-class cells
        # alone (same bits).  SlabSimulator._step makes it; any other caller gets the full setConstVals.
        self.ext.post_projection_(st["p"], st["U"], st["flags"], st.get("density"), st.get("UBC"), st.get("UBCInvMask"),
                                  st.get("densityBC"), st.get("densityBCInvMask"), self._bc_class(st), self._geom(),
                                  bool(density_bc_applied))


class SlabSimulator:
    """`simulate(mconf, batch_dict, None, 'jacobi')` for one rank's slab of a 3D domain (in place on `state`)."""

    # ghost planes of the normalised velocity the CNN projection needs: the MultiScaleNet's receptive field (< 48 cells at full
    # resolution: test_cnn_benchmark_size's crop margin) + 1 for the divergence; a multiple of 4 so that the quarter- and
    # half-resolution grids of a rank's crop coincide with the global ones
    NET_MARGIN = 48
    # ... of which the towers need less: the full-resolution tower's receptive radius is 8 planes (5^3, four 3^3, 5^3), the half-
    # resolution tower's 2 x (2 + 5) = 14 (+ 2 for the resampling above it, + the 8 = 24), the quarter-resolution tower's 4 x 4 = 16
    # (+ 4, + the 24 = 44 <= NET_MARGIN).  A rank evaluates each tower only on owned +- its own margin (nested crops,
    # fnx_multiscale_forward_crop) -- 1.3 x the FLOPs of its owned planes at 64 planes per rank instead of 2.5 x.  The pressure
    # is then exact on the owned planes; the plane below them that velocityUpdate reads comes from the neighbour.
    NET_MARGIN_FULL, NET_MARGIN_HALF = 8, 24

    def __init__(self, layout, mconf, ops=None, group=None, sweeps_per_exchange=4, schedule="deep_first",
                 static_flags=False, cfl_check_every=8, method="jacobi", net=None):
        """cfl_check_every: every that many steps (0 = never) the step starts by reducing max |U| dt over all ranks and
        raises RuntimeError on EVERY rank when it exceeds 1 cell -- the bound the ghost widths, the advection windows and the
        overlapped U / density exchange rest on (a violation would otherwise read stale ghost planes silently).  Costs one
        pass over U and one 4-byte all-reduce(MAX) on the control path."""
        assert schedule in ("last_pass", "edge_first", "deep_first", "deep_beside")
        assert method in ("jacobi", "convnet")
        self.method, self.net = method, net
        if method == "convnet" and layout.world > 1:
            assert net is not None, "method 'convnet' needs the net (a FluidNet, or any x -> p callable for the CPU operator sets)"
            assert layout.halo >= self.NET_MARGIN + 1 and layout.owned % 4 == 0 and layout.halo % 4 == 0, \
                "the CNN projection needs halo >= 49 ghost planes (a multiple of 4) and owned planes a multiple of 4"
        self.cfl_check_every = int(cfl_check_every)
        self.schedule = schedule
        self.static_flags = static_flags     # the caller promises that flags and BC arrays do not change between steps
        self._steps = 0
        self.l = layout
        self.cfg = mconf
        self.ops = ops if ops is not None else NativeOps()
        self.comm = SlabComm(layout, group)
        self.w = min(sweeps_per_exchange, layout.halo)
        self._pbuf = None
        assert layout.world == 1 or layout.owned >= 2 * self.w, "slab too thin for the sweep block"
        assert layout.world == 1 or layout.halo >= 5, "advection + projection need 5 valid ghost planes (CFL <= 1)"
        assert float(mconf.get("pTol", 0.0)) >= 0.0

    def _side_stream(self, t):
        """A second HIP stream for work that may run next to the main stream's (device tensors with the native
        operators only)."""
        if not (t.is_cuda and isinstance(self.ops, NativeOps)):
            return None
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=t.device)
        return self._side

    def phases(self, st):
        """Generator over the step: computes up to the next communication point and yields a request
             ("xchg", fields, width)   blocking ghost exchange
             ("start", fields, width)  post a ghost exchange and keep computing
             ("wait",)                 the posted exchange must have landed
        `step` serves the requests with the real communicator; tests drive several ranks in lock-step."""
        l, cfg, ops = self.l, self.cfg, self.ops
        dt = float(cfg["dt"])
        w = self.w
        assert "density" in st, "the z-slab driver advects a density field (simulate() without one is single-domain only)"
        if self.cfl_check_every > 0 and self._steps % self.cfl_check_every == 0 and hasattr(ops, "max_abs"):
            # CFL guard: max |U| dt over the whole domain (every rank gets the same number and raises or not together)
            m = ops.max_abs(st["U"]).reshape(1).clone()
            yield "allmax", m
            cfl = float(m.item()) * abs(dt)
            if cfl > 1.0:
                raise RuntimeError(f"z-slab step: max |U| dt = {cfl:.3f} cells > 1 -- the slab decomposition (ghost widths, advection "
                                   f"windows) is only valid for CFL <= 1; reduce dt or run the single-domain simulate()")
        if hasattr(ops, "begin_step") and not (self.static_flags and self._steps > 0):
            ops.begin_step()                 # (forget the solver's neighbour mask and the BC class map)
        if hasattr(ops, "static_bcs"):
            ops.static_bcs = bool(self.static_flags and self._steps > 0)
        self._steps += 1
        # advection reaches <= 2 planes beyond its inputs at CFL <= 1 and the BC/buoyancy/divergence stage one more:
        # 4 fresh ghost planes of U and density are enough (the arrays keep `halo` planes for the pressure solve)
        window = getattr(ops, "set_window", None)
        lo_, top_ = l.lo, l.lo + l.owned
        strength, so = float(cfg["maccormackStrength"]), cfg.get("sampleOutsideFluid", False)
        # Compute windows: only the owned planes (+1 for the advected fields, whose +1 neighbours the divergence reads)
        # are produced; ghost planes are refreshed by the exchanges anyway.  Ranks at the domain ends own their border.
        a_, b_ = max(lo_ - 1, 0), min(top_ + 1, l.D_local)
        # planes whose advection reads no ghost plane (output k reads k-3..k+3 at CFL <= 1)
        ia_ = lo_ + 3 if l.rank > 0 else a_
        ib_ = top_ - 3 if l.rank < l.world - 1 else b_
        if window and l.world > 1 and l.owned - 6 >= 8:      # (the same decision on every rank)
            # the ghost exchange of U and density is in flight while the interior planes are advected
            yield "start", [st["U"], st["density"]], min(4, l.halo)
            ops.set_slab(l.z_offset, l.D_global)
            rho_adv, U_adv = torch.empty_like(st["density"]), torch.empty_like(st["U"])
            window(ia_, ib_)
            ops.advect_both(dt, st["density"], st["U"], st["flags"], strength, so, rho_adv, U_adv)
            window(0, 0)
            yield ("wait",)
            ops.set_slab(l.z_offset, l.D_global)
            for ea, eb in ((a_, ia_), (ib_, b_)):
                if eb > ea:
                    window(ea, eb)
                    ops.advect_both(dt, st["density"], st["U"], st["flags"], strength, so, rho_adv, U_adv)
        else:
            yield "xchg", [st["U"], st["density"]], min(4, l.halo)
            ops.set_slab(l.z_offset, l.D_global)
            if window and l.world > 1:
                window(a_, b_)
            if hasattr(ops, "advect_both"):
                rho_adv, U_adv = ops.advect_both(dt, st["density"], st["U"], st["flags"], strength, so)
            else:
                rho_adv = ops.advect_scalar(dt, st["density"], st["U"], st["flags"], strength, so)
                U_adv = ops.advect_vel(dt, st["U"], st["flags"], strength)
        if window and l.world > 1:
            window(lo_, top_)
        if self.method == "convnet":
            yield from self._convnet_projection(st, U_adv, rho_adv, window, lo_, top_)
            return
        div = ops.pre_projection(U_adv, rho_adv, st, cfg)
        if window:
            window(0, 0)
        blocked = l.world > 1 and l.owned >= 4 * w and int(cfg["jacobiIter"]) > w and not float(cfg.get("pTol", 0.0)) > 0.0
        deep = self.schedule in ("deep_first", "deep_beside") and blocked
        # deep_first: the deep parts of the first sweep block read no ghost plane of div: its exchange is in flight behind them
        yield ("start" if deep else "xchg"), [div], max(w - 1, 1)

        ops.set_slab(l.z_offset, l.D_global)
        if float(cfg.get("pTol", 0.0)) > 0.0:
            cur = yield from self._jacobi_ptol(st, div)
        elif deep and self.schedule == "deep_beside":
            cur = yield from self._jacobi_deep_beside(st, div)
        elif deep:
            cur = yield from self._jacobi_deep_first(st, div)
        elif self.schedule == "edge_first" and blocked:
            cur = yield from self._jacobi_edge_first(st, div)
        else:
            cur = yield from self._jacobi_last_pass(st, div)
        if cur is not st["p"]:
            st["p"].copy_(cur)
        yield "xchg", [st["p"]], 1
        ops.set_slab(l.z_offset, l.D_global)
        if window and l.world > 1:
            window(lo_, top_)
        ops.post_projection(st, density_bc_applied=True)     # ops.pre_projection above ran on the same planes and BC arrays
        if window:
            window(0, 0)
        ops.set_slab(0, 0)

    def _convnet_projection(self, st, U_adv, rho_adv, window, lo_, top_):
        """The CNN pressure projection on a z-slab (lib/simulate.py:96-168 with sim_method 'convnet', lib/model.py:118-227).

        The staging pass (setConstVals, addBuoyancy, setConstVals -- no wall BCs in this method) runs on the owned planes.
        _ScaleNet (model.py:8-23: the unbiased std of U over the WHOLE domain) becomes a control-path reduction: every rank sums
        u and u^2 over its owned planes in fp64, the (sum, sumsq) pairs are all-gathered and added up in RANK ORDER on every rank --
        the same bits everywhere, whatever the transport's reduction order would have been (not the single-domain kernel's
        summation order: a tolerance statement, like every CNN comparison).  The normalised velocity's ghost planes (NET_MARGIN + 1)
        are exchanged ONCE; each rank then evaluates the net on its owned planes +- NET_MARGIN -- a crop whose offset and depth are
        multiples of 4, so its resampling grids coincide with the global ones and everything further than the receptive field
        from the crop's artificial faces equals the single-domain result -- and finishes (velocityUpdate, un-normalise,
        setWallBcs, setConstVals) on the owned planes.  The three towers run on NESTED crops of that window (owned +- 8 / 24 / 48
        planes for the full- / half- / quarter-resolution tower: NET_MARGIN_FULL / _HALF), so an interior rank of 64 planes spends 1.3 x
        the FLOPs of its owned planes (2.5 x with every tower on the whole window; a per-layer halo exchange of up to 128-channel
        planes would remove the rest); the pressure of the one plane below the owned ones comes from the neighbour."""
        l, cfg, ops = self.l, self.cfg, self.ops
        G = self.NET_MARGIN
        ops.convnet_stage(U_adv, rho_adv, st, cfg)                   # -> st["U"], st["density"] on the owned planes
        if window:
            window(0, 0)
        U, flags = st["U"], st["flags"]
        B, nc = U.shape[0], U.shape[1]
        own = U[:, :, lo_:top_].double()
        part = torch.stack([own.sum(dim=(1, 2, 3, 4)), (own * own).sum(dim=(1, 2, 3, 4))], 1)      # (B, 2) fp64
        gathered = [part]
        yield "allgather", part, gathered                              # -> gathered: the ranks' pairs in rank order
        tot = torch.zeros_like(part)
        for g_ in gathered:                                            # fixed order: the same bits on every rank
            tot = tot + g_.to(part.device)
        n = float(nc) * l.D_global * U.shape[3] * U.shape[4]
        var = ((tot[:, 1] - tot[:, 0] * tot[:, 0] / n) / (n - 1.0)).clamp_min(0.0)
        thr = float(cfg.get("normalizeInputThreshold", 1e-5))
        s = var.sqrt().float().clamp_min(thr).view(B, 1, 1, 1, 1)      # model.py:14-21
        Un = torch.zeros_like(U)
        Un[:, :, lo_:top_] = U[:, :, lo_:top_] / s
        if l.world > 1:
            yield "xchg", [Un], G + 1
        ops.set_slab(l.z_offset, l.D_global)
        e0 = max(lo_ - G, 0) if l.rank > 0 else 0
        e1 = min(top_ + G, l.D_local) if l.rank < l.world - 1 else l.D_local
        div = ops.divergence(Un, flags)                                 # (valid on e0 .. e1: U_z of plane e1 is a ghost plane)
        x = torch.cat([div[:, :, e0:e1], ops.occupancy(flags)[:, :, e0:e1]], 1).contiguous()
        # nested crops: where the window ends at an artificial face (a neighbour's planes go on beyond it) the full- / half-
        # resolution towers stop NET_MARGIN_FULL / _HALF planes outside the owned ones; at a domain face nothing is trimmed
        cut_lo, cut_hi = l.rank > 0 and lo_ - e0 == G, l.rank < l.world - 1 and e1 - top_ == G
        trim = [G - self.NET_MARGIN_FULL if cut_lo else 0, G - self.NET_MARGIN_FULL if cut_hi else 0,
                G - self.NET_MARGIN_HALF if cut_lo else 0, G - self.NET_MARGIN_HALF if cut_hi else 0]
        pn = torch.zeros_like(st["p"])
        pn[:, :, e0 + trim[0]:e1 - trim[1]] = ops.multiscale(self.net, x, trim)
        if l.world > 1:
            yield "xchg", [pn], 1           # exact on the owned planes; velocityUpdate also reads the neighbour's plane below them
        if window and l.world > 1:
            window(lo_, top_)
        ops.convnet_post(pn, Un, s, st)                                 # velocityUpdate, * s, setWallBcs, setConstVals (owned planes)
        if window:
            window(0, 0)
        ops.set_slab(0, 0)

    def _jacobi_ptol(self, st, div):
        """pTol > 0: the reference's convergence test (fluids_init.cpp:961-979) -- after EVERY sweep the residual
        max_b ||p_new - p_old||_2 over the whole domain is compared with pTol on the host.  Decomposed: one sweep per ghost
        exchange, the squared differences summed over the owned planes and all-reduced (a B-float collective per sweep; the
        single-domain solver pays a host sync per sweep here too).  Temporal blocking does not apply: the exit test needs
        every sweep's result."""
        l, cfg, ops = self.l, self.cfg, self.ops
        if self._pbuf is None or self._pbuf.shape != st["p"].shape or self._pbuf.device != st["p"].device:
            self._pbuf = torch.zeros_like(st["p"])
        cur, nxt = st["p"], self._pbuf
        cur.zero_()
        lo, top = l.lo, l.lo + l.owned
        a = lo if l.rank > 0 else 0
        b = top if l.rank < l.world - 1 else l.D_local
        tol = float(cfg["pTol"])
        for it in range(int(cfg["jacobiIter"])):
            if it > 0:
                yield "xchg", [cur], 1
                ops.set_slab(l.z_offset, l.D_global)
            ops.jacobi_pass(st["flags"], div, cur, nxt, 1, a, b)
            d = (nxt[:, :, lo:top] - cur[:, :, lo:top]).double()
            ss = (d * d).sum(dim=(1, 2, 3, 4))
            yield "allsum", ss
            ops.set_slab(l.z_offset, l.D_global)
            cur, nxt = nxt, cur
            if float(ss.sqrt().max().item()) < tol:
                break
        return cur

    def _jacobi_edge_first(self, st, div):
        """Jacobi schedule "edge_first": the same blocks of w sweeps and the same shrinking plane ranges as "last_pass"
        (after `done` sweeps of a block only the planes within w - done of the owned block are still needed), but every
        pass's range is cut at 2w - done planes inside each internal face and the EDGE parts of ALL passes of the block
        run first:

            edge part of pass i      [lo - w + done_i, lo + 2w - done_i)    (and its mirror image at the upper face)
            interior part of pass i  [lo + 2w - done_i, top - 2w + done_i)

        The edge parts form a closed chain (part i+1 reads exactly what part i wrote) that ends in the w owned planes
        the neighbour needs at sweep s + w; their exchange is posted as soon as the chain is through, and the interior
        parts -- all w/2 passes, not one -- run while it is in flight.  Both chains ping-pong between the SAME two
        arrays: the edge part of pass i+1 overwrites the array of pass i-1 only below plane lo + 2w - done_(i+1), and
        the interior part of pass i reads that array from plane lo + 2w - done_i - n_i upwards, which is not lower as
        long as the passes' sweep counts do not decrease (an odd block runs its single sweep first).  No plane is
        computed twice: the work is that of "last_pass"."""
        l, cfg, ops, w = self.l, self.cfg, self.ops, self.w
        if self._pbuf is None or self._pbuf.shape != st["p"].shape or self._pbuf.device != st["p"].device:
            self._pbuf = torch.zeros_like(st["p"])
        cur, nxt = st["p"], self._pbuf
        fresh = getattr(ops, "zero_start", False)
        if not fresh:
            cur.zero_()
        lo, top = l.lo, l.lo + l.owned
        has_lo, has_hi = l.rank > 0, l.rank < l.world - 1
        flags = st["flags"]
        passes = [1] * (w % 2) + [2] * (w // 2)
        remaining = int(cfg["jacobiIter"])
        # every pass of the solve is a two-sweep pass: they hand each other the pressure in the solver's row-quad layout
        # (both arrays, every plane range, the ghost planes the neighbours send -- they run the same schedule); the last
        # pass of the solve writes rows
        quad = w % 2 == 0 and remaining % 2 == 0 and hasattr(ops, "quad_ok") and ops.quad_ok(flags)
        Q = dict(lay=3) if quad else {}
        zero_in = fresh                      # the block starts from p = 0 everywhere: nothing to read
        block = 0
        while remaining > w:                 # a block that is followed by another one
            remaining -= w
            block += 1
            if block > 1:
                yield ("wait",)              # the ghost planes of `cur`
                ops.set_slab(l.z_offset, l.D_global)
            src, dst, done = cur, nxt, 0
            for pi, n in enumerate(passes):
                done += n
                pin = None if (zero_in and pi == 0) else src
                if has_lo and has_hi and getattr(ops, "two_ranges", False):       # both faces in one launch
                    ops.jacobi_pass(flags, div, pin, dst, n, lo - w + done, lo + 2 * w - done, top - 2 * w + done, **Q)
                else:
                    if has_lo:
                        ops.jacobi_pass(flags, div, pin, dst, n, lo - w + done, lo + 2 * w - done, **Q)
                    if has_hi:
                        ops.jacobi_pass(flags, div, pin, dst, n, top - 2 * w + done, top + w - done, **Q)
                src, dst = dst, src
            fin = src
            yield "start", [fin], w
            ops.set_slab(l.z_offset, l.D_global)
            src, dst, done = cur, nxt, 0
            for pi, n in enumerate(passes):
                done += n
                ops.jacobi_pass(flags, div, None if (zero_in and pi == 0) else src, dst, n,
                                lo + 2 * w - done if has_lo else 0, top - 2 * w + done if has_hi else l.D_local, **Q)
                src, dst = dst, src
            if fin is not cur:
                cur, nxt = nxt, cur
            zero_in = False

        # the last block (<= w sweeps, no exchange after it): whole shrinking ranges
        yield ("wait",)
        ops.set_slab(l.z_offset, l.D_global)
        done = 0
        tail = [2] * (remaining // 2) + [1] * (remaining % 2)
        for ti, n in enumerate(tail):
            done += n
            g = max(w - done, 0)
            kw = (Q if ti < len(tail) - 1 else dict(lay=1)) if quad else {}
            ops.jacobi_pass(flags, div, cur, nxt, n, lo - g if has_lo else 0, top + g if has_hi else l.D_local, **kw)
            cur, nxt = nxt, cur
        return cur

    @staticmethod
    def deep_splits(passes, w):
        """Plane offsets (from an internal face, into the owned block) at which pass k of a "deep_first" sweep block is cut
        into its edge part [face - w + done_k, face + s_k) and its deep part [face + s_k, ...).  s_k >= s_(k-1) + n_k: the
        deep part of pass k only reads what deeper-or-equal parts of pass k-1 wrote (block start: the owned planes).  From
        the second pass on s_k >= w: the deep parts that write into the array the block started from stay clear of the w
        owned planes next to the face, which the neighbour is still receiving from that array."""
        s, out = 0, []
        for k, n in enumerate(passes):
            s = s + n if k == 0 else max(s + n, w)
            out.append(s)
        return out

    def _jacobi_deep_first(self, st, div):
        """Jacobi schedule "deep_first": the same blocks of w sweeps as "edge_first", cut the other way round.  Pass k of a
        block (done_k sweeps into it) is split at s_k planes inside each internal face (deep_splits):

            deep part of pass k   [lo + s_k, top - s_k)              reads no ghost plane and nothing an edge part wrote
            edge part of pass k   [lo - w + done_k, lo + s_k)        (and its mirror image at the upper face)

        The DEEP parts of all passes run first -- they need neither the ghost planes the previous block's exchange is still
        delivering nor (in the first block) the ghost planes of div -- then the exchange is waited for, the edge parts
        follow (a chain of short launches: 6 / 8 / 8 planes per face at w = 6 instead of edge_first's 14 / 10 / 6, whose
        first launch alone marches 18 steps), and the w owned planes next to each face go to the neighbour while the next
        block's deep parts run.  Both chains ping-pong between the same two arrays: the deep part of pass k+1 overwrites
        the array pass k-1 wrote from plane lo + s_(k+1) upwards only, and the edge part of pass k reads it below plane
        lo + s_k + n_k <= lo + s_(k+1) (sweep counts do not decrease within a block).  No plane is computed twice."""
        l, cfg, ops, w = self.l, self.cfg, self.ops, self.w
        if self._pbuf is None or self._pbuf.shape != st["p"].shape or self._pbuf.device != st["p"].device:
            self._pbuf = torch.zeros_like(st["p"])
        cur, nxt = st["p"], self._pbuf
        fresh = getattr(ops, "zero_start", False)
        if not fresh:
            cur.zero_()
        lo, top = l.lo, l.lo + l.owned
        has_lo, has_hi = l.rank > 0, l.rank < l.world - 1
        flags = st["flags"]
        passes = [1] * (w % 2) + [2] * (w // 2)
        splits = self.deep_splits(passes, w)
        assert l.owned >= 2 * splits[-1] + 1, "slab too thin for the deep_first sweep block"
        remaining = int(cfg["jacobiIter"])
        quad = w % 2 == 0 and remaining % 2 == 0 and hasattr(ops, "quad_ok") and ops.quad_ok(flags)
        Q = dict(lay=3) if quad else {}
        zero_in = fresh                      # the solve starts from p = 0 everywhere: the first pass reads nothing
        while remaining > w:                 # a block that is followed by another one
            remaining -= w
            src, dst = cur, nxt
            for pi, n in enumerate(passes):
                ops.jacobi_pass(flags, div, None if (zero_in and pi == 0) else src, dst, n,
                                lo + splits[pi] if has_lo else 0, top - splits[pi] if has_hi else l.D_local, **Q)
                src, dst = dst, src
            yield ("wait",)                  # the ghost planes of `cur` (first block: of div)
            ops.set_slab(l.z_offset, l.D_global)
            src, dst, done = cur, nxt, 0
            for pi, n in enumerate(passes):
                done += n
                pin = None if (zero_in and pi == 0) else src
                if has_lo and has_hi and getattr(ops, "two_ranges", False):       # both faces in one launch
                    ops.jacobi_pass(flags, div, pin, dst, n, lo - w + done, lo + splits[pi], top - splits[pi], **Q)
                else:
                    if has_lo:
                        ops.jacobi_pass(flags, div, pin, dst, n, lo - w + done, lo + splits[pi], **Q)
                    if has_hi:
                        ops.jacobi_pass(flags, div, pin, dst, n, top - splits[pi], top + w - done, **Q)
                src, dst = dst, src
            if src is not cur:
                cur, nxt = nxt, cur
            yield "start", [cur], w
            ops.set_slab(l.z_offset, l.D_global)
            zero_in = False

        # the last block (<= w sweeps, no exchange after it): whole shrinking ranges
        yield ("wait",)
        ops.set_slab(l.z_offset, l.D_global)
        done = 0
        tail = [2] * (remaining // 2) + [1] * (remaining % 2)
        for ti, n in enumerate(tail):
            done += n
            g = max(w - done, 0)
            kw = (Q if ti < len(tail) - 1 else dict(lay=1)) if quad else {}
            ops.jacobi_pass(flags, div, cur, nxt, n, lo - g if has_lo else 0, top + g if has_hi else l.D_local, **kw)
            cur, nxt = nxt, cur
        return cur

    def _jacobi_deep_beside(self, st, div):
        """Jacobi schedule "deep_beside": the plane ranges, arrays and arithmetic of "deep_first" (see there), with the edge chain of
        a block issued BESIDE its deep chain instead of behind it -- on a second stream that also waits for the previous
        exchange and posts the next one:

            main stream   D0 ----------> D1 ----------> D2 ----------> [join] D0' ...
            edge stream   [wait xchg] E0 --(D0)--> E1 --(D1)--> E2, post xchg [join]

        (Dk / Ek: deep / edge part of pass k.)  What orders them is what they read and write: Ek reads the planes Dk-1 wrote
        just inside its split and writes below the split, where Dk-1 was still reading the array Ek overwrites -- so Ek waits
        for Dk-1, nothing else; the next block's D0 reads what E2 wrote and waits for the whole edge chain (the join).  A deep
        launch of the solver occupies three quarters of the wave slots (one resident set of (tile, plane chunk) waves) and
        runs at the pace of VALU issue; the short edge launches (latency-bound marches of 6-8 planes behind 4 lead-in planes)
        fill the rest instead of running alone, and the two stream hand-overs around the exchange leave the main stream.
        A block then takes max(deep chain, exchange + edge chain) instead of their sum.  Without a second stream (CPU
        operator sets of the tests) the launches are issued in the same interleaved order on one queue: same bits."""
        l, cfg, ops, w = self.l, self.cfg, self.ops, self.w
        if self._pbuf is None or self._pbuf.shape != st["p"].shape or self._pbuf.device != st["p"].device:
            self._pbuf = torch.zeros_like(st["p"])
        cur, nxt = st["p"], self._pbuf
        fresh = getattr(ops, "zero_start", False)
        if not fresh:
            cur.zero_()
        lo, top = l.lo, l.lo + l.owned
        has_lo, has_hi = l.rank > 0, l.rank < l.world - 1
        flags = st["flags"]
        passes = [1] * (w % 2) + [2] * (w // 2)
        splits = self.deep_splits(passes, w)
        assert l.owned >= 2 * splits[-1] + 1, "slab too thin for the deep_first sweep block"
        remaining = int(cfg["jacobiIter"])
        quad = w % 2 == 0 and remaining % 2 == 0 and hasattr(ops, "quad_ok") and ops.quad_ok(flags)
        Q = dict(lay=3) if quad else {}
        zero_in = fresh
        side = self._side_stream(cur)
        main = torch.cuda.current_stream(cur.device) if side is not None else None
        ev_deep = [torch.cuda.Event() for _ in passes] if side is not None else None
        import contextlib
        on_edge = (lambda: torch.cuda.stream(side)) if side is not None else contextlib.nullcontext
        first_launch = True
        # (the communication requests name the stream they are to be served on -- a generator must not yield inside a stream
        # context: lockstep_step interleaves several ranks' generators on one thread)
        while remaining > w:                 # a block that is followed by another one
            remaining -= w
            if side is not None:
                side.wait_stream(main)       # fork: behind the previous block's join (first block: behind the staging pass)
            yield ("wait", side)             # the ghost planes of `cur` (first block: of div) -- the EDGE stream waits
            ops.set_slab(l.z_offset, l.D_global)
            src, dst, done = cur, nxt, 0
            for pi, n in enumerate(passes):
                done += n
                pin = None if (zero_in and pi == 0) else src
                ops.jacobi_pass(flags, div, pin, dst, n, lo + splits[pi] if has_lo else 0, top - splits[pi] if has_hi else l.D_local, **Q)
                if side is not None:
                    ev_deep[pi].record(main)
                with on_edge():
                    if side is not None and (pi > 0 or first_launch):
                        # E_pi behind D_(pi-1); the very first launch of a step may build the solver's obstacle mask: E0 behind it
                        side.wait_event(ev_deep[pi - 1 if pi > 0 else 0])
                    if has_lo and has_hi and getattr(ops, "two_ranges", False):       # both faces in one launch
                        ops.jacobi_pass(flags, div, pin, dst, n, lo - w + done, lo + splits[pi], top - splits[pi], **Q)
                    else:
                        if has_lo:
                            ops.jacobi_pass(flags, div, pin, dst, n, lo - w + done, lo + splits[pi], **Q)
                        if has_hi:
                            ops.jacobi_pass(flags, div, pin, dst, n, top - splits[pi], top + w - done, **Q)
                first_launch = False
                src, dst = dst, src
            if src is not cur:
                cur, nxt = nxt, cur
            yield "start", [cur], w, None, side      # posted behind the edge chain, on its stream
            ops.set_slab(l.z_offset, l.D_global)
            if side is not None:
                main.wait_stream(side)       # join: the next block's D0 reads what E_last wrote
            zero_in = False

        # the last block (<= w sweeps, no exchange after it): whole shrinking ranges
        yield ("wait",)
        ops.set_slab(l.z_offset, l.D_global)
        done = 0
        tail = [2] * (remaining // 2) + [1] * (remaining % 2)
        for ti, n in enumerate(tail):
            done += n
            g = max(w - done, 0)
            kw = (Q if ti < len(tail) - 1 else dict(lay=1)) if quad else {}
            ops.jacobi_pass(flags, div, cur, nxt, n, lo - g if has_lo else 0, top + g if has_hi else l.D_local, **kw)
            cur, nxt = nxt, cur
        return cur

    def _jacobi_last_pass(self, st, div):
        """Jacobi schedule "last_pass": blocks of w sweeps between ghost exchanges (temporal blocking in z); the last
        pass of a block first produces the w planes each neighbour needs, posts their exchange, and computes the interior
        meanwhile.  One pass of compute (of w/2) is available to hide a transfer."""
        l, cfg, ops, w = self.l, self.cfg, self.ops, self.w
        if self._pbuf is None or self._pbuf.shape != st["p"].shape or self._pbuf.device != st["p"].device:
            self._pbuf = torch.zeros_like(st["p"])
        cur, nxt = st["p"], self._pbuf
        fresh = getattr(ops, "zero_start", False)      # the operator set takes p_in=None for "p is 0 everywhere"
        if l.world == 1 and fresh:
            # nobody to exchange with: the whole solve in one call (its passes hand each other p in the solver's
            # row-quad layout, which the plane-range passes below do not)
            ops.jacobi_sweeps(st["flags"], div, cur, int(cfg["jacobiIter"]), from_zero=True)
            return cur
        if not fresh:
            cur.zero_()
        lo, top = l.lo, l.lo + l.owned
        remaining, pending = int(cfg["jacobiIter"]), False
        first_pass = True
        while remaining > 0:
            k = min(w, remaining)
            remaining -= k
            if pending:
                yield ("wait",)
                ops.set_slab(l.z_offset, l.D_global)
                pending = False
            passes = [2] * (k // 2) + [1] * (k % 2)
            done = 0
            for pi, n in enumerate(passes):
                done += n
                pin = None if (fresh and first_pass) else cur      # None: p = 0 everywhere, nothing to read
                first_pass = False
                # After the exchange the w ghost planes next to the owned block are fresh; every sweep makes the
                # outermost fresh one stale, so a pass that ends `done` sweeps into the block only has to produce the
                # planes within w - done of the owned block (sides without a neighbour: up to the array end).
                g = max(self.w - done, 0)
                a = lo - g if l.rank > 0 else 0
                b = top + g if l.rank < l.world - 1 else l.D_local
                if pi == len(passes) - 1 and remaining > 0 and l.world > 1:
                    # Last pass of the block: the w planes each neighbour needs go first and their exchange is posted;
                    # the interior planes are independent of them (same input, disjoint output) and run on a side
                    # stream at the same time, so the small edge launches fill CUs instead of serialising with it.
                    ia = lo + w if l.rank > 0 else a
                    ib = top - w if l.rank < l.world - 1 else b
                    # The edge launches are issued FIRST: a pass of the solver fills every wave slot for its whole
                    # duration, so small launches issued after it would only start when it ends (and the exchange
                    # with them).
                    side = self._side_stream(cur) if ib > ia else None
                    if side is not None:
                        side.wait_stream(torch.cuda.current_stream(cur.device))
                    if l.rank > 0:
                        ops.jacobi_pass(st["flags"], div, pin, nxt, n, lo, lo + w)
                    if l.rank < l.world - 1:
                        ops.jacobi_pass(st["flags"], div, pin, nxt, n, top - w, top)
                    if side is not None:
                        with torch.cuda.stream(side):
                            ops.jacobi_pass(st["flags"], div, pin, nxt, n, ia, ib)
                    yield "start", [nxt], w
                    ops.set_slab(l.z_offset, l.D_global)
                    pending = True
                    if side is None:
                        if ib > ia:
                            ops.jacobi_pass(st["flags"], div, pin, nxt, n, ia, ib)   # overlaps the exchange
                    else:
                        torch.cuda.current_stream(cur.device).wait_stream(side)
                elif l.world > 1:
                    ops.jacobi_pass(st["flags"], div, pin, nxt, n, a, b)
                else:
                    ops.jacobi_pass(st["flags"], div, pin, nxt, n, 0, 0)
                cur, nxt = nxt, cur
        return cur

    def step(self, st):
        import contextlib
        handle = None

        def on(stream):      # a request may name the stream it is served on (deep_beside: the edge stream posts and waits)
            return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
        try:
            for req in self.phases(st):
                if req[0] == "xchg":
                    self.comm.exchange(req[1], req[2])
                elif req[0] in ("allmax", "allsum"):
                    if self.l.world > 1:
                        dist.all_reduce(req[1], op=dist.ReduceOp.MAX if req[0] == "allmax" else dist.ReduceOp.SUM, group=self.comm.group)
                elif req[0] == "allgather":
                    if self.l.world > 1:
                        out = [torch.empty_like(req[1]) for _ in range(self.l.world)]
                        dist.all_gather(out, req[1].contiguous(), group=self.comm.group)
                        req[2][:] = out
                elif req[0] == "start":
                    with on(req[4] if len(req) > 4 else None):
                        handle = self.comm.start(req[1], req[2], req[3] if len(req) > 3 else None)
                else:
                    with on(req[1] if len(req) > 1 else None):
                        self.comm.finish(handle)
                    handle = None
        finally:
            self.ops.set_slab(0, 0)
            if hasattr(self.ops, "set_window"):
                self.ops.set_window(0, 0)


class NativeSlabSimulator:
    """The same step through the C++ z-slab driver of the C ABI (`fnx_slab_create` / `fnx_slab_step`,
    csrc/fnx_slab.hip): the launch sequence, the ghost exchanges (RCCL ncclSend/ncclRecv called directly from C++, or the
    in-process communicator) and their overlap with the interior work are all native; Python only hands over the
    tensors.  pTol > 0 runs the reference's per-sweep convergence test (one host sync per sweep, as everywhere).

    comm: an `ext.SlabComm` (`rccl_comm(...)` below, or `ext.slab_comm_loopback(group, rank)`); None for one rank."""

    def __init__(self, layout, mconf, comm=None, sweeps_per_exchange=4, static_flags=False, cfl_check_every=8, batch=1,
                 H=None, W=None, schedule="deep_first", method="jacobi", net=None, direct_sends="auto"):
        """method 'convnet' (with net = a FluidNet): the CNN projection on z-slabs, fnx_slab_step with prm.method 1 -- needs
        layout.halo >= 49 (a multiple of 4) and owned planes a multiple of 4 (include/fluidnet_hip.h)."""
        from ._ext import ext
        assert method in ("jacobi", "convnet") and (method == "jacobi" or net is not None)
        self.ext, self.l, self.cfg, self.comm = ext, layout, mconf, comm
        self.method, self.net = method, net
        # direct_sends "auto" | "never" | "always": where the communicator offers them (peer-store, link model) a sweep block's last edge
        # part stores the planes the neighbours need next straight into their windows (FnxSlabConfig.direct_sends; auto: in deep_beside);
        # same bits either way
        self._args = (int(sweeps_per_exchange), bool(static_flags), int(cfl_check_every), str(schedule), str(direct_sends))
        self._drv = None
        self._ws = None

    def _driver(self, st):
        if self._drv is None:
            B, _, D, H, W = st["flags"].shape
            l = self.l
            self._drv = self.ext.SlabDriver(B, H, W, l.D_global, l.rank, l.world, l.halo, self._args[0], self._args[1],
                                            self._args[2], self.comm, self._args[3], self.method, self._args[4])
            assert self._drv.layout() == [l.owned, l.lo, l.hi, l.z_offset] and D == l.D_local
            self._ws = torch.empty(self._drv.workspace_bytes(), dtype=torch.uint8, device=st["flags"].device)
        return self._drv

    def step(self, st):
        assert "density" in st, "the z-slab driver advects a density field (simulate() without one is single-domain only)"
        cfg = self.cfg
        gv = cfg["gravityVec"]
        self._driver(st).step(st["p"], st["U"], st["flags"], st["density"], st.get("UBC"), st.get("UBCInvMask"), st.get("densityBC"),
                              st.get("densityBCInvMask"), float(cfg["dt"]), float(cfg["maccormackStrength"]),
                              bool(cfg.get("sampleOutsideFluid", False)), float(cfg["buoyancyScale"]),
                              [float(gv["x"]), float(gv["y"]), float(gv["z"])], float(cfg.get("operatingDensity", 0.0)),
                              float(cfg.get("pTol", 0.0)), int(cfg["jacobiIter"]), self._ws,
                              self.net.packed_for(st["flags"].device) if self.method == "convnet" else None,
                              getattr(self.net, "precision_mode", "fp32") if self.method == "convnet" else "fp32",
                              float(cfg.get("normalizeInputThreshold", 1e-5)))


def rccl_comm(rank, world, group=None):
    """An RCCL communicator for the native driver: rank 0 draws the unique id, `torch.distributed` (any backend) carries
    its 128 bytes to the other ranks, every rank then joins with ncclCommInitRank from C++."""
    from ._ext import ext
    box = [ext.slab_rccl_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    return ext.slab_comm_rccl(rank, world, box[0])


def peer_comm(rank, world, mailbox_bytes=16 << 20, group=None, timeout_s=30.0):
    """A peer-store communicator for the native driver (fnx_slab_peer_create / fnx_slab_comm_peer): every rank allocates its region
    (flags + mailbox), `torch.distributed` (any backend) carries the handles, each rank maps its two neighbours'.  Ghost planes then
    travel as device stores into the neighbour's mailbox + flags -- no RCCL on the data path."""
    from ._ext import ext
    peer = ext.SlabPeer(rank, world, int(mailbox_bytes))
    peer.set_timeout(float(timeout_s))
    handles = [None] * world
    if world > 1:
        dist.all_gather_object(handles, peer.handle, group=group)
    return ext.slab_comm_peer(peer, handles[rank - 1] if rank > 0 else None, handles[rank + 1] if rank < world - 1 else None)


def lockstep_step(sims, states, defer=False):
    """Single-process stand-in for n ranks: advances all slabs phase by phase and serves their ghost exchanges with
    direct copies.  Used to validate the decomposition on ONE device (tests); production uses SlabSimulator.step.
    A posted ("start") exchange is served at once by default, so any later write into the planes in flight shows up as
    a mismatch; with defer=True it is served only at the matching "wait", so any READ of a ghost plane before the wait
    (or a write into the planes being sent) shows up instead -- the two orders an asynchronous transfer can take."""
    gens = [s.phases(st) for s, st in zip(sims, states)]
    posted = None

    def serve(reqs):
        width = reqs[0][2]
        multi = any(len(q) > 4 and q[4] is not None for q in reqs)     # posted from side streams: order the copies by full syncs
        if multi:
            torch.cuda.synchronize()
        for r in range(len(sims) - 1):           # pair (r, r+1)
            lo, hi = sims[r].l, sims[r + 1].l
            srcs = [q[3] if len(q) > 3 and q[3] is not None else q[1] for q in (reqs[r], reqs[r + 1])]     # arrays the edge planes are sent from
            for f_lo, f_hi, s_lo, s_hi in zip(reqs[r][1], reqs[r + 1][1], *srcs):
                top = lo.lo + lo.owned
                f_lo[:, :, top:top + width].copy_(s_hi[:, :, hi.lo:hi.lo + width])
                f_hi[:, :, hi.lo - width:hi.lo].copy_(s_lo[:, :, top - width:top])
        if multi:
            torch.cuda.synchronize()

    while True:
        reqs = []
        for g in gens:
            try:
                reqs.append(next(g))
            except StopIteration:
                reqs.append(None)
        if all(r is None for r in reqs):
            break
        assert all(r is not None for r in reqs) and len({r[0] for r in reqs}) == 1, "ranks fell out of step"
        if reqs[0][0] == "allgather":
            parts = [r[1].clone() for r in reqs]
            for r in reqs:
                r[2][:] = [p_.to(r[1].device) for p_ in parts]
            continue
        if reqs[0][0] in ("allmax", "allsum"):
            ts = [r[1] for r in reqs]
            red = torch.stack([t.to(ts[0].device) for t in ts])
            red = red.max(0).values if reqs[0][0] == "allmax" else red.sum(0)
            for t in ts:
                t.copy_(red)
            continue
        if reqs[0][0] == "wait":
            if posted is not None:
                serve(posted)
                posted = None
            continue
        if reqs[0][0] == "start" and defer:
            assert posted is None, "two exchanges in flight"
            posted = reqs
            continue
        serve(reqs)
    assert posted is None, "an exchange was posted and never waited for"
    for s in sims:
        s.ops.set_slab(0, 0)
        if hasattr(s.ops, "set_window"):
            s.ops.set_window(0, 0)
