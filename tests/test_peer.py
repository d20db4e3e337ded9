"""The peer-store communicator (fnx_slab_peer_create / fnx_slab_comm_peer): ghost planes as device stores into mailboxes the
neighbours have mapped through hipIpc handles, ordered by flags -- no RCCL on the data path.  RCCL refuses two ranks on one device;
IPC mappings do not, so these are runs of the C++ driver (fnx_slab_step) with ONE PROCESS PER RANK on the one GPU of the box:
the first real multi-process runs of the decomposition.  Every owned plane must equal the single-domain step bit for bit
(lib/simulate.py:28-171 per owned plane; the reference itself is single device, plume.py:131-135)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _peer_comm(ext, dist, rank, world, mailbox_bytes, timeout_s=20.0):
    """one region per rank, the handles all-gathered over the (gloo) process group, the neighbours' two mapped"""
    peer = ext.SlabPeer(rank, world, mailbox_bytes)
    peer.set_timeout(timeout_s)
    handles = [None] * world
    dist.all_gather_object(handles, peer.handle)
    return ext.slab_comm_peer(peer, handles[rank - 1] if rank > 0 else None, handles[rank + 1] if rank < world - 1 else None)


def _worker(rank, world, port, case, out_dir):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout
    import test_slab as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        D, H, W, halo, w, schedule, iters, ptol, mailbox, nsteps, direct = case
        cfg = dict(T.CFG, jacobiIter=iters, pTol=ptol)
        gs = T.global_state(D, H, W, seed=7)
        layout = SlabLayout(D, world, rank, halo)
        st = T.local_state(gs, layout, dev)
        comm = _peer_comm(ext, dist, rank, world, mailbox)
        sim = NativeSlabSimulator(layout, cfg, comm=comm, sweeps_per_exchange=w, static_flags=True, cfl_check_every=2, schedule=schedule,
                                  direct_sends=direct)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            for _ in range(nsteps):
                sim.step(st)
            torch.cuda.current_stream().synchronize()
        # a probe of the transport itself: average ms of a 1 MiB exchange with each neighbour
        scratch = torch.zeros(4 << 20, dtype=torch.uint8, device=dev)
        ms = ext.slab_comm_probe(comm, 1 << 20, 20, scratch)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), probe_ms=ms, **{k: st[k][:, :, layout.owned_slice].cpu().numpy() for k in ("U", "density", "p")})
        dist.barrier()
        del sim, comm
    finally:
        dist.destroy_process_group()


# D, H, W, halo, w, schedule, iters, pTol, mailbox bytes, steps, direct sends ('auto': in deep_beside; 'never'; 'always')
# (direct: the last edge part of a sweep block stores the planes the neighbours need next straight into their mailbox slots,
#  fnx_jacobi_pass_mirror + FnxSlabComm.direct_exchange; the default where the communicator offers it)
CASES = {
    "deep_first":   (48, 20, 70, 6, 6, "deep_first", 20, 0.0, 1 << 20, 3, "always"),
    "deep_beside":  (48, 20, 70, 6, 4, "deep_beside", 14, 0.0, 1 << 20, 3, "auto"),
    "no_direct":    (48, 20, 70, 6, 6, "deep_beside", 20, 0.0, 1 << 20, 3, "never"),     # the same exchanges through the push copies
    "odd_block":    (40, 20, 70, 6, 5, "deep_beside", 16, 0.0, 1 << 20, 2, "auto"),      # blocks of 1 + 2 + 2 sweeps, rows (no row-quad hand-over)
    "small_slots":  (48, 20, 70, 6, 6, "deep_beside", 20, 0.0, 16384, 2, "always"),      # 6 planes (33 600 B) do not fit a slot: direct_begin declines, chunked pushes
    "chunked":      (32, 20, 70, 6, 4, "edge_first", 11, 0.0, 8192, 2, "auto"),          # mailbox slots smaller than a plane (5 600 B x w): cut into chunks
    "ptol":         (32, 20, 70, 6, 4, "last_pass", 30, 0.05, 1 << 20, 2, "auto"),       # the per-sweep residual all-reduce along the chain
}


@pytest.mark.parametrize("world,name", [(2, "deep_first"), (3, "deep_first"), (2, "deep_beside"), (3, "deep_beside"), (2, "no_direct"),
                                        (2, "odd_block"), (2, "small_slots"), (3, "chunked"), (3, "ptol"),
                                        # eight ranks (round 6: what the driver's multi-GPU run has, rehearsed as eight processes on the one
                                        # GPU): six middle ranks with two neighbours each under deep_beside + direct sends, the handle
                                        # exchange for eight, and the chain all-reduce (CFL guard, pTol residual) in rank order over eight
                                        (8, "deep_beside"), (8, "deep_first"), (8, "ptol"), (8, "chunked")])
def test_peer_store_processes_match_single_domain(tmp_path, world, name):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import test_slab as T
    from fluidnet_cxx_amd import simulate
    from fluidnet_cxx_amd.slab import SlabLayout
    case = CASES[name]
    D, H, W, halo, w, schedule, iters, ptol, mailbox, nsteps, _direct = case
    D = D // 2 * world if world != 2 else D
    case = (D,) + case[1:]
    port = _free_port()
    mp.spawn(_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    dev = torch.device("cuda:0")
    cfg = dict(T.CFG, jacobiIter=iters, pTol=ptol)
    gs = T.global_state(D, H, W, seed=7)
    bd = {k: torch.from_numpy(v).to(dev) for k, v in gs.items()}
    for _ in range(nsteps):
        simulate(cfg, bd, None, "jacobi")
    ref = {k: bd[k].cpu().numpy() for k in ("U", "density", "p")}
    for r in range(world):
        l = SlabLayout(D, world, r, halo)
        z = np.load(tmp_path / f"rank{r}.npz")
        for k in ("U", "density", "p"):
            a, b = z[k], ref[k][:, :, l.z_begin:l.z_begin + l.owned]
            bad = a.view(np.int32) != b.view(np.int32)
            assert not bad.any(), f"{name}, world {world}: {k} differs on {int(bad.sum())} owned cells of rank {r}"
        # (a time only where the ranks do not queue behind each other on the one GPU: eight processes take turns on its hardware
        #  queues, and a 1 MiB probe through 8 KiB slots is 128 launches that each wait for a neighbour's turn)
        assert 0.0 < float(z["probe_ms"]) < (50.0 if world <= 3 else 20000.0)


def _dead_peer_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from fluidnet_cxx_amd._ext import ext
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        comm = _peer_comm(ext, dist, rank, world, 1 << 16, timeout_s=1.0)
        msg = "ok"
        if rank == 0:                                  # rank 1 never calls: the wait must give up, not hang the GPU
            scratch = torch.zeros(1 << 16, dtype=torch.uint8, device=dev)
            try:
                ext.slab_comm_probe(comm, 4096, 1, scratch)       # warm-up exchange + 1: enqueued, time out on the device
                torch.cuda.synchronize()
                ext.slab_comm_probe(comm, 4096, 1, scratch)       # the next call reports it
                msg = "no error"
            except RuntimeError as e:
                msg = str(e)
        open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write(msg)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_peer_store_dead_neighbour_times_out(tmp_path):
    """a neighbour that never arrives: the device-side wait gives up after the timeout (1 s here) and the communicator's next call
    fails with FNX_ECOMM -- a spin must not outlive its peer on a shared GPU"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_dead_peer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    msg = open(tmp_path / "rank0.txt").read()
    assert "timed out" in msg, msg


def _killed_rank_worker(rank, world, port, victim, timeout_s, out_dir):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    import time
    import torch.distributed as dist
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout
    import test_slab as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    D, H, W, halo, w = 16 * world, 20, 70, 6, 6
    cfg = dict(T.CFG, jacobiIter=20)
    layout = SlabLayout(D, world, rank, halo)
    st = T.local_state(T.global_state(D, H, W, seed=7), layout, dev)
    comm = _peer_comm(ext, dist, rank, world, 1 << 20, timeout_s=timeout_s)
    sim = NativeSlabSimulator(layout, cfg, comm=comm, sweeps_per_exchange=w, static_flags=True, cfl_check_every=0, schedule="deep_beside")
    sim.step(st)
    torch.cuda.synchronize()
    dist.barrier()
    if rank == victim:
        os._exit(0)                                        # gone between two steps, without a word: no destructor, no abort call
    t0 = time.time()
    msg = "ok"
    try:
        for _ in range(3):                                 # 3 steps x 19-odd exchanges: every one of them would spin for the time-out
            sim.step(st)
        torch.cuda.synchronize()
        if comm.failed():
            msg = "failed() after the synchronisation"
    except RuntimeError as e:
        msg = "raised: " + str(e)[:120]
    torch.cuda.synchronize()
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write(f"{time.time() - t0:.3f}|{msg}")
    os._exit(0)                                            # (no process-group teardown with a rank missing)


def test_peer_store_rank_killed_between_steps(tmp_path):
    """A rank that dies between two steps (four processes on the one GPU, rank 2 exits without a word): its neighbours' device-side
    waits time out ONCE (1.5 s here) -- the first time-out raises the rank's own abort word, every exchange launch already queued or
    enqueued later leaves at once -- and the failure walks down the chain the same way.  Every survivor reports it (an exception from a
    later call, or `comm.failed()` after the synchronisation: a step that met the dead neighbour on the device returns normally), and
    none of them spins for (exchanges x time-out): three steps are ~60 exchanges = 90 s at one time-out each."""
    import torch.multiprocessing as mp
    world, victim, timeout_s = 4, 2, 1.5
    port = _free_port()
    ctx = mp.spawn(_killed_rank_worker, args=(world, port, victim, timeout_s, str(tmp_path)), nprocs=world, join=False)
    import time
    deadline = time.time() + 120
    while time.time() < deadline and not all((tmp_path / f"rank{r}.txt").exists() for r in range(world) if r != victim):
        time.sleep(0.5)
    for p in ctx.processes:
        p.join(timeout=10)
        if p.is_alive():
            p.kill()
    for r in range(world):
        if r == victim:
            continue
        f = tmp_path / f"rank{r}.txt"
        assert f.exists(), f"rank {r} never finished"
        secs, msg = f.read_text().split("|", 1)
        assert msg != "ok", f"rank {r} did not notice its dead neighbour"
        assert float(secs) < 12 * timeout_s, f"rank {r} spun for {secs} s: the abort word does not stop the queued exchanges ({msg})"


def _bench_leg_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    import json
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        from fluidnet_cxx_amd.slab import SlabLayout
        w = bench.WORKLOADS["plume3d_slab_jacobi"]
        m = bench.mconf_for(w)
        layout = SlabLayout(w["D"] * world, world, rank, halo=6)
        bd = bench.plume_state_torch(w["res"], layout.D_local, dev, layout.z_offset, layout.D_global)
        out, comm = bench.run_native_slab(4, 12, world, rank, dev, bd, m, w["res"], w["D"], "deep_beside", transport="peer")
        if rank == 0:
            json.dump(dict(out=out, comm=comm), open(os.path.join(out_dir, "leg.json"), "w"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_bench_peer_leg_two_processes_one_gpu(tmp_path):
    """bench.py's N > 1 leg over the peer-store transport (`run_native_slab(..., transport="peer")`: communicator probe, timed steps,
    the driver's statistics), rehearsed with two processes on the one GPU -- the timings mean nothing (two ranks share a device), the
    point is that the code path the driver's multi-GPU run takes has executed before: bytes per neighbour and step must be the
    schedule's (4 planes of U and density = 16, 5 of div, 16 x 6 + 1 of p: 118 planes of 1 MiB in 19 exchanges)."""
    import json
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_bench_leg_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = json.load(open(tmp_path / "leg.json"))
    assert r["out"]["state_finite"] and r["out"]["ms_per_step"] > 0 and "peer-store" in r["out"]["transport"]
    assert "graph" in r["out"]["launch"], r["out"]["launch"]        # the peer-store leg replays the captured step
    c = r["comm"]
    assert c["exchanges_per_step"] == 19 and c["bytes_per_neighbour_per_step"] == (4 * 4 + 5 + 16 * 6 + 1) * (1 << 20), c
    assert c["probe_6MiB_ms"] > 0 and c["wait_ms_per_step"] >= 0


def _graph_worker(rank, world, port, case, out_dir):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout
    import test_slab as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        D, H, W, halo, w, schedule, iters, nreplay = case
        cfg = dict(T.CFG, jacobiIter=iters)
        gs = T.global_state(D, H, W, seed=9)
        layout = SlabLayout(D, world, rank, halo)
        st = T.local_state(gs, layout, dev)
        comm = _peer_comm(ext, dist, rank, world, 1 << 20)
        sim = NativeSlabSimulator(layout, cfg, comm=comm, sweeps_per_exchange=w, static_flags=True, cfl_check_every=0, schedule=schedule)
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            for _ in range(2):                              # (the second step builds the BC class map the captured step reuses)
                sim.step(st)
            side.synchronize()
            dist.barrier()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                sim.step(st)                                # captured: 19-odd exchanges per step -- the mailbox slots' turn differs from replay to replay
            dist.barrier()
            for _ in range(nreplay):
                g.replay()
            side.synchronize()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **{k: st[k][:, :, layout.owned_slice].cpu().numpy() for k in ("U", "density", "p")})
        dist.barrier()
        del g, sim, comm
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("schedule,world", [("deep_beside", 2), ("deep_first", 2), ("deep_beside", 8)])
def test_peer_store_step_replays_as_hip_graph(tmp_path, schedule, world):
    """The C++ driver's step over the peer-store transport captured in a HIP graph on each of two processes and replayed: the chunk
    counters of the transport live on the device (and the direct sends' mailbox slot is chosen there), so no launch argument depends on
    how many exchanges came before -- 2 eager steps + 3 replays equal 5 single-domain steps bit for bit (an odd number of exchanges per
    step: every replay uses the other slot of each mailbox than the one before)."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import test_slab as T
    from fluidnet_cxx_amd import simulate
    from fluidnet_cxx_amd.slab import SlabLayout
    nreplay = 3
    case = (24 * world, 20, 70, 6, 6, schedule, 20, nreplay)
    port = _free_port()
    mp.spawn(_graph_worker, args=(world, port, case, str(tmp_path)), nprocs=world, join=True)
    dev = torch.device("cuda:0")
    D, H, W, halo = case[:4]
    cfg = dict(T.CFG, jacobiIter=case[6])
    gs = T.global_state(D, H, W, seed=9)
    bd = {k: torch.from_numpy(v).to(dev) for k, v in gs.items()}
    for _ in range(2 + nreplay):
        simulate(cfg, bd, None, "jacobi")
    ref = {k: bd[k].cpu().numpy() for k in ("U", "density", "p")}
    for r in range(world):
        l = SlabLayout(D, world, r, halo)
        z = np.load(tmp_path / f"rank{r}.npz")
        for k in ("U", "density", "p"):
            a, b = z[k], ref[k][:, :, l.z_begin:l.z_begin + l.owned]
            bad = a.view(np.int32) != b.view(np.int32)
            assert not bad.any(), f"{schedule}: {k} differs on {int(bad.sum())} owned cells of rank {r} after graph replays"


def _cnn_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from fluidnet_cxx_amd import FluidNet
    from fluidnet_cxx_amd._ext import ext
    from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    import test_slab as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        D, H, W, halo = 64 * world, 40, 72, 52
        gs = T.global_state(D, H, W, seed=6)
        gs["U"] = (gs["U"] * 0.4).astype(np.float32)
        net = FluidNet.from_weights(T.CNN_CFG, make_scalenet_weights(0, ndim=3), dev)
        layout = SlabLayout(D, world, rank, halo)
        st = T.local_state(gs, layout, dev)
        # 64 KiB mailbox slots: the 49 ghost planes of U (3 x 49 x 11 520 B) travel in chunks
        comm = _peer_comm(ext, dist, rank, world, 1 << 16)
        sim = NativeSlabSimulator(layout, T.CNN_CFG, comm=comm, static_flags=True, cfl_check_every=2, method="convnet", net=net)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            for n in range(2):
                sim.step(st)
                torch.cuda.current_stream().synchronize()
                np.savez(os.path.join(out_dir, f"rank{rank}_step{n}.npz"),
                         **{k: st[k][:, :, layout.owned_slice].cpu().numpy() for k in ("U", "density", "p")})
        dist.barrier()
        del sim, comm
    finally:
        dist.destroy_process_group()


def test_peer_store_convnet_projection_two_processes(tmp_path):
    """The CNN projection on z-slabs (fnx_slab_step, prm.method 1) over the peer-store transport, two processes on one GPU: the std's
    sums travel along the chain of ranks (the float all-reduce that is an exact all-gather), the 49 ghost planes of U in chunks through
    64-KiB mailbox slots, the one plane of p -- against the single-domain `simulate(..., 'convnet')`: p, U within 1e-5 of |ref|max, the
    first step's density bit for bit (lib/model.py:118-227, lib/simulate.py:96-168)."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import test_slab as T
    from fluidnet_cxx_amd import FluidNet, simulate
    from fluidnet_cxx_amd.slab import SlabLayout
    from fluidnet_cxx_amd.weights import make_scalenet_weights
    world = 2
    port = _free_port()
    mp.spawn(_cnn_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    dev = torch.device("cuda:0")
    D, H, W, halo = 64 * world, 40, 72, 52
    gs = T.global_state(D, H, W, seed=6)
    gs["U"] = (gs["U"] * 0.4).astype(np.float32)
    net = FluidNet.from_weights(T.CNN_CFG, make_scalenet_weights(0, ndim=3), dev)
    bd = {k: torch.from_numpy(v).to(dev) for k, v in gs.items()}
    for n in range(2):
        simulate(T.CNN_CFG, bd, net, "convnet")
        ref = {k: bd[k].cpu().numpy() for k in ("U", "density", "p")}
        for r in range(world):
            l = SlabLayout(D, world, r, halo)
            z = np.load(tmp_path / f"rank{r}_step{n}.npz")
            own = slice(l.z_begin, l.z_begin + l.owned)
            if n == 0:
                assert np.array_equal(z["density"].view(np.int32), ref["density"][:, :, own].view(np.int32)), f"density, rank {r}"
            for k in ("p", "U"):
                scale = float(np.abs(ref[k]).max())
                d = float(np.abs(z[k].astype(np.float64) - ref[k][:, :, own]).max())
                assert d <= 1e-5 * scale, f"step {n + 1}: {k} on rank {r}: max |d| = {d:.3e} > 1e-5 * {scale:.3e}"


def test_bench_eight_rank_rehearsal_on_one_gpu(tmp_path):
    """`bench.py --gpus 8 --rehearse-one-gpu`: the exact launcher path of the driver's multi-GPU run -- self-spawn under
    torch.distributed.run, eight ranks, the job watchdog, the fall-back when the Python driver's leg fails, the C++ driver over the
    peer-store transport as a replayed HIP graph, the communicator probe and statistics, ONE JSON line under 4 kB -- with all eight
    ranks on the one GPU (gloo process group; RCCL refuses two ranks on one device, so its legs are skipped and say so).  The timings mean
    nothing; the bytes per neighbour and step and the exchange count must be the schedule's."""
    import json
    import subprocess
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "2", "--rehearse-one-gpu", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    assert len(lines[0]) < 4096, len(lines[0])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["world_size"] == 8 and d["config"]["workload"] == "plume3d_slab_jacobi", d["config"]
    assert "REHEARSAL" in d["config"]["backend"]
    assert d["value"] and d["value"] > 0 and d["ms_per_step"] > 0
    nd = d["native_driver"]
    assert nd["state_finite"] and "peer-store" in nd["transport"] and "graph" in nd["launch"], nd
    assert "error" in d["python_driver"] and "error" in d["native_driver_rccl"], (d.get("python_driver"), d.get("native_driver_rccl"))
    c = d["comm"]
    want = (4 * 4 + 5 + 16 * 6 + 1) * (1 << 20)            # (the printed line rounds to four significant digits)
    assert c["exchanges_per_step"] == 19 and abs(c["bytes_per_neighbour_per_step"] - want) <= 1e-3 * want, c


def test_bench_fallback_line_when_a_rank_dies_in_a_supplementary_leg(tmp_path):
    """`bench.py` at N > 1 hands its line as it stands to a monitor process before every leg that has never met a multi-GPU box; when
    a rank is killed inside such a leg (here: the last rank SIGKILLs itself at the start of the peer-store leg, the launcher then
    terminates the others) there is still ONE JSON line on stdout -- the armed line from the monitor, marked `fallback`, or the job's
    final line through the monitor if rank 0 outlived the launcher's SIGTERM long enough to finish -- instead of none or two."""
    import json
    import subprocess
    env = dict(os.environ, PYTHONPATH=REPO, FNX_BENCH_TEST_CRASH="peer")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse-one-gpu", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    d = json.loads(lines[0])
    # two endings, both ONE line: the launcher's SIGTERM reached rank 0 first (the monitor printed the armed line, marked `fallback`), or
    # rank 0's own leg noticed the dead rank first, failed, and the job's final line went out with that leg's error in it
    if "fallback" in d:
        assert "peer-store" in d["fallback"], d["fallback"]
    else:
        assert "error" in d.get("native_driver_peer", d.get("native_driver", {})), d
    assert d["n_gpus"] == 2 and d["config"]["workload"] == "plume3d_slab_jacobi"
