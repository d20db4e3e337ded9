"""GPU-box probe: step time of a MIDDLE rank of the z-slab decomposition through the C++ driver (fnx_slab_step), on one GPU,
as a function of an assumed interconnect -- tools/slab_overlap_model.py for the native driver, without Python between launches.

One 512 x 512 x (64 + 2*6) slab (rank 1 of 3: the per-GPU shape of bench.py --gpus 3..8) is stepped with the link-model
communicator of the C ABI (fnx_slab_comm_link_model): every ghost exchange occupies the communication stream for
latency + bytes / bandwidth and then fills the ghost planes from the slab's own edge planes.  Reported per schedule and link:
ms per step, eager and as a HIP-graph replay, next to the ghost-free single slab (rank 0 of 1).
usage: slab_native_model.py [schedules=deep_first,deep_beside] [w=6] [halo=max(6, w)]      env MODEL_LINKS="0:0,20:75" MODEL_STEPS=20 MODEL_DIRECT=auto|never|always"""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from fluidnet_cxx_amd._ext import ext
from fluidnet_cxx_amd.slab import NativeSlabSimulator, SlabLayout

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)


def timed(step, n):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    schedules = (sys.argv[1] if len(sys.argv) > 1 else "deep_first,deep_beside").split(",")
    wsw = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    halo = int(sys.argv[3]) if len(sys.argv) > 3 else max(6, wsw)
    n = int(os.environ.get("MODEL_STEPS", 20))
    w = bench.WORKLOADS["plume3d_slab_jacobi"]; m = bench.mconf_for(w)
    cfgs = ((0, 0), (15, 150), (20, 75), (25, 55), (30, 40))
    if os.environ.get("MODEL_LINKS"):
        cfgs = tuple(tuple(int(v) for v in c.split(":")) for c in os.environ["MODEL_LINKS"].split(","))
    # the ghost-free slab
    l1 = SlabLayout(64, 1, 0, halo)
    st = bench.plume_state_torch(512, l1.D_local, dev, 0, 64)
    sim = NativeSlabSimulator(l1, m, comm=None, sweeps_per_exchange=wsw, static_flags=True, cfl_check_every=0)
    for _ in range(30):
        sim.step(st)
    base = timed(lambda: sim.step(st), n)
    print(f"ghost-free slab (1 rank): {base:.3f} ms/step", flush=True)
    layout = SlabLayout(64 * 3, 3, 1, halo)
    for schedule in schedules:
        for lat, gbps in cfgs:
            st = bench.plume_state_torch(512, layout.D_local, dev, layout.z_offset, layout.D_global)
            comm = ext.slab_comm_link_model(float(lat), float(gbps))
            direct = os.environ.get("MODEL_DIRECT", "auto")      # direct sends from the last edge part of a block: auto | never | always
            sim = NativeSlabSimulator(layout, m, comm=comm, sweeps_per_exchange=wsw, static_flags=True, cfl_check_every=0, schedule=schedule,
                                      direct_sends=direct)
            for _ in range(10):
                sim.step(st)
            eager = timed(lambda: sim.step(st), n)
            graph = None
            # (hipStreamEndCapture of the three-stream deep_beside step segfaults inside the HIP runtime of ROCm 7.0.2: eager only)
            try:
                # (MODEL_GRAPH=force tries it anyway: since the exchange rides on the edge stream the step has one stream less)
                if (schedule == "deep_beside" and os.environ.get("MODEL_GRAPH") != "force") or os.environ.get("MODEL_GRAPH", "1") == "0":
                    raise RuntimeError("skipped")
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    sim.step(st)
                graph = timed(g.replay, n)
            except Exception as e:  # noqa: BLE001
                if str(e) != "skipped":
                    sys.stderr.write(f"graph capture failed: {e}\n")
            gtxt = f", graph replay {graph:.3f}" if graph is not None else ""
            best = min(eager, graph) if graph is not None else eager
            print(f"{schedule} (direct sends: {direct}) w={wsw} halo={halo}: link {gbps:4d} GB/s + {lat:2d} us -> eager {eager:.3f} ms/step{gtxt}   "
                  f"(ghost-free / middle = {base / best * 100:.1f} %)", flush=True)
            del sim, comm


main()
