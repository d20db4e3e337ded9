"""GPU-box probe: step time of a MIDDLE rank of the z-slab decomposition as a function of the interconnect, on one GPU.

One 512 x 512 x (64 + 2*6) slab (the per-GPU shape of bench.py --gpus 3..8) is advanced by the slab driver with a
stand-in communicator: every ghost exchange runs on its own stream, which waits for the main stream where the exchange is
posted (as RCCL's does), spins for  latency + bytes / bandwidth  and then fills the ghost planes (from the slab's own
edge planes: the physics is a periodic stack of this slab, the launch sequence and sizes are the real ones); the main
stream waits for it where the driver waits.  This measures how much of a transfer each Jacobi schedule hides, for assumed
link rates -- the multi-GPU box itself is only available to the round-end driver.
usage: slab_overlap_model.py [schedule=deep_first|edge_first|last_pass] [w=6]"""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import bench
from fluidnet_cxx_amd.slab import SlabLayout, SlabSimulator

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)


def calibrate():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000000); torch.cuda.synchronize()
    e0.record(); torch.cuda._sleep(20000000); e1.record(); torch.cuda.synchronize()
    return 20000000 / (e0.elapsed_time(e1) * 1e3)          # spin cycles per microsecond


class ModelComm:
    def __init__(self, layout, cyc_per_us, latency_us, gbps):
        self.l, self.cyc, self.lat, self.gbps = layout, cyc_per_us, latency_us, gbps
        self.stream = torch.cuda.Stream(device=dev)
        self.us_total = 0.0

    def start(self, fields, width, sources=None):
        l = self.l
        main = torch.cuda.current_stream(dev)
        self.stream.wait_stream(main)
        nbytes = sum(f[:, :, :width].numel() * 4 for f in fields)           # per direction and neighbour
        us = (self.lat + nbytes / (self.gbps * 1e3)) if self.gbps > 0 else 0.0
        self.us_total += us
        with torch.cuda.stream(self.stream):
            if us > 0:
                torch.cuda._sleep(int(us * self.cyc))
            for f, s in zip(fields, sources if sources is not None else fields):
                top = l.lo + l.owned
                f[:, :, l.lo - width:l.lo].copy_(s[:, :, top - width:top])      # what the lower neighbour would send
                f[:, :, top:top + width].copy_(s[:, :, l.lo:l.lo + width])
        return True

    def finish(self, handle):
        if handle:
            torch.cuda.current_stream(dev).wait_stream(self.stream)

    def exchange(self, fields, width):
        self.finish(self.start(fields, width))


def main():
    schedule = sys.argv[1] if len(sys.argv) > 1 else "deep_first"
    wsw = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    w = bench.WORKLOADS["plume3d_slab_jacobi"]; m = bench.mconf_for(w)
    layout = SlabLayout(64 * 3, 3, 1, 6)                                     # the middle one of three ranks
    cyc = calibrate()
    cfgs = ((0, 0), (15, 150), (20, 75), (25, 55), (30, 40))
    if os.environ.get("MODEL_LINKS"):                                       # e.g. "0:0,20:75"  (latency_us:GB/s)
        cfgs = tuple(tuple(int(v) for v in c.split(":")) for c in os.environ["MODEL_LINKS"].split(","))
    for lat, gbps in cfgs:
        st = bench.plume_state_torch(512, layout.D_local, dev, layout.z_offset, layout.D_global)
        sim = SlabSimulator(layout, m, sweeps_per_exchange=wsw, schedule=schedule, static_flags=True, cfl_check_every=0)
        sim.comm = ModelComm(layout, cyc, lat, gbps)
        for _ in range(3):
            sim.step(st)
        torch.cuda.synchronize()
        sim.comm.us_total = 0.0
        n = int(os.environ.get("MODEL_STEPS", 10))
        t0 = time.perf_counter()
        for _ in range(n):
            sim.step(st)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{schedule} w={wsw}: link {gbps:4d} GB/s + {lat:2d} us -> {ms:.3f} ms/step "
              f"(transfers posted: {sim.comm.us_total / n / 1e3:.3f} ms per step)", flush=True)


main()
