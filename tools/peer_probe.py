"""GPU-box probe: latency / rate of one ghost exchange through the peer-store communicator between TWO PROCESSES on one GPU (the
mapped mailboxes are local HBM here: what is measured is the launch + flag round trip + two device copies, not a link), next to
the in-process loopback communicator's event-ordered copies.  usage: peer_probe.py [reps=50]"""
import os, socket, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
SIZES = [4096, 65536, 1 << 20, 4 << 20, 6 << 20, 16 << 20]


def worker(rank, world, port, reps):
    import torch.distributed as dist
    from fluidnet_cxx_amd._ext import ext
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    peer = ext.SlabPeer(rank, world, 16 << 20)
    hs = [None] * world
    dist.all_gather_object(hs, peer.handle)
    comm = ext.slab_comm_peer(peer, hs[rank - 1] if rank > 0 else None, hs[rank + 1] if rank < world - 1 else None)
    scratch = torch.zeros(4 * SIZES[-1], dtype=torch.uint8, device=dev)
    for n in SIZES:
        dist.barrier()
        ms = ext.slab_comm_probe(comm, n, reps, scratch)
        if rank == 0:
            print(f"peer-store, {world} processes on one GPU: {n:>9d} B per neighbour and direction: {ms * 1e3:8.1f} us per exchange  ({n / ms / 1e6:7.1f} GB/s per direction)", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    for world in (2, 3):
        mp.spawn(worker, args=(world, port + world, reps), nprocs=world, join=True)
