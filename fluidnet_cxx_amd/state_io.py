"""Restart files in the reference drivers' format (plume.py:168-175 reads, :423-424 writes):
`torch.save({'batch_dict': batch_dict, 'it': it}, file)`, so a simulation started under either backend can be
continued under the other.  Tensors are written from / read onto whatever device the caller names."""
import torch

_FIELDS = ("p", "U", "flags", "density", "UBC", "UBCInvMask", "densityBC", "densityBCInvMask", "flags_stick")


def save_restart(path, batch_dict, it):
    """Write the checkpoint the reference's `restart_sim` branch loads (tensors stored on the CPU)."""
    cpu = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in batch_dict.items()}
    torch.save({"batch_dict": cpu, "it": int(it)}, path)


def load_restart(path, device, trusted=False):
    """Returns (batch_dict, it) with every tensor contiguous fp32 on `device`; accepts files written by the reference
    (tensors pickled as CUDA tensors are mapped) and by save_restart.  The payload is tensors plus an int, so the file
    is read with `weights_only=True` (no arbitrary unpickling); `trusted=True` opts into the full unpickler for files
    that carry other Python objects."""
    blob = torch.load(path, map_location="cpu", weights_only=not trusted)
    assert isinstance(blob, dict) and "batch_dict" in blob and "it" in blob, "not a fluidnet restart file"
    bd = {}
    for k, v in blob["batch_dict"].items():
        if torch.is_tensor(v):
            v = v.to(device=device, dtype=torch.float32 if v.is_floating_point() or k in _FIELDS else v.dtype).contiguous()
        bd[k] = v
    for k in ("p", "U", "flags"):
        assert k in bd and bd[k].dim() == 5, f"restart file lacks a 5-D '{k}'"
    return bd, int(blob["it"])


def rollout(mconf, batch_dict, net, sim_method, steps, workspace=None):
    """`steps` consecutive `simulate` calls without autograd (the long-term loop of fluid_net_train.py:349-373 and of
    the drivers); any batch size.  flags are taken as fixed over the rollout."""
    from ._simulate import simulate
    with torch.no_grad():
        for n in range(int(steps)):
            simulate(mconf, batch_dict, net, sim_method, workspace=workspace, static_flags=workspace is not None and n > 0)
    return batch_dict
