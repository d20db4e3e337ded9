"""Deterministic, platform-independent synthetic weights for the MultiScale pressure net.

The reference's pretrained blob is not in its tree (`.MISSING_LARGE_BLOBS`), so every
CNN number in this repo is quoted on random-init weights of the reference architecture
(`pytorch/lib/multi_scale_net.py:111-116`).  The values come from a splitmix64 integer hash
rather than from a framework RNG, so golden vectors, tests and the benchmark regenerate exactly
the same tensors on any machine.

Layout: a dict name -> float32 array with torch's own parameter names and shapes
(`multiScale.convN_4.encode.0.weight`: (Cout, Cin, k, k) ...), bound = 1/sqrt(fan_in) like
torch's default Conv init.
"""
import numpy as np

# (tower, sequential index inside `encode`, Cin, Cout, k, relu_after)
# reference multi_scale_net.py:21-116 (Dropout layers are identity in eval and shift the indices:
# the plume driver builds the net with dropout=False? no -- MultiScaleNet always builds its towers with
# dropout=True, so the last conv sits one index further).
def scalenet_layers(in_dims=2, ndim=2):
    """Layer table of MultiScaleNet(in_dims). `ndim`=3 gives the Conv3d analogue."""
    t4 = [(in_dims, 32, 3, True), (32, 64, 3, True), (64, 32, 3, False), (32, 1, 3, False)]
    t2 = [(in_dims + 1, 32, 5, True), (32, 64, 3, True), (64, 128, 3, True), (128, 64, 3, True),
          (64, 32, 3, False), (32, 1, 3, False)]
    t1 = [(in_dims + 1, 32, 5, True), (32, 64, 3, True), (64, 128, 3, True), (128, 64, 3, True),
          (64, 32, 3, False), (32, 8, 5, False)]
    # indices of the Conv modules inside nn.Sequential (ReLU and Dropout occupy slots)
    idx4 = [0, 2, 4, 6]
    idx2 = [0, 2, 4, 6, 8, 10]
    out = []
    for name, tower, idx in (("convN_4", t4, idx4), ("convN_2", t2, idx2), ("convN_1", t1, idx2)):
        for (cin, cout, k, relu), i in zip(tower, idx):
            out.append(dict(name=f"multiScale.{name}.encode.{i}", cin=cin, cout=cout, k=k, relu=relu,
                            tower=name))
    out.append(dict(name="multiScale.final", cin=8, cout=1, k=1, relu=False, tower="final"))
    return out


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_uniform(n, stream, seed=0):
    """n float32 values in [-1, 1), reproducible everywhere (pure uint64 arithmetic)."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream) * np.uint64(0x9E3779B1)
        idx = np.arange(n, dtype=np.uint64) + (base << np.uint64(20))
        z = _splitmix64(idx)
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)      # [0,1) with 24 bits
    return (2.0 * u - 1.0).astype(np.float32)


def make_scalenet_weights(seed=0, in_dims=2, ndim=2, gain=1.0):
    """name -> np.float32 array for every Conv weight/bias of MultiScaleNet."""
    w = {}
    for li, L in enumerate(scalenet_layers(in_dims, ndim)):
        kshape = (L["k"],) * ndim
        fan_in = L["cin"] * int(np.prod(kshape))
        bound = gain / np.sqrt(fan_in)
        nw = L["cout"] * fan_in
        w[L["name"] + ".weight"] = (hash_uniform(nw, 2 * li, seed) * np.float32(bound)).reshape(
            (L["cout"], L["cin"]) + kshape)
        w[L["name"] + ".bias"] = hash_uniform(L["cout"], 2 * li + 1, seed) * np.float32(bound)
    return w
