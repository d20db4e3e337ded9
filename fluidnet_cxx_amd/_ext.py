"""Loader of the native extension `fluidnet_cpp` (built in-tree by fluidnet_cxx_amd/build.py).

The product path has no CPU or PyTorch fallback: if the extension or libfluidnet_hip.so is missing this
module raises, and every operator raises on non-GPU tensors.
"""
import importlib
import os
import sys

import torch  # noqa: F401  (must be imported first: the extension resolves libamdhip64 / libtorch through it)

_HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    if "fluidnet_cpp" in sys.modules:
        return sys.modules["fluidnet_cpp"]
    so = os.path.join(_HERE, "fluidnet_cpp.so")
    lib = os.path.join(_HERE, "libfluidnet_hip.so")
    if not (os.path.exists(so) and os.path.exists(lib)):
        from . import build
        build.build_all()
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)
    try:
        return importlib.import_module("fluidnet_cpp")
    except ImportError as e:  # fail loudly: there is no fallback
        raise ImportError(f"fluidnet_cxx_amd: native extension failed to load ({e}); run "
                          f"`python -c 'import __graft_entry__ as g; g.build()'`") from e


ext = load()
