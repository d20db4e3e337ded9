"""fluidnet_cxx_amd: MI355X-native fluid time-step behind fluidnet_cxx's operator surface.

    from fluidnet_cxx_amd import fluid, simulate, FluidNet     # mirrors the reference's `lib`

Importing the operator modules loads the native extension; there is no CPU fallback.
"""


def __getattr__(name):
    import importlib
    if name == "fluid":
        return importlib.import_module(".fluid", __name__)
    if name == "simulate":
        return importlib.import_module("._simulate", __name__).simulate
    if name in ("save_restart", "load_restart", "rollout"):
        return getattr(importlib.import_module(".state_io", __name__), name)
    if name == "output":
        return importlib.import_module(".output", __name__)
    if name in ("FluidNet", "MultiScaleNet"):
        return getattr(importlib.import_module(".model", __name__), name)
    raise AttributeError(name)
