"""m3vit_amd: MI355X (gfx950) implementation of the M3ViT MoE-ViT forward/backward hot path.

The compute lives in libm3vit_hip.so (hand-written HIP, C ABI in include/m3vit_hip.h);
this package is the host-side mirror of the reference's `fmoe`-shaped layer API.
"""
__version__ = "0.1.0"


def install_fmoe_shim():
    """Register m3vit_amd.fmoe as the top-level package `fmoe` (and its submodules) so that an
    unmodified reference checkout (`from fmoe.layers import FMoE, _fmoe_general_global_forward`, ...)
    binds to the HIP implementation.  Refuses to shadow a real fastmoe install."""
    import importlib
    import sys
    if "fmoe" in sys.modules and not getattr(sys.modules["fmoe"], "__m3vit_shim__", False):
        raise RuntimeError("a different `fmoe` package is already imported")
    pkg = importlib.import_module("m3vit_amd.fmoe")
    pkg.__m3vit_shim__ = True
    sys.modules["fmoe"] = pkg
    for sub in ("layers", "linear", "functions", "gates", "gates.base_gate", "gates.naive_gate", "distributed"):
        sys.modules["fmoe." + sub] = importlib.import_module("m3vit_amd.fmoe." + sub)
    return pkg
