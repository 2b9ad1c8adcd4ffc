"""m3vit_amd: MI355X (gfx950) implementation of the M3ViT MoE-ViT forward/backward hot path.

The compute lives in libm3vit_hip.so (hand-written HIP, C ABI in include/m3vit_hip.h);
this package is the host-side mirror of the reference's `fmoe`-shaped layer API.
"""
__version__ = "0.1.0"
