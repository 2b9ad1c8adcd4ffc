"""CPU oracle for the M3ViT MoE-ViT hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``m3vit_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and there only as the checker / reported baseline.

Parity pinning: the reference ships no tests, fixtures or
golden vectors for this path and its production layer needs the un-vendored
``fmoe`` package, so the oracle is pinned against outputs of the reference's own
importable pure-torch twins (``models/moe/gates.py``, ``models/moe/moe.py``,
``models/moe/parallel_experts.py``) generated in the build container by
``tests/gen_golden.py`` and committed under ``tests/golden/``.  The attention /
LayerNorm / block arithmetic cannot be pinned to reference code (its backbone
files do not import here: ``cv2``/``timm``/``fmoe`` missing) and is pinned to the
torch ops the reference calls instead.
"""
from .ref_torch import *  # noqa: F401,F403
