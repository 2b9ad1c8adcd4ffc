"""CPU oracle for the M3ViT MoE-ViT hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``m3vit_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and there only as the checker / reported baseline.

Parity pinning: the reference ships no tests, fixtures or
golden vectors for this path and its production layer needs the un-vendored
``fmoe`` package, so the oracle is pinned against outputs of the reference's own
importable pure-torch twins (``models/moe/gates.py``, ``models/moe/moe.py``,
``models/moe/parallel_experts.py``) generated in the build container by
``tests/gen_golden.py`` and committed under ``tests/golden/``.  Since round 5 the
attention / dense block / dense backbone / balance-helper arithmetic is pinned to
the reference's own backbone files too (``models/moe/ckpt/vision_transformer_moe.py``,
``models/backbones/vit.py``; fixtures g8 - g10): their imports of ``cv2`` / ``timm`` /
``tree`` are satisfied by import-line-only placeholder modules whose every attribute
raises when used (``tests/gen_golden.py``), ``fmoe`` by this repository's shim.  Only
the MoE block / MoE backbone AS A WHOLE stay pinned through their parts: running them
needs fastmoe's CUDA kernels.

The functions are device-agnostic (round 5): ``tests/amp_error_table.py`` and
``tests/test_full_size.py`` run them on the GPU under ``torch.autocast`` as the stand-in
for the reference's AMP arithmetic - still test-side only.
"""
from .ref_torch import *  # noqa: F401,F403
