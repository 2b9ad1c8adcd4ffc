"""`fmoe`-shaped API surface backed by libm3vit_hip.so.

The reference imports these names from the third-party fastmoe package (absent from the
reference tree): FMoE, _fmoe_general_global_forward (custom_moe_layer.py:6), FMoELinear (:7),
prepare_forward / ensure_comm / MOEScatter / MOEGather / AllGather / Slice (:13-15),
NaiveGate (:16), BaseGate (noisy_gate_vmoe.py:4), DistributedGroupedDataParallel
(train_fastmoe.py:460).  `m3vit_amd.install_fmoe_shim()` registers this package as `fmoe` in
sys.modules so that an unmodified reference checkout imports it.
"""
from .layers import FMoE, _fmoe_general_global_forward, mark_module_parallel_comm  # noqa: F401
from .linear import FMoELinear  # noqa: F401
from .distributed import DistributedGroupedDataParallel  # noqa: F401
from . import functions, gates, layers, linear  # noqa: F401
