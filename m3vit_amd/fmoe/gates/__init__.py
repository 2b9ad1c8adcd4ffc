from .base_gate import BaseGate  # noqa: F401
from .naive_gate import NaiveGate  # noqa: F401
