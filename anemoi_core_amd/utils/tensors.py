"""Cache-key helper: identity of a tensor's contents for the static-graph caches."""
from torch import Tensor


def version(t: Tensor) -> int:
    """In-place modification counter; inference tensors do not track one (they are immutable outside inference mode)."""
    return -1 if t.is_inference() else t._version
