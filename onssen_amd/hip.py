"""Loader of the product library libonssen_hip.so (built in-tree by
``__graft_entry__.build()``).  There is no fallback: if the library is
missing or does not export the full ABI, importing a kernel raises."""
import os

from ._abi import Lib, OnssenError

_LIB = None
# ONSSEN_HIP_LIB: a profiling / A-B build of the same source (tools/ab_variants.py), never a different implementation
LIB_PATH = os.environ.get("ONSSEN_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libonssen_hip.so")


def get_lib():
    global _LIB
    if _LIB is None:
        # PyTorch-ROCm ships its own libamdhip64 (same SONAME as /opt/rocm's).  Load torch first so
        # that this process has ONE HIP runtime and our kernels launch on the streams torch hands us;
        # dlopen-ing our library first would pull in a second runtime ("no ROCm-capable device").
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise OnssenError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                "onssen_amd has no CPU or PyTorch-op fallback for its hot path.")
        _LIB = Lib(LIB_PATH)
    return _LIB
