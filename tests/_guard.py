"""FVK_GUARD_ALLOC=1: every device allocation of this process comes from the guard-page allocator (scripts/probes/guard_alloc.cpp: each tensor
ends where its mapping ends, an unmapped granule behind it), so that an out-of-bounds access of ANY kernel is a GPU page fault instead of a
silent read of a neighbour.  Installed at import time — before the first device allocation — by tests/conftest.py (the pytest process) and by
the test modules whose worker functions run in spawned children.  A measurement device; never set in the normal suite."""
import os


def maybe_install():
    if os.environ.get("FVK_GUARD_ALLOC") != "1" or os.environ.get("_FVK_GUARD_INSTALLED") == str(os.getpid()):
        return False
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "probes", "libguard_alloc.so")
    if not os.path.exists(so):
        raise RuntimeError(f"FVK_GUARD_ALLOC=1 but {so} is not built (hipcc -O2 -shared -fPIC scripts/probes/guard_alloc.cpp -o {so})")
    import torch
    alloc = torch.cuda.memory.CUDAPluggableAllocator(so, "guard_malloc", "guard_free")
    torch.cuda.memory.change_current_allocator(alloc)
    os.environ["_FVK_GUARD_INSTALLED"] = str(os.getpid())
    return True
