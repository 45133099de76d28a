"""pt_comm_* (RCCL behind the C ABI) at world size 1 on GPU 0, in a process of its own:
  python tools/comm_probe.py [none|first|after]
none: no torch in the process (the library on /opt/rocm's HIP runtime and RCCL); first: torch is imported BEFORE the library
is loaded (bench.py's order: the library binds to torch's bundled HIP runtime by soname and must find torch's RCCL next to
it); after: the library first, torch afterwards (two HIP runtimes mapped; the library must keep to /opt/rocm's RCCL).
Renders a small frame, gathers it through ncclAllGather + the un-permute kernel, compares.  Prints "gather ok"."""
import ctypes as C
import os
import sys

order = sys.argv[1] if len(sys.argv) > 1 else "none"
if order == "first":
    import torch  # noqa: F401
    torch.cuda.device_count()

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ptlib

L = ptlib.product()
assert L.pt_device_count() >= 1, "needs a GPU"
ctx = C.c_void_p()
assert L.pt_ctx_create(0, C.byref(ctx)) == 0, L.pt_last_error()
if order == "after":
    import torch  # noqa: F401
    torch.cuda.device_count()
import test_gpu_parity

test_gpu_parity._comm_gather_world_1(L, ctx)
L.pt_ctx_destroy(ctx)
print("gather ok: ncclAllGather at world size 1, torch %s" % order)
