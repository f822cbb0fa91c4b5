"""Teardown diagnostics, Python form (no compiled code): create a CU-masked HIP stream through ctypes, run a fill on it, leak or
destroy it, optionally with PyTorch's own bundled HIP/HSA runtime initialised in the same process (bench.py imports torch for the
gloo barrier and torch.cuda.synchronize, so two HSA runtimes live in that process).
usage: masked_stream_py.py <leak|destroy|plainleak> <torch|notorch>"""
import ctypes as C
import sys

mode, with_torch = sys.argv[1], sys.argv[2] == "torch"
if with_torch:
    import torch
    torch.cuda.synchronize(0)
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
s = C.c_void_p()
d = C.c_void_p()
assert hip.hipMalloc(C.byref(d), 1 << 20) == 0
if mode == "plainleak":
    assert hip.hipStreamCreateWithPriority(C.byref(s), 0, 0) == 0
else:
    mask = (C.c_uint32 * 8)(*([0xFFFFFFFF] * 7 + [0]))
    assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, mask) == 0
for _ in range(20):
    assert hip.hipMemsetAsync(d, 1, 1 << 20, s) == 0
assert hip.hipStreamSynchronize(s) == 0
if mode == "destroy":
    assert hip.hipStreamDestroy(s) == 0
print(f"[{mode} {'torch' if with_torch else 'notorch'}] done, leaving the interpreter", file=sys.stderr, flush=True)
