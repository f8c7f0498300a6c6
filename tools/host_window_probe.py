"""Can the CPU write device memory directly (VRAM through the PCIe aperture), how fast, and does the device see the bytes?
usage: python tools/host_window_probe.py <hipExtMallocWithFlags flags: 0 default, 1 fine-grained, 3 uncached>   (one process per kind: a fault would end it)"""
import ctypes, time, sys, os
import numpy as np
import torch
hip = ctypes.CDLL("libamdhip64.so")
flags = int(sys.argv[1])
x = torch.zeros(4, device="cuda")  # context
p = ctypes.c_void_p()
rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(8 << 20), ctypes.c_uint(flags))
print(flags, "hipExtMallocWithFlags rc", rc, hex(p.value or 0), flush=True)
if rc != 0:
    sys.exit(0)
src = np.arange(1 << 21, dtype=np.float32)
print(flags, "writing from the CPU ...", flush=True)
ctypes.memmove(p.value, src.ctypes.data, 4096)
print(flags, "CPU write did not fault", flush=True)
for nbytes in (3424, 34240, 109568, 1753088, 3506176):
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); ctypes.memmove(p.value, src.ctypes.data, nbytes); ts.append(time.perf_counter() - t0)
    print(flags, "CPU write of %d bytes: median %.2f us" % (nbytes, 1e6 * np.median(ts)), flush=True)
back = np.zeros(1 << 21, dtype=np.float32)
hip.hipMemcpy(ctypes.c_void_p(back.ctypes.data), p, ctypes.c_size_t(109568), ctypes.c_int(2))
print(flags, "device sees the CPU's bytes:", bool((back[:109568 // 4] == src[:109568 // 4]).all()), flush=True)
ts = []
for _ in range(50):
    t0 = time.perf_counter(); ctypes.memmove(back.ctypes.data, p.value, 3424); ts.append(time.perf_counter() - t0)
print(flags, "CPU read of 3424 bytes: median %.2f us" % (1e6 * np.median(ts)), flush=True)

# the same bytes through the copy engine, from pageable and from pinned host memory (what Tensor.to(device) does)
t_dev = torch.empty(1753088 // 4, device="cuda")
for name, host in (("pageable", torch.from_numpy(src[:1753088 // 4].copy())), ("pinned", torch.from_numpy(src[:1753088 // 4].copy()).pin_memory())):
    ts = []
    for _ in range(50):
        torch.cuda.synchronize(); t0 = time.perf_counter(); t_dev.copy_(host, non_blocking=False); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(flags, "copy engine, %s host memory, 1753088 bytes: median %.2f us" % (name, 1e6 * np.median(ts)), flush=True)
