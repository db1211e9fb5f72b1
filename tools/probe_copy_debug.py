"""Debug / demonstration of icv_probe_copy_path's controls in one process (run on the GPU box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd import native
lib = native.lib()
n = 8 << 20
dev = "cuda:0"
src_dev = torch.randint(0, 255, (n,), dtype=torch.uint8, device=dev)
src_pin = torch.randint(0, 255, (n,), dtype=torch.uint8).pin_memory()
dst = torch.zeros((n,), dtype=torch.uint8, device=dev)
dst_pin = torch.zeros((n,), dtype=torch.uint8).pin_memory()
for name, s, d in (("device->device", src_dev, dst), ("pinned host->device", src_pin, dst), ("device->pinned host", src_dev, dst_pin)):
    out = []
    for _ in range(5):
        kind, ms = ctypes.c_int(-1), ctypes.c_double(0.0)
        native.check(lib.icv_probe_copy_path(s.data_ptr(), d.data_ptr(), n, ctypes.byref(kind), ctypes.byref(ms)), "probe")
        out.append((kind.value, round(ms.value, 3)))
    print(name, out, "(kind 1 = copy engine, 2 = blit kernel, 0 = inconclusive; ms = the copy's own duration)", flush=True)
