"""What does hipStreamWaitValue32 run as on this runtime - a command-processor packet or a spinning kernel?  (csrc/ipc.hip relies on it.)
Stream A waits for a host-pinned flag >= 1, a copy follows it; 50 ms later stream B writes 1.  Run under `rocprofv3 --kernel-trace --stats`:
a `__amd_rocclr_streamOpsWait` kernel in the trace = a wave spins for the whole wait."""
import ctypes, time
import torch
hip = ctypes.CDLL("libamdhip64.so")
flag = torch.zeros((16,), dtype=torch.int32).pin_memory()
a, b = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros((1 << 20,), device="cuda"); y = torch.empty_like(x)
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
dp = ctypes.c_void_p()
assert hip.hipHostGetDevicePointer(ctypes.byref(dp), ctypes.c_void_p(flag.data_ptr()), 0) == 0
for rep in range(3):
    flag.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = hip.hipStreamWaitValue32(ctypes.c_void_p(a.cuda_stream), dp, rep + 1, 0, 0xFFFFFFFF)          # flags 0 = hipStreamWaitValueGte
    assert rc == 0, rc
    with torch.cuda.stream(a):
        y.copy_(x, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(a)
    time.sleep(0.05)
    done_early = ev.query()
    assert hip.hipStreamWriteValue32(ctypes.c_void_p(b.cuda_stream), dp, rep + 1, 0) == 0
    a.synchronize()
    print(f"rep {rep}: copy behind the wait finished early: {done_early}; whole sequence {1e3 * (time.perf_counter() - t0):.1f} ms")
