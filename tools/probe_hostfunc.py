"""Can a stream wait for a flag WITHOUT a wave on this runtime?  hipLaunchHostFunc with a host function that polls the flag
(tools/kv_occupy.hip: occ_host_wait) instead of hipStreamWaitValue32 (a spinning kernel here: profiles/r05/stream_ops_probe.txt).
  A  latency: flag set by the host -> the copy queued behind the wait has finished
  B  do host functions of different streams run concurrently (stream 2's flag is set while stream 1's function still polls)?
  C  does a polling host function stall an unrelated stream's completion (memset + synchronize on a third stream)?
  D  same measurement with hipStreamWaitValue32 for comparison
Run on the GPU box: python tools/probe_hostfunc.py  (under rocprofv3 --kernel-trace to see that no wait kernel appears)."""
import ctypes, os, subprocess, sys, time
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libkvoccupy.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "kv_occupy.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-w", os.path.join(here, "kv_occupy.hip"), "-o", so], check=True)
occ = ctypes.CDLL(so)
occ.occ_host_wait.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
hip = ctypes.CDLL("libamdhip64.so")
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
flags = torch.zeros((64,), dtype=torch.int32).pin_memory()
dp = ctypes.c_void_p()
assert hip.hipHostGetDevicePointer(ctypes.byref(dp), ctypes.c_void_p(flags.data_ptr()), 0) == 0
x = torch.zeros((1 << 18,), device="cuda"); y = torch.empty_like(x)
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()


def wait(kind, stream, word, value):
    if kind == "hostfunc":
        assert occ.occ_host_wait(ctypes.c_void_p(stream.cuda_stream), ctypes.c_void_p(flags.data_ptr() + 4 * word), value, None) == 0
    else:
        assert hip.hipStreamWaitValue32(ctypes.c_void_p(stream.cuda_stream), ctypes.c_void_p(dp.value + 4 * word), value, 0, 0xFFFFFFFF) == 0


def until(ev, limit=5.0):
    t0 = time.perf_counter()
    while not ev.query():
        if time.perf_counter() - t0 > limit:
            return None
    return time.perf_counter() - t0


for kind in ("hostfunc", "waitvalue"):
    lat = []
    for rep in range(20):
        flags.zero_()
        wait(kind, s1, 0, 1)
        with torch.cuda.stream(s1):
            y.copy_(x, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(s1)
        time.sleep(0.01)
        assert not ev.query(), "the wait did not hold the stream"
        flags[0] = 1
        lat.append(until(ev))
    lat = sorted(l for l in lat if l is not None)
    print(f"[{kind}] A latency flag -> copy behind the wait finished: median {1e6 * lat[len(lat) // 2]:.0f} us, min {1e6 * lat[0]:.0f}, max {1e6 * lat[-1]:.0f} ({len(lat)}/20)")
    flags.zero_()
    wait(kind, s1, 1, 1); wait(kind, s2, 2, 1)
    with torch.cuda.stream(s1):
        e1 = torch.cuda.Event(); e1.record(s1)
    with torch.cuda.stream(s2):
        e2 = torch.cuda.Event(); e2.record(s2)
    time.sleep(0.01)
    flags[2] = 1                       # stream 2 first
    t2 = until(e2, 1.0)
    print(f"[{kind}] B stream 2 released while stream 1 still waits: " + (f"proceeds after {1e6 * t2:.0f} us (waits of different streams are independent)" if t2 is not None else "BLOCKED behind stream 1's wait"))
    t0 = time.perf_counter()
    with torch.cuda.stream(s3):
        y.zero_()
    s3.synchronize()
    print(f"[{kind}] C memset + synchronize on an unrelated stream while stream 1 waits: {1e6 * (time.perf_counter() - t0):.0f} us")
    flags[1] = 1
    torch.cuda.synchronize()
