"""What does a co-running K|V transfer cost the attention it is supposed to hide under?  (one GPU.)

Self-attention at the per-rank shard shapes of the 8-GPU run (14B: n = 9 360 = cfg2 x sp4, n = 4 680 = sp8; 37 440 keys in 4 ramped
chunks, K|V as the column halves of one [S, 2d] matrix: exactly the launches of the loop) is timed alone and then beside a
transfer that runs for its whole duration on a second stream:
  * `k` resident copy work-groups (tools/kv_occupy.hip), k = 1..32, in two footprints: "light" (no LDS: can share a CU with
    other waves) and "64K LDS" (cannot share a CU with an attention work-group: takes the CU for itself) — what an RCCL
    all-gather with k channels is to the attention launch;
  * a loop of hipMemcpyAsync device-to-device copies (same device: the runtime's blit KERNEL — not what the copy-engine
    transport does between two devices, where the copy is SDMA; here it shows what a chip-wide copy kernel costs);
  * a 1-rank RCCL all-gather loop on libicvideo's own communicator (`icv_allgather_kv`), with NCCL_MAX_NCHANNELS from the
    environment (run the tool once per setting);
  * `k` pending hipStreamWaitValue32 on k side streams (WHAT=spin): what the copy-engine transport (csrc/ipc.hip) keeps resident
    while a peer's rows are not published yet - on this runtime each wait is a one-wave kernel that spins
    (profiles/r05/stream_ops_probe.txt); k = 3 / 7 = the pull streams of an sp4 / sp8 group, waiting for the WHOLE measurement
    (the worst case: in the loop a wait lasts for the skew between two ranks).
Prints one table per shard shape: attention ms, slow-down vs alone, and the bytes/s the transfer moved meanwhile.
    python tools/kv_contention.py            (on the GPU box; ITERS=5 repetitions per cell)
"""
import ctypes, math, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infinicube_amd import native
from infinicube_amd.videogen.ops import HipOps
from infinicube_amd.videogen.seqpar import chunk_bounds

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libkvoccupy.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "kv_occupy.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-w", os.path.join(here, "kv_occupy.hip"), "-o", so], check=True)
occ = ctypes.CDLL(so)
occ.occ_start.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
occ.occ_host_wait.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
assert occ.occ_init() == 0

ops = HipOps("cuda:0")
lib = ops.lib
H, S = 40, 37440
d = H * 128
SCALE = math.log(2.0)
ITERS = int(os.environ.get("ITERS", "5"))
WARM = os.environ.get("WARM", "1") == "1"
WHAT = os.environ.get("WHAT", "occupy,blit,rccl,gemm").split(",")
torch.manual_seed(0)
kv = torch.cat([(torch.randn((S, d), device="cuda") * (128 ** -0.5 * math.log2(math.e))).to(torch.bfloat16),
                torch.randn((S, d), device="cuda").to(torch.bfloat16)], dim=1).contiguous()
kh, vh = kv[:, :d], kv[:, d:]
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
SLICE = 32 << 20                     # bytes each copy work-group loops over
src = torch.empty((32 * SLICE,), dtype=torch.uint8, device="cuda").random_(0, 255)
dst = torch.empty_like(src)
copied = torch.zeros((129,), dtype=torch.int64, device="cuda")      # [0] = bytes moved, then (XCC_ID, HW_ID) per work-group


def chunk_call(q, kk, vv, o, acc, ml, first, last):
    native.check(lib.icv_attention_fwd_chunk(q.data_ptr(), q.stride(0), kk.data_ptr(), kk.stride(0), vv.data_ptr(), vv.stride(0), o.data_ptr(), o.stride(0),
                                             acc.data_ptr(), acc.stride(0), ml.data_ptr(), q.shape[0], kk.shape[0], H, SCALE, int(first), int(last),
                                             main.cuda_stream), "chunk")


def time_attention(run, before=None, after=None):
    """median-of-3 of (ITERS x run) on the main stream; `before` starts the co-runner, `after` stops it and returns bytes moved."""
    run(); torch.cuda.synchronize()
    res = []
    for _ in range(3):
        if before:
            before()
            time.sleep(0.003)          # the co-runner is resident before the first attention work-group is dispatched
            if WARM:
                run()                  # untimed: the clocks are back up after the idle gap above (WARM=1; see the "control" rows)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(ITERS):
            run()
        e1.record()
        e1.synchronize()
        wall = time.perf_counter() - t0
        moved = after() if after else 0
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) / ITERS, moved / wall if wall > 0 else 0.0))
    return sorted(res)[1]


def occupier(k, lds):
    def before():
        copied.zero_()
        torch.cuda.synchronize()
        rc = occ.occ_start(k, src.data_ptr(), dst.data_ptr(), SLICE, lds, copied.data_ptr(), 1 << 14, side.cuda_stream)
        assert rc == 0, rc

    def after():
        occ.occ_stop()
        side.synchronize()
        return int(copied[0].item())
    return before, after


def blit_loop(nbytes):
    state = {}

    def before():
        state["n"] = 0
        with torch.cuda.stream(side):
            for _ in range(64):        # queued ahead: the stream stays busy for the whole measurement
                dst[:nbytes].copy_(src[:nbytes], non_blocking=True)
                state["n"] += 1

    def after():
        side.synchronize()
        return 0
    return before, after


hip = ctypes.CDLL("libamdhip64.so")
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
spin_flag = torch.zeros((16,), dtype=torch.int32).pin_memory()
spin_dp = ctypes.c_void_p()
assert hip.hipHostGetDevicePointer(ctypes.byref(spin_dp), ctypes.c_void_p(spin_flag.data_ptr()), 0) == 0
spin_streams = [torch.cuda.Stream() for _ in range(7)]
spin_seq = [0]


def spin_waits(k, hostfunc=False):
    """NOTE: a pending wait blocks every stream that shares its HARDWARE queue (streams are dealt to GPU_MAX_HW_QUEUES queues):
    if the measuring stream lands behind one, only the watchdog below ends the measurement (it is reported)."""
    import threading
    state = {}

    def release(late):
        if late:
            print(f"    !! watchdog released the {k} waits after 20 s: the main stream was queued BEHIND a waiting stream "
                  f"(GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(default)')})", flush=True)
        spin_flag[0] = spin_seq[0]          # host write: releases every waiter

    def before():
        spin_seq[0] += 1
        for st in spin_streams[:k]:
            if hostfunc:
                assert occ.occ_host_wait(ctypes.c_void_p(st.cuda_stream), ctypes.c_void_p(spin_flag.data_ptr()), spin_seq[0], None) == 0
            else:
                assert hip.hipStreamWaitValue32(ctypes.c_void_p(st.cuda_stream), spin_dp, spin_seq[0], 0, 0xFFFFFFFF) == 0
        state["t"] = threading.Timer(20.0, release, args=(True,))
        state["t"].start()

    def after():
        state["t"].cancel()
        release(False)
        for st in spin_streams[:k]:
            st.synchronize()
        return 0
    return before, after


rccl = None
if "rccl" in WHAT:
    try:
        idbuf = ctypes.create_string_buffer(native.COMM_ID_BYTES)
        native.check(lib.icv_comm_unique_id(idbuf), "uid")
        hcomm = ctypes.c_void_p()
        native.check(lib.icv_comm_create(idbuf.raw, 0, 1, ctypes.byref(hcomm)), "comm")
        rccl = hcomm
    except Exception as e:  # noqa: BLE001
        print(f"(1-rank RCCL communicator unavailable: {e})")


def rccl_loop(nbytes):
    def before():
        for _ in range(64):
            native.check(lib.icv_allgather_kv(rccl, src.data_ptr(), dst.data_ptr(), nbytes // 4096, 4096, side.cuda_stream), "ag")

    def after():
        side.synchronize()
        return 0
    return before, after


print(f"# attention under a co-running transfer; NCCL_MAX_NCHANNELS={os.environ.get('NCCL_MAX_NCHANNELS', '(unset)')} ITERS={ITERS}")
for world in (4, 8):
    n = S // world
    q = torch.randn((n, d), device="cuda").to(torch.bfloat16)
    o = torch.empty_like(q)
    acc = torch.empty((n, d), device="cuda")
    ml = torch.empty((n, H, 2), device="cuda")
    b = [world * x for x in chunk_bounds(n, 4)]
    fl = 4.0 * n * S * d

    def run():
        for c in range(4):
            chunk_call(q, kh[b[c]:b[c + 1]], vh[b[c]:b[c + 1]], o, acc, ml, c == 0, c == 3)

    b2b, _ = time_attention(run)
    # the BASELINE goes through the same harness as every co-runner row (start hook, 3 ms for the co-runner to become resident, an
    # untimed launch, the timed launches, stop hook): measured late in round 5, the idle gap alone costs the launches that follow it
    # 2-8 % (clock ramp), which the first version of this tool booked on the co-runner
    alone, _ = time_attention(run, lambda: None, lambda: 0)
    print(f"--- n = {n} query rows (1/{world} shard), 37 440 keys in 4 ramped chunks, 40 heads: alone {alone:.3f} ms = {fl / alone / 1e9:.0f} TF/s "
          f"(same harness, no co-runner; back-to-back launches without the harness: {b2b:.3f} ms)")
    print(f"{'co-runner':44s} {'attn ms':>8s} {'slow-down':>10s} {'moved GB/s':>11s}")
    rows = []
    if "occupy" in WHAT:
        for lds, name in ((0, "light"), (65536, "64K LDS")):
            for k in (1, 2, 4, 8, 16, 32):
                ms, rate = time_attention(run, *occupier(k, lds))
                rows.append((f"{k:2d} copy work-groups ({name})", ms, rate))
    if "control" in WHAT:      # no co-runner at all: what the harness itself (idle gap before the timed region) does to the number
        ms, _ = time_attention(run, lambda: None, lambda: 0)
        rows.append(("control: NO co-runner, same harness", ms, 0.0))
    if "spin" in WHAT:
        for k in (1, 3, 7):
            ms, _ = time_attention(run, *spin_waits(k))
            rows.append((f"{k} pending hipStreamWaitValue32 (spin waves)", ms, 0.0))
    if "hostfunc" in WHAT:
        for k in (1, 3, 7):
            ms, _ = time_attention(run, *spin_waits(k, hostfunc=True))
            rows.append((f"{k} pending host-function waits (no wave)", ms, 0.0))
    chunk_bytes = 2 * 2 * d * (n - n // 10)       # ~ one large chunk of one peer's K|V rows
    if "blit" in WHAT:
        ms, _ = time_attention(run, *blit_loop(min(chunk_bytes, src.numel())))
        rows.append((f"hipMemcpyAsync D2D loop ({min(chunk_bytes, src.numel()) >> 20} MiB, same-device blit kernel)", ms, 0.0))
    if rccl is not None:
        ms, _ = time_attention(run, *rccl_loop(min(chunk_bytes, src.numel()) // 4096 * 4096))
        rows.append(("1-rank RCCL all-gather loop", ms, 0.0))
    for name, ms, rate in rows:
        print(f"{name:44s} {ms:8.3f} {100 * (ms / alone - 1):9.1f}% {rate / 1e9:11.1f}")
    if "gemm" in WHAT:
        # the other kernel a transfer overlaps with: the Q projection [n, d] x [d, d] that runs while the first K|V chunks travel
        from infinicube_amd.videogen.ops import EPI_BF16
        a = torch.randn((n, d), device="cuda").to(torch.bfloat16)
        w = (torch.randn((d, d), device="cuda") / math.sqrt(d)).to(torch.bfloat16)
        bias = torch.randn((d,), device="cuda")
        og = torch.empty((n, d), dtype=torch.bfloat16, device="cuda")

        def run_g():
            for _ in range(4):
                ops.gemm(a, w, bias, og, EPI_BF16)

        g_alone, _ = time_attention(run_g, lambda: None, lambda: 0)
        print(f"    Q-projection GEMM [{n}, {d}] x [{d}, {d}] (x4 per sample): alone {g_alone / 4:.3f} ms = {2.0 * n * d * d / (g_alone / 4) / 1e9:.0f} TF/s")
        for lds, nm in ((0, "light"), (65536, "64K LDS")):
            for k in (1, 8, 32):
                ms, _ = time_attention(run_g, *occupier(k, lds))
                print(f"    {k:2d} copy work-groups ({nm:7s}): {ms / 4:.3f} ms ({100 * (ms / g_alone - 1):+.1f} %)")
