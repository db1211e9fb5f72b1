"""Can an RCCL kernel START while every CU holds an arrival-driven attention work-group that is WAITING for the rows that kernel delivers?

The arrival-driven attention (csrc/attn7p.hip) occupies all 256 CUs with 8-wave work-groups that hold 128 KiB of LDS each; a work-group
that reaches a piece whose rows have not landed spins until they do.  The copy-engine transport delivers rows with SDMA copies and
one-wave / one-thread flag kernels, which were SEEN to get scheduled beside the spinning work-groups (tests/test_attn_pieces_gpu.py).
The RCCL transports deliver rows with CHANNEL KERNELS: if those cannot become resident beside the attention work-groups (LDS / VGPR
budget of the CU), "<rccl transport>+arrival" would dead-lock until the in-kernel deadline on a real node - a hazard no 1-GPU
rehearsal over gloo (host-blocking waits) can show.  This probe makes it visible on ONE GPU: a 1-rank RCCL communicator
(icv_comm_create, world 1: ncclAllGather of one rank is a device copy done by RCCL's own kernel, with NCCL_MAX_NCHANNELS channels) delivers
the rows of a flagged piece AFTER the attention launch has filled the machine; the flag is raised behind it.
Outcome per setting: 'delivered under the launch' (the attention finished, no time-out) or 'STARVED' (the launch ran into its deadline).
    python tools/probe_rccl_under_arrival.py        (on the GPU box)
"""
import ctypes, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd import native
from infinicube_amd.videogen.ops import HipOps

ops = HipOps("cuda:0")
lib = ops.lib
H, n, S = 40, 4680, 37440
d = H * 128
SCALE = math.log(2.0)
torch.manual_seed(0)
q = torch.randn((n, d), device="cuda").to(torch.bfloat16)
kv = torch.cat([(torch.randn((S, d), device="cuda") * (128 ** -0.5 * math.log2(math.e))).to(torch.bfloat16),
                torch.randn((S, d), device="cuda").to(torch.bfloat16)], dim=1).contiguous()
own, late = kv[:n], kv[n:2 * n]                      # own rows (in place) | the piece RCCL delivers
rest = kv[2 * n:]
staged = late.clone()
want = torch.zeros((n, d), dtype=torch.bfloat16, device="cuda")
pieces_all = [(own[:, :d], own[:, d:], -1, 0), (late[:, :d], late[:, d:], -1, 0), (rest[:, :d], rest[:, d:], -1, 0)]
ops.attention_pieces(q, pieces_all, want, H, SCALE)
torch.cuda.synchronize()
idbuf = ctypes.create_string_buffer(native.COMM_ID_BYTES)
native.check(lib.icv_comm_unique_id(idbuf), "icv_comm_unique_id")
comm = ctypes.c_void_p()
native.check(lib.icv_comm_create(idbuf.raw, 0, 1, ctypes.byref(comm)), "icv_comm_create")
side = torch.cuda.Stream()
flags = torch.zeros((2,), dtype=torch.int32, device="cuda")
err = torch.zeros((1,), dtype=torch.int32, device="cuda")
mark = torch.zeros((2,), dtype=torch.int32, device="cuda")
row_bytes = 2 * d * 2
print(f"# NCCL_MAX_NCHANNELS={os.environ.get('NCCL_MAX_NCHANNELS', '(unset: RCCL default)')}; attention: {H * ((n + 255) // 256)} work-groups of 8 waves / 128 KiB LDS on 256 CUs; "
      f"the late piece = {n} rows x {row_bytes} B = {n * row_bytes / 1e6:.0f} MB delivered by a 1-rank ncclAllGather", flush=True)
for rep, deliver in enumerate(("rccl", "rccl", "blit")):
    late.fill_(float("nan")); flags.zero_(); err.zero_()
    o = torch.zeros_like(want)
    torch.cuda.synchronize()
    t0 = time.time()
    pieces = [(own[:, :d], own[:, d:], -1, 0), (late[:, :d], late[:, d:], 0, rep + 1), (rest[:, :d], rest[:, d:], -1, 0)]
    ops.attention_pieces(q, pieces, o, H, SCALE, flags=flags, err=err, timeout_us=3_000_000)
    with torch.cuda.stream(side):
        ops.flag_write(mark, 0, 1, delay_us=1000)          # 1 ms: by now every CU holds a work-group, the first round is past its own rows
        if deliver == "rccl":
            native.check(lib.icv_allgather_kv(comm, staged.data_ptr(), late.data_ptr(), n, row_bytes, side.cuda_stream), "icv_allgather_kv")
        else:
            late.copy_(staged)                               # control: the runtime's blit kernel (what a same-device copy-engine pull is)
        ops.flag_write(flags, 0, rep + 1)
    torch.cuda.synchronize()
    ms = (time.time() - t0) * 1e3
    e = int(err.item()) & 0xffffffff
    ok = e == 0 and bool(torch.equal(o, want))
    print(f"{deliver:5s} delivery, run {rep}: launch + delivery took {ms:8.1f} ms; " + ("delivered UNDER the launch, result bit-identical" if ok else
          f"STARVED: the attention ran into its 3 s deadline (err {e:#x}) - the delivering kernel could not start beside the waiting work-groups"), flush=True)
lib.icv_comm_destroy(comm)
