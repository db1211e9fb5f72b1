"""Does a kernel that WAITS for a flag on the current stream block the stream that is to write the flag?  (Streams share hardware queues:
GPU_MAX_HW_QUEUES, default 4.)  For side streams 0..N-1 of torch's pool and for a high-priority stream: an arrival-gated attention launch
(time-out 0.2 s) on the current stream, the flag written from the side stream right after - did the launch time out?"""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd.videogen.ops import HipOps
ops = HipOps("cuda:0")
DEV = "cuda:0"
H, d, Sq, m = 1, 128, 64, 128
q = torch.randn((Sq, d), device=DEV).to(torch.bfloat16)
kv = torch.randn((2 * m, 2 * d), device=DEV).to(torch.bfloat16)
o = torch.zeros_like(q)
flags = torch.zeros((4,), dtype=torch.int32, device=DEV)
err = torch.zeros((1,), dtype=torch.int32, device=DEV)
value = 0


def trial(side):
    global value
    value += 1
    err.zero_()
    torch.cuda.synchronize()
    t0 = time.time()
    ops.attention_pieces(q, [(kv[:m, :d], kv[:m, d:], -1, 0), (kv[m:, :d], kv[m:, d:], 1, value)], o, H, 1.0 / math.sqrt(128), flags=flags, err=err, timeout_us=200_000)
    with torch.cuda.stream(side):
        ops.flag_write(flags, 1, value)
    torch.cuda.synchronize()
    return int(err.item()) != 0, time.time() - t0


print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "unset (4)"))
pool = [torch.cuda.Stream(device=DEV) for _ in range(12)]
for i, s in enumerate(pool):
    trial(s)                                    # first use of the stream
    res = [trial(s) for _ in range(3)]
    print(f"pool stream {i:2d}: timed out {[r[0] for r in res]}  {['%.3f s' % r[1] for r in res]}")
hp = torch.cuda.Stream(device=DEV, priority=-1)
trial(hp)
res = [trial(hp) for _ in range(3)]
print(f"high-priority stream: timed out {[r[0] for r in res]}  {['%.3f s' % r[1] for r in res]}")
