"""Self-attention at the per-rank SHARD shapes of the N-GPU run (14B: n = 9 360 query rows = 1/4 shard of cfg2 x sp4, n = 4 680 =
1/8 shard of sp8, against all 37 440 keys arriving in ramped chunks): where does the time go relative to one unchunked
launch, and what do the launch-shape options buy?  Run on the GPU box.  """
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinicube_amd import native
from infinicube_amd.videogen.ops import HipOps
from infinicube_amd.videogen.seqpar import chunk_bounds

ops = HipOps("cuda:0")
lib = ops.lib
H, S = 40, 37440
d = H * 128
SCALE = math.log(2.0)
ITERS = int(os.environ.get("ITERS", "5"))
torch.manual_seed(0)
k = (torch.randn((S, d), device="cuda") * (128 ** -0.5 * math.log2(math.e))).to(torch.bfloat16)
v = torch.randn((S, d), device="cuda").to(torch.bfloat16)


def timeit(fn):
    fn(); torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(ITERS):
            fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / ITERS)
    return sorted(best)[1]


def chunk_call(q, kk, vv, o, acc, ml, heads, first, last, stream):
    native.check(lib.icv_attention_fwd_chunk(q.data_ptr(), q.stride(0), kk.data_ptr(), kk.stride(0), vv.data_ptr(), vv.stride(0), o.data_ptr(), o.stride(0),
                                             acc.data_ptr(), acc.stride(0), ml.data_ptr(), q.shape[0], kk.shape[0], heads, SCALE, int(first), int(last),
                                             stream), "chunk")


for world in (4, 8):
    n = S // world
    q = torch.randn((n, d), device="cuda").to(torch.bfloat16)
    o = torch.empty_like(q)
    acc = torch.empty((n, d), device="cuda")
    ml = torch.empty((n, H, 2), device="cuda")
    fl = 4.0 * n * S * d
    rows = []
    main = torch.cuda.current_stream()
    t_one = timeit(lambda: ops.attention(q, k, v, o, H, SCALE))
    rows.append(("one launch (no chunks)", t_one))
    for C in (4, 2):
        b = [world * x for x in chunk_bounds(n, C)]

        def seq():
            for c in range(C):
                chunk_call(q, k[b[c]:b[c + 1]], v[b[c]:b[c + 1]], o, acc, ml, H, c == 0, c == C - 1, main.cuda_stream)
        rows.append((f"{C} ramped chunks", timeit(seq)))
        if C == 4:     # the layout the DiT really reads: K and V as the two column halves of ONE gathered [S, 2d] row matrix
            kv = torch.cat([k, v], dim=1).contiguous()
            kh, vh = kv[:, :d], kv[:, d:]

            def seq_kv():
                for c in range(C):
                    chunk_call(q, kh[b[c]:b[c + 1]], vh[b[c]:b[c + 1]], o, acc, ml, H, c == 0, c == C - 1, main.cuda_stream)
            rows.append((f"{C} ramped chunks, K|V halves of one [S, 2d] matrix", timeit(seq_kv)))
            del kv, kh, vh
            # round 6: ONE arrival-gated launch over the pieces of the same exchange (csrc/attn7p.hip): own rows as one piece, then
            # (chunk, peer) pieces cut on the 64-key tile grid - every flag already satisfied: the kernel-only view of the schedule
            ba = chunk_bounds(n, C, 64)
            kv2 = torch.cat([k, v], dim=1).contiguous()
            own = kv2[:n]
            peers_kv = kv2[n:].view(world - 1, n, 2 * d)
            pieces = [(own[:, :d], own[:, d:], -1, 0)] + [(peers_kv[j, ba[c]:ba[c + 1], :d], peers_kv[j, ba[c]:ba[c + 1], d:], -1, 0)
                                                          for c in range(C) for j in range(world - 1)]
            rows.append((f"ONE arrival-gated launch, {len(pieces)} pieces (own + {C} chunks x {world - 1} peers)", timeit(lambda: ops.attention_pieces(q, pieces, o, H, SCALE))))
            rows.append(("ONE arrival-gated launch, 1 piece (all keys)", timeit(lambda: ops.attention_pieces(q, [(kv2[:, :d], kv2[:, d:], -1, 0)], o, H, SCALE))))
            del kv2, own, peers_kv, pieces
        lib.icv_set_option(b"attn7_short", 1 << 30)
        rows.append((f"{C} ramped chunks, 4-wave blocks two per CU", timeit(seq)))
        lib.icv_set_option(b"attn7_short", -1)
        for G in (2, 4):
            hg = H // G
            streams = [torch.cuda.Stream() for _ in range(G)]
            mls = [torch.empty((n, hg, 2), device="cuda") for _ in range(G)]

            def split():
                ev = torch.cuda.Event(); ev.record()
                for g, st in enumerate(streams):
                    st.wait_event(ev)
                    cs = slice(g * hg * 128, (g + 1) * hg * 128)
                    for c in range(C):
                        chunk_call(q[:, cs], k[b[c]:b[c + 1], cs], v[b[c]:b[c + 1], cs], o[:, cs], acc[:, cs], mls[g], hg, c == 0, c == C - 1, st.cuda_stream)
                    e = torch.cuda.Event(); e.record(st); main.wait_event(e)
            rows.append((f"{C} ramped chunks, heads split over {G} streams", timeit(split)))
    print(f"--- n = {n} query rows (1/{world} shard), 37 440 keys, 40 heads")
    for name, ms in rows:
        print(f"{name:52s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {100 * t_one / ms:5.1f} % of the unchunked launch")
t_full = None
q = torch.randn((S, d), device="cuda").to(torch.bfloat16); o = torch.empty_like(q)
t_full = timeit(lambda: ops.attention(q, k, v, o, H, SCALE))
print(f"full S x S launch: {t_full:.3f} ms {4.0 * S * S * d / t_full / 1e9:.1f} TF/s; 1/4 of it {t_full / 4:.3f} ms, 1/8 {t_full / 8:.3f} ms")
