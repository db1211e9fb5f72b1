"""Round occupancy and XCD placement of the self-attention launch at the per-rank shard shapes.

Every attn7 work-group records {start, end (100 MHz ticks), HW_ID, XCC_ID} (icv_attention_trace).  From one launch per shape:
  * the ROUND STRUCTURE: 40 heads x ceil(n / 256) work-groups, one per CU at a time -> how many CUs are busy over time, how long the
    launch runs with fewer than all of them (the tail), and what a perfectly packed launch of the same work would take;
  * the PLACEMENT: attn7's (head, q-block) map assumes work-group b runs on XCD b mod 8 (each XCD's L2 then streams ONE head's
    K/V) - the fraction of work-groups for which that holds, alone and beside ONE resident copy work-group on a second stream
    (tools/kv_occupy.hip): the mechanism behind profiles/r05/kv_contention.md.
Run on the GPU box:  python tools/attn_round_trace.py
"""
import ctypes, math, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infinicube_amd import native
from infinicube_amd.videogen.ops import HipOps

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libkvoccupy.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-w", os.path.join(here, "kv_occupy.hip"), "-o", so], check=True)
occ = ctypes.CDLL(so)
occ.occ_start.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
assert occ.occ_init() == 0
ops = HipOps("cuda:0")
lib = ops.lib
H, S = 40, 37440
d = H * 128
SCALE = math.log(2.0)
torch.manual_seed(0)
k = (torch.randn((S, d), device="cuda") * (128 ** -0.5 * math.log2(math.e))).to(torch.bfloat16)
v = torch.randn((S, d), device="cuda").to(torch.bfloat16)
side = torch.cuda.Stream()
src = torch.empty((64 << 20,), dtype=torch.uint8, device="cuda").random_(0, 255)
dst = torch.empty_like(src)
copied = torch.zeros((129,), dtype=torch.int64, device="cuda")      # [0] = bytes moved, then (XCC_ID, HW_ID) per work-group


def traced_launch(n, co_runner=None):
    q = torch.randn((n, d), device="cuda").to(torch.bfloat16)
    o = torch.empty_like(q)
    nwg = H * ((n + 255) // 256)
    for _ in range(2):
        ops.attention(q, k, v, o, H, SCALE)
    torch.cuda.synchronize()
    buf = torch.zeros((nwg, 4), dtype=torch.int64, device="cuda")
    if co_runner:
        copied.zero_(); torch.cuda.synchronize()
        assert occ.occ_start(co_runner[0], src.data_ptr(), dst.data_ptr(), 32 << 20, co_runner[1], copied.data_ptr(), 1 << 14, side.cuda_stream) == 0
        time.sleep(0.003)
    native.check(lib.icv_attention_trace(buf.data_ptr(), nwg), "trace on")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.attention(q, k, v, o, H, SCALE)
    e1.record(); e1.synchronize()
    native.check(lib.icv_attention_trace(None, 0), "trace off")
    if co_runner:
        occ.occ_stop(); side.synchronize()
    return buf.cpu(), e0.elapsed_time(e1), nwg


def analyse(tr, ms, nwg, label):
    t0, t1 = tr[:, 0].double(), tr[:, 1].double()
    base = t0.min()
    st, en = (t0 - base) / 100.0, (t1 - base) / 100.0          # microseconds
    span = float(en.max())
    dur = en - st
    ev = sorted([(float(a), 1) for a in st] + [(float(b), -1) for b in en])
    active, last, busy_area, t_full, t_tail_start, peak = 0, 0.0, 0.0, 0.0, None, 0
    for t, dlt in ev:
        busy_area += active * (t - last)
        if active >= 256:
            t_full += t - last
        last = t
        active += dlt
        peak = max(peak, active)
    # the tail: from the moment the last work-group has STARTED (no more work to hand out) to the end
    tail = span - float(st.max())
    xcc = tr[:, 3] & 0xF
    # the dispatcher deals work-groups to the XCDs round-robin but continues where the previous launch stopped: the map the kernel
    # relies on is "b mod 8 -> ONE physical XCD" (any fixed rotation), so measure the share of the most common (xcc - b) mod 8
    rot = (xcc - torch.arange(nwg)) % 8
    placed = float(torch.bincount(rot, minlength=8).max()) / nwg
    per_xcd = [int((xcc == i).sum()) for i in range(8)]
    # per-XCD finish time and mean work-group duration: does ONE XCD lag (a CU short) or all of them (a global effect)?
    xcd_end = [float(en[xcc == i].max()) if per_xcd[i] else 0.0 for i in range(8)]
    xcd_dur = [float(dur[xcc == i].mean()) if per_xcd[i] else 0.0 for i in range(8)]
    cus = tr[:, 2] & 0xFFFF          # HW_ID low bits (wave / simd / cu / sh / se ids): distinct values ~ distinct (se, sh, cu, simd, wave) slots
    n_slots = int(torch.unique(torch.stack([xcc, (tr[:, 2] >> 8) & 0xF, (tr[:, 2] >> 12) & 0x1, (tr[:, 2] >> 13) & 0x7], 1), dim=0).shape[0])
    ideal = busy_area / 256.0
    return dict(label=label, nwg=nwg, rounds=nwg / 256.0, launch_us=ms * 1e3, span_us=span, mean_wg_us=float(dur.mean()), p95_wg_us=float(dur.quantile(0.95)),
                busy_cu_equiv=busy_area / span, ideal_us=ideal, tail_us=tail, placed=placed, per_xcd=per_xcd, peak=peak,
                xcd_end=xcd_end, xcd_dur=xcd_dur, n_cus=n_slots)


rows = []
for n, what in ((37440, "full S (1 GPU)"), (9360, "1/4 shard (cfg2 x sp4)"), (4680, "1/8 shard (sp8)")):
    for co, cname in ((None, "alone"), ((1, 0), "+ 1 light copy work-group"), ((1, 65536), "+ 1 copy work-group holding 64 KiB LDS")):
        tr, ms, nwg = traced_launch(n, co)
        rows.append(analyse(tr, ms, nwg, f"n = {n} {what}, {cname}"))
print("| launch | work-groups (rounds of 256 CUs) | launch | mean / p95 work-group | CUs busy (time average) | perfectly packed | tail after the last work-group started | b mod 8 -> one XCD (any rotation) | distinct CUs used | per-XCD finish (us) | per-XCD mean work-group (us) |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    fl = 4.0 * (r["nwg"] // 40 * 256) * 0  # (flops are in DESIGN.md; this table is about occupancy)
    print(f"| {r['label']} | {r['nwg']} ({r['rounds']:.2f}) | {r['launch_us'] / 1e3:.3f} ms | {r['mean_wg_us']:.0f} / {r['p95_wg_us']:.0f} us | {r['busy_cu_equiv']:.1f} of 256 "
          f"| {r['ideal_us'] / 1e3:.3f} ms ({100 * r['ideal_us'] / r['span_us']:.1f} % of the span) | {r['tail_us']:.0f} us ({100 * r['tail_us'] / r['span_us']:.1f} %) "
          f"| {100 * r['placed']:.1f} % | {r['n_cus']} | {' '.join(f'{x:.0f}' for x in r['xcd_end'])} | {' '.join(f'{x:.0f}' for x in r['xcd_dur'])} |")
