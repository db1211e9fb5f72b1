"""Can the attention launch be kept OFF a few CUs that are left to a co-running transfer kernel?  (make the
exchange cheap for the compute it overlaps; follow-up of profiles/r05/kv_contention.md.)

A launch's work-groups are dealt to the 8 XCDs statically and a resident copy work-group (an RCCL channel) costs ITS XCD 4-5 CUs'
worth of attention throughput.  hipExtStreamCreateWithCUMask gives a stream whose kernels may only use the CUs of a mask: if the
attention runs on a stream that leaves ONE CU per XCD free, do 8 resident copy work-groups settle on exactly those CUs, and what
does the attention then lose?  The probe (i) learns the mask-bit -> (XCD, CU) map from the work-group trace, (ii) times the
shard-shape attention on the masked stream alone and beside 8 / 16 resident copy work-groups holding 64 KiB LDS.
Run on the GPU box:  python tools/cu_mask_probe.py
"""
import ctypes, math, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from infinicube_amd import native
from infinicube_amd.videogen.ops import HipOps

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libkvoccupy.so")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-w", os.path.join(here, "kv_occupy.hip"), "-o", so], check=True)
occ = ctypes.CDLL(so)
occ.occ_start.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
occ.occ_masked_stream.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
occ.occ_stream_destroy.argtypes = [ctypes.c_void_p]
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
copied = torch.zeros((129,), dtype=torch.int64, device="cuda")
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(clear_bits):
    words = (NCU + 31) // 32
    m = [0xFFFFFFFF] * words
    for b in clear_bits:
        m[b // 32] &= ~(1 << (b % 32))
    arr = (ctypes.c_uint32 * words)(*m)
    h = ctypes.c_void_p()
    assert occ.occ_masked_stream(arr, words, ctypes.byref(h)) == 0, "hipExtStreamCreateWithCUMask failed"
    return h, torch.cuda.ExternalStream(h.value)


def slot(tr_row):
    hw, xcc = int(tr_row[2]), int(tr_row[3]) & 0xF
    return (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xF)        # (XCC, SE, SH, CU)


def run(n, stream, co=None, iters=3):
    q = torch.randn((n, d), device="cuda").to(torch.bfloat16)
    o = torch.empty_like(q)
    nwg = H * ((n + 255) // 256)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for _ in range(2):
            ops.attention(q, k, v, o, H, SCALE)
    torch.cuda.synchronize()
    buf = torch.zeros((nwg, 4), dtype=torch.int64, device="cuda")
    occ_slots = []
    if co:
        copied.zero_(); torch.cuda.synchronize()
        assert occ.occ_start(co, src.data_ptr(), dst.data_ptr(), 32 << 20 if co <= 2 else (64 << 20) // co // 4096 * 4096, 65536, copied.data_ptr(), 1 << 14, side.cuda_stream) == 0
        time.sleep(0.003)
    native.check(lib.icv_attention_trace(buf.data_ptr(), nwg), "trace on")
    ms = []
    with torch.cuda.stream(stream):
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            ops.attention(q, k, v, o, H, SCALE)
            e1.record(stream); e1.synchronize()
            ms.append(e0.elapsed_time(e1))
    native.check(lib.icv_attention_trace(None, 0), "trace off")
    if co:
        occ.occ_stop(); side.synchronize()
        c = copied.cpu()
        occ_slots = [(int(c[1 + 2 * i]) & 0xF, (int(c[2 + 2 * i]) >> 13) & 7, (int(c[2 + 2 * i]) >> 12) & 1, (int(c[2 + 2 * i]) >> 8) & 0xF) for i in range(min(co, 64))]
    tr = buf.cpu()
    used = {slot(r) for r in tr}
    return sorted(ms)[len(ms) // 2], used, occ_slots


main = torch.cuda.current_stream()
base_ms, all_cus, _ = run(9360, main)
print(f"unmasked stream, n = 9360: {base_ms:.3f} ms on {len(all_cus)} distinct CUs ({NCU} reported)")
# (i) which CUs does clearing bit b remove?  bits 0..9 and a few further ones
print("mask bit -> CU removed (XCC, SE, SH, CU):")
bitmap = {}
for b in list(range(0, 10)) + [32, 33, 64, 128, 255]:
    if b >= NCU:
        continue
    h, st = masked_stream([b])
    _, used, _ = run(9360, st, iters=1)
    gone = sorted(all_cus - used)
    bitmap[b] = gone
    print(f"  bit {b:3d}: {gone}")
    occ.occ_stream_destroy(h)
# (ii) one CU per XCD left free: try the two obvious hypotheses for which bits those are
for name, bits in (("bits 0..7 (bit b -> XCD b mod 8)", list(range(8))), ("bits 0, 32, 64, ... (32 bits per XCD)", [32 * i for i in range(8)])):
    h, st = masked_stream(bits)
    for n in (9360, 4680):
        alone, used, _ = run(n, st)
        base, _, _ = run(n, main)
        free = sorted(all_cus - used)
        line = f"{name}: n = {n}: unmasked {base:.3f} ms; masked alone {alone:.3f} ms ({100 * (alone / base - 1):+.1f} %), free CUs per XCD {[sum(1 for f in free if f[0] == x) for x in range(8)]}"
        for co in (8, 16):
            t_unm, _, where_u = run(n, main, co)
            t_msk, _, where_m = run(n, st, co)
            on_free = sum(1 for w in where_m if w in free)
            line += f"; + {co} resident copy work-groups: unmasked {t_unm:.3f} ms ({100 * (t_unm / base - 1):+.1f} %), masked {t_msk:.3f} ms ({100 * (t_msk / base - 1):+.1f} % vs unmasked alone; {on_free} of {co} co-runners on the free CUs)"
        print(line)
    occ.occ_stream_destroy(h)
