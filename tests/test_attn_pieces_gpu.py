"""K6, arrival-driven (csrc/attn7p.hip, icv_attention_fwd_pieces): ONE attention launch over a list of K|V pieces gated by arrival
flags - the sequence-parallel schedule of SURVEY.md §8e ("process K/V chunks in arrival order (own shard first) with online-softmax
merging").  Parity vs the fp32 oracle at the attention tolerance of tests/test_kernels_gpu.py, bit-identity with the plain launch
where the tile sequence is the same, and the protocol itself: rows that are written AFTER the launch started (a late peer), the
work-group progress before they land, and the bounded wait."""
import math

import pytest
import torch

from oracle import wan_ref as R
from test_kernels_gpu import assert_bf16_close, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SCALE = 1.0 / math.sqrt(128)


def _qkv(Sq, Skv, H, seed):
    d = H * 128
    q = rnd((Sq, d), seed).to(torch.bfloat16)
    kv = rnd((Skv, 2 * d), seed + 1).to(torch.bfloat16)      # K|V rows as ONE matrix, as the sequence-parallel workspace holds them
    return q, kv


def _views(kv, bounds, d):
    return [(kv[a:b, :d], kv[a:b, d:]) for a, b in zip(bounds[:-1], bounds[1:])]


@pytest.mark.parametrize("unit", [False, True])
def test_tile_aligned_pieces_in_memory_order_are_bit_identical_to_the_plain_launch(hip_ops, unit):
    """Same tiles, same order, same arithmetic: cutting the key axis at multiples of 64 rows changes which DMA requests share an
    interval and nothing else (odd tile counts per piece shift the pairing).  `unit`: scale = ln 2, i.e. the unit-scale code path
    the DiT runs (softmax scale folded into K)."""
    Sq, Skv, H = 700, 64 * 37 + 11, 3
    d = H * 128
    q, kv = _qkv(Sq, Skv, H, 300)
    if unit:
        kv[:, :d] = (kv[:, :d].float() * SCALE * 1.4426950408889634).to(torch.bfloat16)
    scale = math.log(2.0) if unit else SCALE
    q, kv = q.to(DEV), kv.to(DEV)
    o_ref = torch.zeros((Sq, d), dtype=torch.bfloat16, device=DEV)
    hip_ops.lib.icv_set_option(b"attn7_plain", 1)        # the reference = attn7.hip's own kernel (the default launch IS one piece of attn7p since round 6)
    try:
        hip_ops.attention(q, kv[:, :d], kv[:, d:], o_ref, H, scale)
    finally:
        hip_ops.lib.icv_set_option(b"attn7_plain", 0)
    o_def = torch.zeros_like(o_ref)
    hip_ops.attention(q, kv[:, :d], kv[:, d:], o_def, H, scale)
    assert torch.equal(o_def, o_ref), "the default launch (one piece of attn7p) differs from attn7.hip's kernel"
    for bounds in ([0, Skv], [0, 64, Skv], [0, 64 * 5, 64 * 6, 64 * 20, Skv], [0, 64 * 36, Skv]):
        o = torch.zeros_like(o_ref)
        hip_ops.attention_pieces(q, [(k, v, -1, 0) for k, v in _views(kv, bounds, d)], o, H, scale)
        assert torch.equal(o, o_ref), f"pieces {bounds}: differs from the plain launch"


def test_ragged_pieces_any_order_match_the_oracle_and_the_chunked_launches(hip_ops):
    """Ragged pieces (rows not a multiple of 64: a masked tail tile inside the key axis), one-row and empty pieces, out of memory order -
    against the fp32 oracle, and against the carried-state chunk launches the single launch replaces."""
    Sq, H = 520, 2
    d = H * 128
    bounds = [0, 1, 1, 130, 777, 777 + 64, 2001]
    q, kv = _qkv(Sq, bounds[-1], H, 310)
    kv[1990, :d] = q[7] * 5.0                       # a spike in the last (ragged) piece and one early: both rescale paths
    kv[70, :d] = q[300] * 5.0
    ref = R.attention(q.float(), kv[:, :d].float(), kv[:, d:].float(), H)
    q, kv = q.to(DEV), kv.to(DEV)
    views = _views(kv, bounds, d)
    order = [3, 0, 5, 1, 2, 4]
    o = torch.zeros((Sq, d), dtype=torch.bfloat16, device=DEV)
    hip_ops.attention_pieces(q, [(views[i][0], views[i][1], -1, 0) for i in order], o, H, SCALE)
    o2 = torch.zeros_like(o)
    hip_ops.attention_pieces(q, [(views[i][0], views[i][1], -1, 0) for i in order], o2, H, SCALE)
    assert torch.equal(o, o2), "non-deterministic output"
    assert_bf16_close(o, ref, "attention_pieces, ragged", abs_floor=2.0 ** -5, rms_bound=2.0 ** -7)
    acc = torch.zeros((Sq, d), dtype=torch.float32, device=DEV)
    ml = torch.zeros((Sq, H, 2), dtype=torch.float32, device=DEV)
    oc = torch.zeros_like(o)
    live = [i for i in order if views[i][0].shape[0]]
    for j, i in enumerate(live):
        hip_ops.attention_chunk(q, views[i][0], views[i][1], oc, acc, ml, H, SCALE, first=j == 0, last=j == len(live) - 1)
    assert_bf16_close(o, oc.float(), "attention_pieces vs chunk launches", abs_floor=2.0 ** -6, rms_bound=2.0 ** -8)


@pytest.mark.parametrize("Sq,H,late_ms", [(1024, 4, 3), (4680, 40, 2)])
def test_rows_that_land_after_the_launch_started(hip_ops, Sq, H, late_ms):
    """The protocol: the launch starts while two pieces hold GARBAGE (NaN); a side stream delivers them late - copy, then the flag,
    exactly what the K|V exchange does - one after ~late_ms, one straight away.  The result must equal the all-present launch bit
    for bit (same piece order), the per-(work-group, piece) trace must show that work-groups consumed the other pieces BEFORE the
    late rows existed (progress under the transfer: SURVEY §8e), and nobody may have timed out.  (4680, 40): 760 work-groups = 2.97
    rounds of 256 CUs, every CU holding a work-group that spins - the flag writer still has to get a wave slot."""
    d = H * 128
    rows = [1500, 1100, 700, 900]                   # own | early peer | LATE peer | present peer
    bounds = [0]
    for r in rows:
        bounds.append(bounds[-1] + r)
    q, kv = _qkv(Sq, bounds[-1], H, 320)
    q, kv = q.to(DEV), kv.to(DEV)
    views = _views(kv, bounds, d)
    want = torch.zeros((Sq, d), dtype=torch.bfloat16, device=DEV)
    hip_ops.attention_pieces(q, [(k, v, -1, 0) for k, v in views], want, H, SCALE)
    staged = kv.clone()
    kv[bounds[1]:bounds[3]] = float("nan")                                    # pieces 1 and 2 are not there yet
    flags = torch.zeros((4,), dtype=torch.int32, device=DEV)
    err = torch.zeros((1,), dtype=torch.int32, device=DEV)
    nwg = H * ((Sq + 255) // 256)
    trace = torch.zeros((nwg, 4), dtype=torch.int64, device=DEV)
    mark = torch.zeros((2,), dtype=torch.int64, device=DEV)
    side = torch.cuda.Stream(device=DEV, priority=-1)       # its own hardware-queue class: a default-class stream can share the launch's queue and never run (profiles/r06/stream_queue_share_probe.txt)
    o = torch.zeros_like(want)
    for value in (7, 8):          # twice: the first pass pays the side stream's first-use costs (its queue, the copy kernels' module load)
        kv[bounds[1]:bounds[3]] = float("nan")
        trace.zero_()
        o.zero_()
        torch.cuda.synchronize()
        pieces = [(views[0][0], views[0][1], -1, 0), (views[1][0], views[1][1], 1, value), (views[2][0], views[2][1], 2, value),
                  (views[3][0], views[3][1], -1, 0)]
        hip_ops.attention_pieces(q, pieces, o, H, SCALE, flags=flags, err=err, timeout_us=5_000_000, trace=trace)
        with torch.cuda.stream(side):
            kv[bounds[1]:bounds[2]].copy_(staged[bounds[1]:bounds[2]])            # the early peer: rows, then its flag
            hip_ops.flag_write(flags, 1, value)
            hip_ops.flag_write(mark.view(torch.int32), 0, 1, delay_us=late_ms * 1000)   # hold the stream: the late peer
            kv[bounds[2]:bounds[3]].copy_(staged[bounds[2]:bounds[3]])
            hip_ops.flag_write(flags, 2, value)
        torch.cuda.synchronize()
    assert int(err.item()) == 0, f"a work-group gave up waiting: err word {int(err.item()) & 0xffffffff:#x}"
    assert torch.isfinite(o.float()).all(), "rows were read before they landed"
    assert torch.equal(o, want), "late rows: result differs from the all-present launch"
    t = trace.cpu()                                                           # 100 MHz ticks
    t0 = int(t[:, 0].min())
    first_round = t[:, 0] < t0 + 50_00                                         # work-groups that started within 50 us of the launch
    waited = (t[first_round, 2] - t[first_round, 0]).float() / 100.0          # us from their start to their start of the LATE piece
    assert waited.min() >= 0.6 * late_ms * 1000, f"a first-round work-group started the late piece after {waited.min():.0f} us: before its rows existed?"
    early = (t[first_round, 1] - t[first_round, 0]).float() / 100.0
    assert early.max() < 0.5 * late_ms * 1000, "the pieces that WERE there must be consumed while the late one is still in flight"
    print(f"Sq={Sq} H={H}: {int(first_round.sum())} first-round work-groups; own piece done + early piece started after {early.median():.0f} us (max {early.max():.0f}), "
          f"late piece started after {waited.median():.0f} us (late by {late_ms} ms)")


def test_a_piece_that_never_arrives_is_a_bounded_wait_and_an_error_word(hip_ops):
    """A dead peer: the flag is never written.  The launch must END (time-out 20 ms) and say which piece it gave up on."""
    Sq, H = 512, 2
    d = H * 128
    q, kv = _qkv(Sq, 1024, H, 330)
    q, kv = q.to(DEV), kv.to(DEV)
    views = _views(kv, [0, 512, 1024], d)
    flags = torch.zeros((2,), dtype=torch.int32, device=DEV)
    err = torch.zeros((1,), dtype=torch.int32, device=DEV)
    o = torch.zeros((Sq, d), dtype=torch.bfloat16, device=DEV)
    import time
    t0 = time.time()
    hip_ops.attention_pieces(q, [(views[0][0], views[0][1], -1, 0), (views[1][0], views[1][1], 0, 1)], o, H, SCALE, flags=flags, err=err,
                             timeout_us=20_000)
    torch.cuda.synchronize()
    assert time.time() - t0 < 5.0
    assert (int(err.item()) & 0xffffffff) == (0x80000000 | 1)


def test_autotune_drops_a_candidate_whose_rows_never_come_under_the_waiting_attention(hip_ops, monkeypatch):
    """The hazard no 1-GPU rehearsal of a transport can show: a kernel-based transport (an RCCL channel kernel) that finds no room on CUs
    full of WAITING attention work-groups would starve the arrival-driven launch until its in-kernel deadline (a 1-rank ncclAllGather is
    a plain device copy - tools/probe_rccl_under_arrival.py - so RCCL itself cannot be provoked here).  seqpar.autotune_kv_exchange
    therefore warms every candidate up under a SHORT deadline and drops the one whose launch gave up.  Here a stand-in engine has one
    candidate whose flag is never raised and one that is fine: the first must cost about the short deadline, be reported as starved,
    and the second must win."""
    import time
    from infinicube_amd.videogen import seqpar
    monkeypatch.setattr(seqpar, "AUTOTUNE_WAIT_DEADLINE_US", 50_000)
    H, Sq = 2, 512
    d = H * 128
    q, kv = _qkv(Sq, 1024, H, 340)
    q, kv = q.to(DEV), kv.to(DEV)
    views = _views(kv, [0, 512, 1024], d)

    class Engine:
        sp_timeout_us = 60_000_000

        def set_kv_exchange(self, mode, chunks):
            self.mode = mode
            self.err = torch.zeros((1,), dtype=torch.int32, device=DEV)
            self.flags = torch.zeros((1,), dtype=torch.int32, device=DEV)
            self.o = torch.zeros((Sq, d), dtype=torch.bfloat16, device=DEV)
            return self

        def run(self):
            gate = 0 if self.mode == "starved+arrival" else -1
            hip_ops.attention_pieces(q, [(views[0][0], views[0][1], -1, 0), (views[1][0], views[1][1], gate, 1)], self.o, H, SCALE,
                                     flags=self.flags, err=self.err, timeout_us=self.sp_timeout_us)

        def exchange_gave_up(self):
            e = int(self.err.item()) & 0xffffffff
            return f"attention gave up waiting for K|V piece {e & 0xffff}" if e else None

    eng = Engine()
    t0 = time.time()
    best, table = seqpar.autotune_kv_exchange(eng, eng.run, torch.cuda.synchronize, [("starved+arrival", 4), ("fine+arrival", 4)], reps=1)
    assert time.time() - t0 < 5.0, "the starved candidate must cost the SHORT deadline, not the product's"
    assert best == ("fine+arrival", 4) and eng.mode == "fine+arrival" and eng.sp_timeout_us == 60_000_000
    assert table[0]["ms"] is None and "starved or stalled" in table[0]["error"] and "piece 1" in table[0]["error"] and table[1]["ms"] > 0
