"""An independent reader for the H.264 subset infinicube_amd/videogen/h264pcm.py writes — test infrastructure.

No third-party H.264 decoder exists in this image, so the stream is checked by a parser written from the syntax
tables of ITU-T H.264 (03/2010 numbering): §7.3.1 NAL unit (emulation prevention removed), §7.3.2.1.1 sequence
parameter set incl. Annex E VUI, §7.3.2.2 picture parameter set, §7.3.3 slice header, §7.3.4 slice data (CAVLC,
I slices) and §7.3.5 macroblock layer restricted to mb_type I_PCM.  It shares no code with the writer: it reads
bits, it never writes them.  Anything outside the subset raises, so a stream that parses here is a stream a full
decoder can follow field for field.
"""
import struct

import numpy as np


class BitReader:
    def __init__(self, data: bytes):
        self.d, self.pos = data, 0

    def u(self, n: int) -> int:
        v = 0
        for _ in range(n):
            byte = self.d[self.pos >> 3]
            v = (v << 1) | ((byte >> (7 - (self.pos & 7))) & 1)
            self.pos += 1
        return v

    def ue(self) -> int:
        zeros = 0
        while self.u(1) == 0:
            zeros += 1
            if zeros > 32:
                raise ValueError("bad Exp-Golomb code")
        return (1 << zeros) - 1 + (self.u(zeros) if zeros else 0)

    def se(self) -> int:
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)

    def byte_aligned(self) -> bool:
        return self.pos % 8 == 0

    def more_rbsp_data(self) -> bool:
        """True unless only the rbsp_trailing_bits (1 then zeros) remain."""
        last = len(self.d) - 1
        while last >= 0 and self.d[last] == 0:
            last -= 1
        if last < 0:
            return False
        stop_bit = last * 8 + 7 - ((self.d[last] & -self.d[last]).bit_length() - 1)
        return self.pos < stop_bit

    def trailing(self):
        if self.u(1) != 1:
            raise ValueError("rbsp_stop_one_bit missing")
        while not self.byte_aligned():
            if self.u(1) != 0:
                raise ValueError("non-zero alignment bit after the stop bit")
        if self.pos != len(self.d) * 8:
            raise ValueError("bytes after rbsp_trailing_bits")


def unescape(nal_payload: bytes) -> bytes:
    """Remove emulation_prevention_three_byte (§7.3.1); reject forbidden byte patterns."""
    out, zeros, i = bytearray(), 0, 0
    while i < len(nal_payload):
        b = nal_payload[i]
        if zeros >= 2:
            if b == 3:
                if i + 1 < len(nal_payload) and nal_payload[i + 1] > 3:
                    raise ValueError("emulation prevention byte not followed by 00..03")
                zeros, i = 0, i + 1
                continue
            if b <= 2:
                raise ValueError(f"forbidden sequence 00 00 {b:02x} inside a NAL unit")
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
        i += 1
    return bytes(out)


def parse_sps(rbsp: bytes) -> dict:
    r = BitReader(rbsp)
    s = {"profile_idc": r.u(8), "constraint_flags": r.u(8), "level_idc": r.u(8), "sps_id": r.ue()}
    if s["profile_idc"] in (100, 110, 122, 244, 44, 83, 86, 118, 128):
        raise ValueError("high-profile SPS fields are outside the subset")
    s["log2_max_frame_num"] = r.ue() + 4
    s["poc_type"] = r.ue()
    if s["poc_type"] == 0:
        s["log2_max_poc_lsb"] = r.ue() + 4
    elif s["poc_type"] == 1:
        raise ValueError("pic_order_cnt_type 1 is outside the subset")
    s["max_num_ref_frames"] = r.ue()
    s["gaps_allowed"] = r.u(1)
    s["mb_w"] = r.ue() + 1
    s["mb_h"] = r.ue() + 1
    s["frame_mbs_only"] = r.u(1)
    if not s["frame_mbs_only"]:
        raise ValueError("interlaced streams are outside the subset")
    s["direct_8x8"] = r.u(1)
    s["crop"] = (0, 0, 0, 0)
    if r.u(1):
        s["crop"] = (r.ue(), r.ue(), r.ue(), r.ue())          # left, right, top, bottom (units of 2 luma samples in 4:2:0)
    s["vui"] = None
    if r.u(1):
        v = {}
        if r.u(1):                                             # aspect_ratio_info_present_flag
            if r.u(8) == 255:
                r.u(16); r.u(16)
        if r.u(1):                                             # overscan_info_present_flag
            r.u(1)
        if r.u(1):                                             # video_signal_type_present_flag
            v["video_format"], v["full_range"] = r.u(3), r.u(1)
            if r.u(1):
                v["primaries"], v["transfer"], v["matrix"] = r.u(8), r.u(8), r.u(8)
        if r.u(1):                                             # chroma_loc_info_present_flag
            r.ue(); r.ue()
        if r.u(1):                                             # timing_info_present_flag
            v["num_units_in_tick"], v["time_scale"], v["fixed_frame_rate"] = r.u(32), r.u(32), r.u(1)
        if r.u(1) or r.u(1):                                   # nal / vcl hrd parameters
            raise ValueError("HRD parameters are outside the subset")
        v["pic_struct_present"] = r.u(1)
        if r.u(1):                                             # bitstream_restriction_flag
            v["mv_over_boundaries"] = r.u(1)
            v["max_bytes_per_pic_denom"], v["max_bits_per_mb_denom"] = r.ue(), r.ue()
            v["log2_max_mv_h"], v["log2_max_mv_v"] = r.ue(), r.ue()
            v["max_num_reorder_frames"], v["max_dec_frame_buffering"] = r.ue(), r.ue()
        s["vui"] = v
    r.trailing()
    s["width"] = s["mb_w"] * 16 - 2 * (s["crop"][0] + s["crop"][1])
    s["height"] = s["mb_h"] * 16 - 2 * (s["crop"][2] + s["crop"][3])
    return s


def parse_pps(rbsp: bytes) -> dict:
    r = BitReader(rbsp)
    p = {"pps_id": r.ue(), "sps_id": r.ue(), "cabac": r.u(1), "bottom_field_poc": r.u(1), "slice_groups": r.ue() + 1}
    if p["slice_groups"] != 1:
        raise ValueError("slice groups are outside the subset")
    p["num_ref_l0"], p["num_ref_l1"] = r.ue() + 1, r.ue() + 1
    p["weighted_pred"], p["weighted_bipred"] = r.u(1), r.u(2)
    p["init_qp"], p["init_qs"], p["chroma_qp_offset"] = r.se() + 26, r.se() + 26, r.se()
    p["deblocking_control"], p["constrained_intra"], p["redundant_pic_cnt"] = r.u(1), r.u(1), r.u(1)
    if r.more_rbsp_data():
        raise ValueError("PPS extension fields are outside the subset")
    r.trailing()
    return p


def decode_idr_pcm_slice(rbsp: bytes, nal_ref_idc: int, sps: dict, pps: dict):
    """-> (header dict, Y, Cb, Cr) for an IDR picture coded as ONE I slice of I_PCM macroblocks."""
    r = BitReader(rbsp)
    h = {"first_mb": r.ue(), "slice_type": r.ue(), "pps_id": r.ue(), "frame_num": r.u(sps["log2_max_frame_num"])}
    if h["slice_type"] % 5 != 2:
        raise ValueError("only I slices are in the subset")
    h["idr_pic_id"] = r.ue()                                   # IdrPicFlag (nal_unit_type 5)
    if sps["poc_type"] == 0:
        h["poc_lsb"] = r.u(sps["log2_max_poc_lsb"])
    if pps["redundant_pic_cnt"]:
        r.ue()
    if nal_ref_idc == 0:
        raise ValueError("an IDR picture must have nal_ref_idc != 0")
    h["no_output_of_prior_pics"], h["long_term_reference"] = r.u(1), r.u(1)      # dec_ref_pic_marking of an IDR picture
    h["slice_qp"] = pps["init_qp"] + r.se()
    if pps["deblocking_control"]:
        h["disable_deblocking"] = r.ue()
        if h["disable_deblocking"] != 1:
            r.se(); r.se()
    if pps["cabac"]:
        raise ValueError("CABAC slice data is outside the subset")
    if h["first_mb"] != 0 or h["frame_num"] != 0:
        raise ValueError("one slice per IDR picture, frame_num 0")
    mbw, mbh = sps["mb_w"], sps["mb_h"]
    y = np.empty((mbh * 16, mbw * 16), np.uint8)
    cb = np.empty((mbh * 8, mbw * 8), np.uint8)
    cr = np.empty((mbh * 8, mbw * 8), np.uint8)
    for mb in range(mbw * mbh):
        mb_type = r.ue()
        if mb_type != 25:
            raise ValueError(f"macroblock {mb}: mb_type {mb_type}, only I_PCM (25) is in the subset")
        while not r.byte_aligned():
            if r.u(1) != 0:
                raise ValueError("pcm_alignment_zero_bit is not zero")
        o = r.pos >> 3
        blk = np.frombuffer(r.d, np.uint8, 384, o)
        if sps["profile_idc"] not in (100, 110, 122, 244) and (blk == 0).any():
            raise ValueError("pcm sample equal to 0 (forbidden outside the High profiles)")
        r.pos += 384 * 8
        my, mx = divmod(mb, mbw)
        y[my * 16:(my + 1) * 16, mx * 16:(mx + 1) * 16] = blk[:256].reshape(16, 16)
        cb[my * 8:(my + 1) * 8, mx * 8:(mx + 1) * 8] = blk[256:320].reshape(8, 8)
        cr[my * 8:(my + 1) * 8, mx * 8:(mx + 1) * 8] = blk[320:384].reshape(8, 8)
    r.trailing()                                               # rbsp_slice_trailing_bits
    c = sps["crop"]
    y = y[2 * c[2]: y.shape[0] - 2 * c[3], 2 * c[0]: y.shape[1] - 2 * c[1]]
    cb = cb[c[2]: cb.shape[0] - c[3], c[0]: cb.shape[1] - c[1]]
    cr = cr[c[2]: cr.shape[0] - c[3], c[0]: cr.shape[1] - c[1]]
    return h, y, cb, cr


# ---- ISO base media file: the boxes an `avc1` track needs ----------------------------------------------------------
def _children(buf, start, end):
    pos = start
    while pos + 8 <= end:
        size, kind = struct.unpack_from(">I4s", buf, pos)
        hdr = 8
        if size == 1:
            size, hdr = struct.unpack_from(">Q", buf, pos + 8)[0], 16
        if size < hdr or pos + size > end:
            raise ValueError(f"corrupt box {kind!r} at {pos}")
        yield kind, pos + hdr, pos + size
        pos += size


def _find(buf, path, start=0, end=None):
    end = len(buf) if end is None else end
    for kind, s, e in _children(buf, start, end):
        if kind == path[0]:
            return (s, e) if len(path) == 1 else _find(buf, path[1:], s, e)
    raise KeyError(path[0])


def read_avc_mp4(path: str):
    """-> dict(sps, pps, headers, frames [(Y, Cb, Cr)], fps, width, height, brands) of an mp4 with one avc1 track."""
    buf = open(path, "rb").read()
    tops = [k for k, _, _ in _children(buf, 0, len(buf))]
    if tops != [b"ftyp", b"mdat", b"moov"]:
        raise ValueError(f"top-level boxes {tops}")
    s, e = _find(buf, [b"ftyp"])
    brands = [buf[s:s + 4]] + [buf[i:i + 4] for i in range(s + 8, e, 4)]
    s, _ = _find(buf, [b"moov", b"trak", b"mdia", b"mdhd"])
    timescale, _dur = struct.unpack_from(">II", buf, s + 12)
    stbl = _find(buf, [b"moov", b"trak", b"mdia", b"minf", b"stbl"])
    s, _ = _find(buf, [b"stts"], *stbl)
    entries, count, delta = struct.unpack_from(">III", buf, s + 4)
    s, _ = _find(buf, [b"stsz"], *stbl)
    uniform, n = struct.unpack_from(">II", buf, s + 4)
    sizes = struct.unpack_from(f">{n}I", buf, s + 12) if uniform == 0 else (uniform,) * n
    try:
        s, _ = _find(buf, [b"stco"], *stbl)
        off = struct.unpack_from(">I", buf, s + 8)[0]
    except KeyError:
        s, _ = _find(buf, [b"co64"], *stbl)
        off = struct.unpack_from(">Q", buf, s + 8)[0]
    has_stss = any(k == b"stss" for k, _, _ in _children(buf, *stbl))
    s, e = _find(buf, [b"stsd"], *stbl)
    entry_size, codec = struct.unpack_from(">I4s", buf, s + 8)
    if codec != b"avc1" or entries != 1 or count != n:
        raise ValueError(f"sample entry {codec!r}, stts entries {entries}, {count} vs {n} samples")
    width, height = struct.unpack_from(">HH", buf, s + 16 + 24)
    a, ae = _find(buf, [b"avcC"], s + 16 + 78, s + 8 + entry_size)
    version, profile, compat, level, len_size, n_sps = struct.unpack_from(">6B", buf, a)
    if version != 1 or (len_size & 3) != 3 or (n_sps & 31) != 1:
        raise ValueError("avcC: one SPS, 4-byte NAL lengths expected")
    (l_sps,) = struct.unpack_from(">H", buf, a + 6)
    sps_nal = buf[a + 8: a + 8 + l_sps]
    n_pps = buf[a + 8 + l_sps]
    (l_pps,) = struct.unpack_from(">H", buf, a + 9 + l_sps)
    pps_nal = buf[a + 11 + l_sps: a + 11 + l_sps + l_pps]
    if n_pps != 1 or a + 11 + l_sps + l_pps != ae:
        raise ValueError("avcC: one PPS expected, no trailing bytes")
    for nal, typ in ((sps_nal, 7), (pps_nal, 8)):
        if nal[0] & 0x80 or (nal[0] & 31) != typ:
            raise ValueError(f"parameter set NAL header {nal[0]:#x}")
    sps, pps = parse_sps(unescape(sps_nal[1:])), parse_pps(unescape(pps_nal[1:]))
    if (profile, compat, level) != (sps["profile_idc"], sps["constraint_flags"], sps["level_idc"]):
        raise ValueError("avcC profile / level bytes differ from the SPS")
    headers, frames = [], []
    for sz in sizes:
        (ln,) = struct.unpack_from(">I", buf, off)
        if ln + 4 != sz:
            raise ValueError("one length-prefixed NAL unit per sample expected")
        nal = buf[off + 4: off + 4 + ln]
        if nal[0] & 0x80 or (nal[0] & 31) != 5:
            raise ValueError(f"sample is not an IDR slice (NAL header {nal[0]:#x})")
        h, y, cb, cr = decode_idr_pcm_slice(unescape(nal[1:]), (nal[0] >> 5) & 3, sps, pps)
        headers.append(h)
        frames.append((y, cb, cr))
        off += sz
    return dict(sps=sps, pps=pps, headers=headers, frames=frames, fps=timescale / delta, width=width, height=height, brands=brands,
                has_stss=has_stss)


def yuv420_to_rgb(y, cb, cr):
    """BT.601 limited range, nearest-neighbour chroma upsampling (what a plain player does)."""
    yf = (y.astype(np.float32) - 16.0) * (255.0 / 219.0)
    cbf = np.repeat(np.repeat(cb.astype(np.float32) - 128.0, 2, 0), 2, 1) * (255.0 / 224.0)
    crf = np.repeat(np.repeat(cr.astype(np.float32) - 128.0, 2, 0), 2, 1) * (255.0 / 224.0)
    r = yf + 1.402 * crf
    g = yf - 0.344136 * cbf - 0.714136 * crf
    b = yf + 1.772 * cbf
    return np.clip(np.rint(np.stack([r, g, b], -1)), 0, 255).astype(np.uint8)
