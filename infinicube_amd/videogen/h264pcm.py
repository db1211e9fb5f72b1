"""Dependency-free H.264/AVC writer for ``io.save_video`` where imageio / ffmpeg are absent (this image, the GPU boxes).

The reference's mp4 is H.264 (libx264 through imageio-ffmpeg) [R infinicube/videogen/inference.py:229-232;
R infinicube/utils/fileio_utils.py:58-140] and stage 3 opens it by path with an ffmpeg-class reader
[R infinicube/inference/scene_gaussian_generation.py:290-293].  This writer keeps the CODEC and the CONTAINER — an
``avc1`` track in an ISO base-media file that any H.264 decoder plays — with the one macroblock type that needs no
transform, quantiser or entropy-coder tables: every picture is an IDR picture, every macroblock is I_PCM
(ITU-T H.264 §7.3.5: mb_type 25 in an I slice, then 256 luma + 64 Cb + 64 Cr raw samples).  Constrained Baseline
profile, CAVLC, one slice per picture, deblocking disabled in the slice header.  The stream is LOSSLESS in 4:2:0
(the only loss is the RGB -> BT.601 limited-range YCbCr 4:2:0 conversion every libx264 mp4 goes through too), so the
file is large: 1.5 bytes per pixel per frame (93 x 480 x 832: 55.7 MB; libx264 at the reference's settings: a few
MB).  ``ICV_MP4_CODEC=mjpeg`` selects the smaller Motion-JPEG file of mp4mux.py instead.

No decoder exists in this image to cross-check against (no ffmpeg / libav / cv2 / decord — probed), so the stream is
verified by an independent parser written from the syntax tables of the standard (tests/h264_subset_decoder.py):
NAL framing and emulation prevention, every SPS / VUI / PPS / slice-header field, macroblock alignment, and the
exact reconstruction of the input YCbCr planes.
"""
from __future__ import annotations

import re
import struct
from typing import List, Sequence, Tuple

import numpy as np

from .mp4mux import _MATRIX, _box, _full


class _Bits:
    """MSB-first bit writer (H.264 §7.2: u(n), ue(v), se(v))."""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def u(self, nbits: int, value: int):
        assert 0 <= value < (1 << nbits)
        self.acc = (self.acc << nbits) | value
        self.n += nbits
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def ue(self, value: int):
        assert value >= 0
        v = value + 1
        nb = v.bit_length()
        self.u(nb - 1, 0)
        self.u(nb, v)

    def se(self, value: int):
        self.ue(2 * value - 1 if value > 0 else -2 * value)

    def align_zero(self):
        if self.n:
            self.u(8 - self.n, 0)

    def trailing(self):                      # rbsp_trailing_bits: stop bit, then zeros to the byte boundary
        self.u(1, 1)
        self.align_zero()

    def bytes(self) -> bytes:
        assert self.n == 0, "not byte aligned"
        return bytes(self.out)


_EPB = re.compile(rb"\x00\x00(?=[\x00-\x03])")


def _nal(ref_idc: int, unit_type: int, rbsp: bytes) -> bytes:
    """NAL unit = header byte + RBSP with emulation-prevention bytes (§7.4.1): 00 00 0x -> 00 00 03 0x; a trailing 00
    would also need one (cannot happen: every RBSP here ends in the non-zero trailing-bits byte)."""
    return bytes([(ref_idc << 5) | unit_type]) + _EPB.sub(b"\x00\x00\x03", rbsp)


def sps_rbsp(width: int, height: int, fps: float) -> bytes:
    mbw, mbh = (width + 15) // 16, (height + 15) // 16
    b = _Bits()
    b.u(8, 66)                 # profile_idc: Baseline
    b.u(8, 0xC0)               # constraint_set0_flag, constraint_set1_flag (Constrained Baseline), rest 0
    # level: raw samples exceed the bit rates of the lower levels, and Annex A.3.1 bounds the bytes of one access unit by
    # 384 * max(PicSizeInMbs, MaxMBPS / 172) / MinCR: an all-PCM picture (384 B per macroblock + the slice header) fits under
    # level 5.1 (MaxMBPS 983 040 -> 1.10 MB) up to ~2800 macroblocks (93 x 480 x 832: 1560) and under 5.2 (2.31 MB) beyond
    # (720 x 1280: 3600)
    b.u(8, 51 if mbw * mbh <= 2800 else 52)
    b.ue(0)                    # seq_parameter_set_id
    b.ue(0)                    # log2_max_frame_num_minus4
    b.ue(2)                    # pic_order_cnt_type 2: output order = decoding order
    b.ue(0)                    # max_num_ref_frames: every picture is IDR
    b.u(1, 0)                  # gaps_in_frame_num_value_allowed_flag
    b.ue(mbw - 1)              # pic_width_in_mbs_minus1
    b.ue(mbh - 1)              # pic_height_in_map_units_minus1
    b.u(1, 1)                  # frame_mbs_only_flag
    b.u(1, 1)                  # direct_8x8_inference_flag
    crop_r, crop_b = mbw * 16 - width, mbh * 16 - height
    if crop_r or crop_b:
        assert crop_r % 2 == 0 and crop_b % 2 == 0, "4:2:0 cropping works in units of two luma samples"
        b.u(1, 1)
        b.ue(0); b.ue(crop_r // 2); b.ue(0); b.ue(crop_b // 2)   # left, right, top, bottom
    else:
        b.u(1, 0)
    b.u(1, 1)                  # vui_parameters_present_flag
    b.u(1, 0)                  # aspect_ratio_info_present_flag
    b.u(1, 0)                  # overscan_info_present_flag
    b.u(1, 1)                  # video_signal_type_present_flag
    b.u(3, 5)                  # video_format: unspecified
    b.u(1, 0)                  # video_full_range_flag: limited range
    b.u(1, 1)                  # colour_description_present_flag
    b.u(8, 6); b.u(8, 6); b.u(8, 6)   # primaries / transfer / matrix: SMPTE 170M (BT.601), what rgb_to_yuv420 applies
    b.u(1, 0)                  # chroma_loc_info_present_flag
    b.u(1, 1)                  # timing_info_present_flag
    ticks = 1000
    b.u(32, ticks)             # num_units_in_tick
    b.u(32, int(round(2 * fps * ticks)))   # time_scale: frame rate = time_scale / (2 num_units_in_tick)
    b.u(1, 1)                  # fixed_frame_rate_flag
    b.u(1, 0)                  # nal_hrd_parameters_present_flag
    b.u(1, 0)                  # vcl_hrd_parameters_present_flag
    b.u(1, 0)                  # pic_struct_present_flag
    b.u(1, 1)                  # bitstream_restriction_flag
    b.u(1, 1)                  # motion_vectors_over_pic_boundaries_flag
    b.ue(0); b.ue(0)           # max_bytes_per_pic_denom, max_bits_per_mb_denom: unlimited
    b.ue(16); b.ue(16)         # log2_max_mv_length_horizontal / vertical
    b.ue(0)                    # max_num_reorder_frames
    b.ue(0)                    # max_dec_frame_buffering
    b.trailing()
    return b.bytes()


def pps_rbsp() -> bytes:
    b = _Bits()
    b.ue(0); b.ue(0)           # pic_parameter_set_id, seq_parameter_set_id
    b.u(1, 0)                  # entropy_coding_mode_flag: CAVLC
    b.u(1, 0)                  # bottom_field_pic_order_in_frame_present_flag
    b.ue(0)                    # num_slice_groups_minus1
    b.ue(0); b.ue(0)           # num_ref_idx_l0 / l1_default_active_minus1
    b.u(1, 0); b.u(2, 0)       # weighted_pred_flag, weighted_bipred_idc
    b.se(0); b.se(0); b.se(0)  # pic_init_qp_minus26, pic_init_qs_minus26, chroma_qp_index_offset
    b.u(1, 1)                  # deblocking_filter_control_present_flag
    b.u(1, 0)                  # constrained_intra_pred_flag
    b.u(1, 0)                  # redundant_pic_cnt_present_flag
    b.trailing()
    return b.bytes()


def rgb_to_yuv420(rgb: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """uint8 [H, W, 3] -> (Y [H, W], Cb [H/2, W/2], Cr [H/2, W/2]) uint8, BT.601 limited range, chroma = mean of each
    2 x 2 block (H, W even)."""
    f = rgb.astype(np.float32)
    r, g, bl = f[..., 0], f[..., 1], f[..., 2]
    y = 16.0 + (65.481 * r + 128.553 * g + 24.966 * bl) / 255.0
    cb = 128.0 + (-37.797 * r - 74.203 * g + 112.0 * bl) / 255.0
    cr = 128.0 + (112.0 * r - 93.786 * g - 18.214 * bl) / 255.0
    h, w = y.shape
    pool = lambda p: p.reshape(h // 2, 2, w // 2, 2).mean(axis=(1, 3))   # noqa: E731
    # [1, 255]: outside the High profiles a PCM sample must not be 0 (H.264 §7.4.5); limited-range values never are
    q = lambda p: np.clip(np.rint(p), 1, 255).astype(np.uint8)          # noqa: E731
    return q(y), q(pool(cb)), q(pool(cr))


def _pad_plane(p: np.ndarray, h: int, w: int) -> np.ndarray:
    if p.shape == (h, w):
        return p
    return np.pad(p, ((0, h - p.shape[0]), (0, w - p.shape[1])), mode="edge")


def idr_slice_nal(y: np.ndarray, cb: np.ndarray, cr: np.ndarray, idr_pic_id: int) -> bytes:
    """One IDR picture = one I slice of I_PCM macroblocks."""
    mbh, mbw = (y.shape[0] + 15) // 16, (y.shape[1] + 15) // 16
    y, cb, cr = _pad_plane(y, mbh * 16, mbw * 16), _pad_plane(cb, mbh * 8, mbw * 8), _pad_plane(cr, mbh * 8, mbw * 8)
    b = _Bits()
    b.ue(0)                    # first_mb_in_slice
    b.ue(7)                    # slice_type 7: I, and every slice of the picture is I
    b.ue(0)                    # pic_parameter_set_id
    b.u(4, 0)                  # frame_num (log2_max_frame_num = 4): 0 in an IDR picture
    b.ue(idr_pic_id)           # consecutive IDR pictures must differ here
    b.u(1, 0); b.u(1, 0)       # dec_ref_pic_marking: no_output_of_prior_pics_flag, long_term_reference_flag
    b.se(0)                    # slice_qp_delta
    b.ue(1)                    # disable_deblocking_filter_idc 1: off
    b.ue(25)                   # first macroblock: mb_type 25 = I_PCM ...
    b.align_zero()             # ... pcm_alignment_zero_bit up to the byte boundary
    head = b.bytes()
    n = mbh * mbw
    mb = np.empty((n, 2 + 384), dtype=np.uint8)
    mb[:, 0], mb[:, 1] = 0x0D, 0x00       # ue(25) = 0000 11010 then 7 alignment zeros: later macroblocks start byte aligned
    mb[:, 2:258] = y.reshape(mbh, 16, mbw, 16).transpose(0, 2, 1, 3).reshape(n, 256)
    mb[:, 258:322] = cb.reshape(mbh, 8, mbw, 8).transpose(0, 2, 1, 3).reshape(n, 64)
    mb[:, 322:386] = cr.reshape(mbh, 8, mbw, 8).transpose(0, 2, 1, 3).reshape(n, 64)
    body = mb.reshape(-1)[2:].tobytes()   # the first macroblock's mb_type sits in `head`
    return _nal(3, 5, head + body + b"\x80")          # rbsp_slice_trailing_bits


def encode_h264_pcm(frames: Sequence, fps: float) -> Tuple[bytes, bytes, List[bytes], int, int]:
    """PIL frames -> (SPS NAL, PPS NAL, [IDR slice NAL per frame], width, height)."""
    size, slices = None, []
    for i, fr in enumerate(frames):
        im = fr.convert("RGB")
        if size is None:
            size = im.size
            if size[0] % 2 or size[1] % 2:
                raise ValueError(f"4:2:0 video needs even dimensions, got {size}")
        elif im.size != size:
            raise ValueError(f"frame size changed inside the clip: {im.size} != {size}")
        slices.append(idr_slice_nal(*rgb_to_yuv420(np.asarray(im, dtype=np.uint8)), idr_pic_id=i & 1))
    if not slices:
        raise ValueError("no frames to write")
    return _nal(3, 7, sps_rbsp(size[0], size[1], fps)), _nal(3, 8, pps_rbsp()), slices, size[0], size[1]


def mux_avc_mp4(sps: bytes, pps: bytes, slices: Sequence[bytes], width: int, height: int, fps: float) -> bytes:
    """-> bytes of an .mp4: ftyp, mdat (4-byte length-prefixed NAL units), moov with one `avc1` track."""
    n = len(slices)
    timescale, delta = int(round(fps * 1000)), 1000
    duration = n * delta
    ftyp = _box(b"ftyp", b"isom", struct.pack(">I", 0x200), b"isomiso2avc1mp41")
    samples = [struct.pack(">I", len(s)) + s for s in slices]
    payload = b"".join(samples)
    big = len(payload) + 8 >= 1 << 32
    mdat = (struct.pack(">I4sQ", 1, b"mdat", 16 + len(payload)) if big else struct.pack(">I4s", 8 + len(payload), b"mdat")) + payload
    first_sample = len(ftyp) + (16 if big else 8)
    avcc = _box(b"avcC", bytes([1, sps[1], sps[2], sps[3], 0xFF, 0xE1]), struct.pack(">H", len(sps)), sps,
                bytes([1]), struct.pack(">H", len(pps)), pps)
    entry = _box(b"avc1", b"\0" * 6, struct.pack(">H", 1), b"\0" * 16, struct.pack(">HH", width, height),
                 struct.pack(">II", 0x00480000, 0x00480000), b"\0" * 4, struct.pack(">H", 1), b"\0" * 32,
                 struct.pack(">Hh", 0x0018, -1), avcc)
    stbl = _box(b"stbl",
                _full(b"stsd", 0, 0, struct.pack(">I", 1), entry),
                _full(b"stts", 0, 0, struct.pack(">III", 1, n, delta)),
                _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n, 1)),
                _full(b"stsz", 0, 0, struct.pack(">II", 0, n), b"".join(struct.pack(">I", len(s)) for s in samples)),
                _full(b"co64", 0, 0, struct.pack(">IQ", 1, first_sample)) if big else
                _full(b"stco", 0, 0, struct.pack(">II", 1, first_sample)))
    # no stss box: every sample is a sync sample (all pictures are IDR)
    minf = _box(b"minf", _full(b"vmhd", 0, 1, b"\0" * 8),
                _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1), _full(b"url ", 0, 1))), stbl)
    mdia = _box(b"mdia", _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, timescale, duration, 0x55C4, 0)),
                _full(b"hdlr", 0, 0, b"\0" * 4, b"vide", b"\0" * 12, b"VideoHandler\0"), minf)
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, duration), b"\0" * 8, struct.pack(">hhhH", 0, 0, 0, 0),
                 _MATRIX, struct.pack(">II", width << 16, height << 16))
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration), struct.pack(">IH", 0x10000, 0x100),
                 b"\0" * 10, _MATRIX, b"\0" * 24, struct.pack(">I", 2))
    return ftyp + mdat + _box(b"moov", mvhd, _box(b"trak", tkhd, mdia))


def write_h264_mp4(frames: Sequence, path: str, fps: float = 10) -> None:
    sps, pps, slices, w, h = encode_h264_pcm(frames, fps)
    with open(path, "wb") as f:
        f.write(mux_avc_mp4(sps, pps, slices, w, h, fps))
