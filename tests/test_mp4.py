"""The dependency-free mp4 fallback of save_video (B-outer side effect: an mp4 exists at output_path on return)."""
import struct

import numpy as np
import pytest
from PIL import Image

from infinicube_amd.videogen import io as vio
from infinicube_amd.videogen import mp4mux


def _frames(n=7, h=48, w=80):
    yy, xx = np.mgrid[0:h, 0:w]
    out = []
    for i in range(n):
        a = np.stack([(xx * 3 + i * 9) % 256, (yy * 5) % 256, np.full_like(xx, 30 * i % 256)], -1).astype(np.uint8)
        out.append(Image.fromarray(a, mode="RGB"))
    return out


def test_save_video_writes_a_parsable_mp4(tmp_path, capsys, monkeypatch):
    monkeypatch.setenv("ICV_MP4_CODEC", "mjpeg")
    frames = _frames()
    path = tmp_path / "sub" / "video_480p_front.mp4"
    vio.save_video(frames, str(path), fps=10, quality=8)
    assert "Motion-JPEG" in capsys.readouterr().out            # the fallback announces itself (no imageio here)
    buf = path.read_bytes()
    assert buf[4:8] == b"ftyp" and buf[8:12] == b"isom"
    # top-level boxes tile the file exactly: ftyp, mdat, moov
    kinds, pos = [], 0
    while pos < len(buf):
        size, kind = struct.unpack_from(">I4s", buf, pos)
        kinds.append(kind); pos += size
    assert kinds == [b"ftyp", b"mdat", b"moov"] and pos == len(buf)
    back, fps, (w, h) = mp4mux.read_mjpeg_mp4(str(path))
    assert len(back) == len(frames) and fps == 10.0 and (w, h) == frames[0].size
    for a, b in zip(frames, back):
        err = np.asarray(a, np.float64) - np.asarray(b, np.float64)
        assert 10 * np.log10(255.0 ** 2 / (err ** 2).mean()) > 28.0      # JPEG q90, 4:2:0 on a synthetic pattern
    # the esds names the JPEG object type, the sample table describes every frame, one chunk at the first sample
    assert b"mp4v" in buf and b"esds" in buf
    i = buf.index(b"esds")
    assert 0x6C in buf[i: i + 40]
    first = buf.index(b"\xff\xd8")
    j = buf.index(b"stco")
    assert struct.unpack_from(">II", buf, j + 8) == (1, first)


def test_mux_rejects_bad_input():
    with pytest.raises(ValueError, match="no frames"):
        mp4mux.encode_jpeg_frames([])
    a, b = _frames(1)[0], _frames(1, 32, 32)[0]
    with pytest.raises(ValueError, match="frame size changed"):
        mp4mux.encode_jpeg_frames([a, b])
    assert mp4mux.jpeg_quality(8) == 90 and mp4mux.jpeg_quality(0) == 50 and mp4mux.jpeg_quality(10) == 95


# ---------------------------------------------------------------------------------------------------------------------
# default writer without imageio: H.264 (Constrained Baseline, IDR pictures of I_PCM macroblocks) in an avc1 track.
# Checked by tests/h264_subset_decoder.py, a parser written from the standard's syntax tables that shares no code with
# the writer (no third-party H.264 decoder exists in this image: ffmpeg, libav*, cv2, decord, imageio, av, gstreamer
# were probed and are absent).
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("h,w", [(48, 80), (40, 72), (18, 34)])     # multiples of 16, and two cropped sizes
def test_default_save_video_is_h264_and_round_trips(tmp_path, capsys, monkeypatch, h, w):
    import h264_subset_decoder as H
    from infinicube_amd.videogen import h264pcm
    monkeypatch.delenv("ICV_MP4_CODEC", raising=False)
    frames = _frames(5, h, w)
    frames[1] = Image.fromarray(np.zeros((h, w, 3), np.uint8))           # black: long runs of equal bytes
    frames[2] = Image.fromarray(np.full((h, w, 3), 255, np.uint8))
    path = tmp_path / "video.mp4"
    vio.save_video(frames, str(path), fps=10, quality=8)
    assert "H.264" in capsys.readouterr().out                             # the fallback announces itself
    d = H.read_avc_mp4(str(path))
    assert d["brands"][0] == b"isom" and b"avc1" in d["brands"] and not d["has_stss"]
    assert (d["width"], d["height"], d["fps"]) == (w, h, 10.0) and len(d["frames"]) == 5
    sps, pps = d["sps"], d["pps"]
    assert (sps["profile_idc"], sps["constraint_flags"] & 0xC0, sps["level_idc"]) == (66, 0xC0, 51)
    assert (sps["width"], sps["height"]) == (w, h) and sps["poc_type"] == 2 and sps["max_num_ref_frames"] == 0
    v = sps["vui"]
    assert v["time_scale"] / (2 * v["num_units_in_tick"]) == 10.0 and v["fixed_frame_rate"] == 1
    assert (v["full_range"], v["matrix"]) == (0, 6) and v["max_num_reorder_frames"] == 0
    assert pps["cabac"] == 0 and pps["deblocking_control"] == 1 and pps["init_qp"] == 26
    ids = [hd["idr_pic_id"] for hd in d["headers"]]
    assert all(a != b for a, b in zip(ids, ids[1:])), "consecutive IDR pictures must carry different idr_pic_id"
    assert all(hd["slice_type"] == 7 and hd["disable_deblocking"] == 1 and hd["slice_qp"] == 26 for hd in d["headers"])
    for fr, (y, cb, cr) in zip(frames, d["frames"]):
        wy, wcb, wcr = h264pcm.rgb_to_yuv420(np.asarray(fr))
        assert np.array_equal(y, wy) and np.array_equal(cb, wcb) and np.array_equal(cr, wcr), "the stream is lossless in YCbCr 4:2:0"
        back = H.yuv420_to_rgb(y, cb, cr).astype(np.float64)
        err = np.asarray(fr, np.float64) - back
        assert 10 * np.log10(255.0 ** 2 / max((err ** 2).mean(), 1e-9)) > 24.0   # chroma subsampling of a synthetic stripe pattern
    flat = np.asarray(frames[2], np.float64) - H.yuv420_to_rgb(*d["frames"][2])
    assert np.abs(flat).max() <= 2                                         # a flat frame survives to rounding


def test_h264_emulation_prevention_and_bit_writer():
    import h264_subset_decoder as H
    from infinicube_amd.videogen import h264pcm
    raw = bytes([0, 0, 0, 0, 0, 1, 0, 0, 2, 0, 0, 3, 0, 0, 4, 7, 0, 0, 0x80])
    nal = h264pcm._nal(3, 5, raw)
    assert nal[0] == 0x65 and b"\x00\x00\x00" not in nal[1:] and b"\x00\x00\x01" not in nal[1:] and b"\x00\x00\x02" not in nal[1:]
    assert H.unescape(nal[1:]) == raw
    b = h264pcm._Bits()
    for v in (0, 1, 2, 7, 25, 255, 65535):
        b.ue(v)
    for v in (0, 1, -1, 5, -17):
        b.se(v)
    b.trailing()
    r = H.BitReader(b.bytes())
    assert [r.ue() for _ in range(7)] == [0, 1, 2, 7, 25, 255, 65535] and [r.se() for _ in range(5)] == [0, 1, -1, 5, -17]
    r.trailing()
    with pytest.raises(ValueError, match="forbidden sequence"):
        H.unescape(bytes([1, 0, 0, 1]))
    with pytest.raises(ValueError, match="even dimensions"):
        h264pcm.encode_h264_pcm(_frames(1, 17, 32), 10)


def test_h264_level_follows_the_picture_size():
    import h264_subset_decoder as H
    from infinicube_amd.videogen import h264pcm
    assert H.parse_sps(h264pcm.sps_rbsp(832, 480, 10))["level_idc"] == 51
    big = H.parse_sps(h264pcm.sps_rbsp(1280, 720, 10))
    assert big["level_idc"] == 52 and (big["width"], big["height"]) == (1280, 720) and big["mb_w"] * big["mb_h"] == 3600


def test_unknown_codec_is_refused(tmp_path, monkeypatch):
    monkeypatch.setenv("ICV_MP4_CODEC", "vp9")
    with pytest.raises(ValueError, match="ICV_MP4_CODEC"):
        vio.write_video_without_ffmpeg(_frames(1), str(tmp_path / "x.mp4"))
