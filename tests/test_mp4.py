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


def test_save_video_writes_a_parsable_mp4(tmp_path, capsys):
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
