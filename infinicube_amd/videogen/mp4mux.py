"""Dependency-free mp4 writer (Motion-JPEG in an ISO base-media file) — the fallback of ``io.save_video``.

The reference hands the frames to diffsynth's ``save_video`` = imageio + ffmpeg/libx264
[R infinicube/videogen/inference.py:229-232], and stage 3 later opens the file by path
[R infinicube/inference/scene_gaussian_generation.py:290-293].  ``io.save_video`` does the same whenever imageio
is importable.  Where it is not (this image, the GPU boxes), the contract "an mp4 exists at ``output_path`` on
return" is kept with this writer instead of raising: every frame is a baseline JPEG (PIL), muxed the way
``ffmpeg -c:v mjpeg out.mp4`` does it — one video track, sample entry ``mp4v`` whose ``esds`` carries
objectTypeIndication 0x6C (ISO/IEC 10918-1 JPEG), constant frame duration.  Any ffmpeg-based reader (imageio,
decord, cv2) decodes it; it is intra-only, so larger than libx264 output.  Not bit-comparable with the
reference's file (different codec) — a documented difference, printed when the fallback is taken.
"""

from __future__ import annotations

import io as _io
import struct
from typing import List, Sequence, Tuple

_MATRIX = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


def _box(kind: bytes, *payload: bytes) -> bytes:
    body = b"".join(payload)
    return struct.pack(">I4s", 8 + len(body), kind) + body


def _full(kind: bytes, version: int, flags: int, *payload: bytes) -> bytes:
    return _box(kind, struct.pack(">I", (version << 24) | flags), *payload)


def _descr(tag: int, body: bytes) -> bytes:
    n = len(body)
    return bytes([tag, 0x80 | (n >> 21) & 0x7F, 0x80 | (n >> 14) & 0x7F, 0x80 | (n >> 7) & 0x7F, n & 0x7F]) + body


def jpeg_quality(quality: int) -> int:
    """imageio's 0..10 quality scale -> PIL JPEG quality (8 -> 90)."""
    return max(1, min(95, 50 + 5 * int(quality)))


def encode_jpeg_frames(frames: Sequence, quality: int = 8) -> Tuple[List[bytes], int, int]:
    out, size = [], None
    for fr in frames:
        im = fr.convert("RGB")
        if size is None:
            size = im.size
        elif im.size != size:
            raise ValueError(f"frame size changed inside the clip: {im.size} != {size}")
        buf = _io.BytesIO()
        im.save(buf, format="JPEG", quality=jpeg_quality(quality), subsampling="4:2:0", optimize=False)
        out.append(buf.getvalue())
    if not out:
        raise ValueError("no frames to write")
    return out, size[0], size[1]


def mux_mjpeg_mp4(samples: Sequence[bytes], width: int, height: int, fps: float) -> bytes:
    """JPEG samples -> bytes of an .mp4 (ftyp, mdat, moov; one chunk holding every sample)."""
    n = len(samples)
    timescale = int(round(fps * 1000))
    delta = 1000
    duration = n * delta
    ftyp = _box(b"ftyp", b"isom", struct.pack(">I", 0x200), b"isomiso2mp41")
    mdat_payload = b"".join(samples)
    big = len(mdat_payload) + 8 >= 1 << 32
    mdat = (struct.pack(">I4sQ", 1, b"mdat", 16 + len(mdat_payload)) if big else struct.pack(">I4s", 8 + len(mdat_payload), b"mdat")) + mdat_payload
    first_sample = len(ftyp) + (16 if big else 8)

    maxrate = max(len(s) for s in samples) * 8 * fps
    avgrate = len(mdat_payload) * 8 * fps / n
    dec_cfg = _descr(4, struct.pack(">BB", 0x6C, 0x11) + struct.pack(">I", max(len(s) for s in samples))[1:] +
                     struct.pack(">II", int(maxrate), int(avgrate)))
    esds = _full(b"esds", 0, 0, _descr(3, struct.pack(">HB", 1, 0) + dec_cfg + _descr(6, b"\x02")))
    entry = _box(b"mp4v", b"\0" * 6, struct.pack(">H", 1), b"\0" * 16, struct.pack(">HH", width, height),
                 struct.pack(">II", 0x00480000, 0x00480000), b"\0" * 4, struct.pack(">H", 1), b"\0" * 32,
                 struct.pack(">Hh", 0x0018, -1), esds)
    stbl = _box(b"stbl",
                _full(b"stsd", 0, 0, struct.pack(">I", 1), entry),
                _full(b"stts", 0, 0, struct.pack(">III", 1, n, delta)),
                _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n, 1)),
                _full(b"stsz", 0, 0, struct.pack(">II", 0, n), b"".join(struct.pack(">I", len(s)) for s in samples)),
                _full(b"co64", 0, 0, struct.pack(">IQ", 1, first_sample)) if big else
                _full(b"stco", 0, 0, struct.pack(">II", 1, first_sample)))
    minf = _box(b"minf", _full(b"vmhd", 0, 1, b"\0" * 8),
                _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1), _full(b"url ", 0, 1))), stbl)
    mdia = _box(b"mdia", _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, timescale, duration, 0x55C4, 0)),
                _full(b"hdlr", 0, 0, b"\0" * 4, b"vide", b"\0" * 12, b"VideoHandler\0"), minf)
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, duration), b"\0" * 8, struct.pack(">hhhH", 0, 0, 0, 0),
                 _MATRIX, struct.pack(">II", width << 16, height << 16))
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, timescale, duration), struct.pack(">IH", 0x10000, 0x100),
                 b"\0" * 10, _MATRIX, b"\0" * 24, struct.pack(">I", 2))
    return ftyp + mdat + _box(b"moov", mvhd, _box(b"trak", tkhd, mdia))


def write_mjpeg_mp4(frames: Sequence, path: str, fps: float = 10, quality: int = 8) -> None:
    samples, w, h = encode_jpeg_frames(frames, quality)
    data = mux_mjpeg_mp4(samples, w, h, fps)
    with open(path, "wb") as f:
        f.write(data)


# ---- reader for the files this module writes (tests, and a way to load them back without ffmpeg) ---------------
def _children(buf: bytes, start: int, end: int):
    pos = start
    while pos + 8 <= end:
        size, kind = struct.unpack_from(">I4s", buf, pos)
        hdr = 8
        if size == 1:
            size = struct.unpack_from(">Q", buf, pos + 8)[0]
            hdr = 16
        if size < hdr or pos + size > end:
            raise ValueError(f"corrupt box {kind!r} at {pos}")
        yield kind, pos + hdr, pos + size
        pos += size


def _find(buf: bytes, path: Sequence[bytes], start: int = 0, end: int = None):
    end = len(buf) if end is None else end
    for kind, s, e in _children(buf, start, end):
        if kind == path[0]:
            return (s, e) if len(path) == 1 else _find(buf, path[1:], s, e)
    raise KeyError(path[0])


def read_mjpeg_mp4(path: str):
    """-> (list of PIL frames, fps, (width, height)) for a file written by ``write_mjpeg_mp4``."""
    from PIL import Image
    buf = open(path, "rb").read()
    s, _ = _find(buf, [b"moov", b"trak", b"mdia", b"mdhd"])
    timescale, _dur = struct.unpack_from(">II", buf, s + 12)
    stbl = _find(buf, [b"moov", b"trak", b"mdia", b"minf", b"stbl"])
    s, _ = _find(buf, [b"stts"], *stbl)
    _, count, delta = struct.unpack_from(">III", buf, s + 4)
    s, _ = _find(buf, [b"stsz"], *stbl)
    _, n = struct.unpack_from(">II", buf, s + 4)
    sizes = struct.unpack_from(f">{n}I", buf, s + 12)
    try:
        s, _ = _find(buf, [b"stco"], *stbl)
        off = struct.unpack_from(">I", buf, s + 8)[0]
    except KeyError:
        s, _ = _find(buf, [b"co64"], *stbl)
        off = struct.unpack_from(">Q", buf, s + 8)[0]
    s, _ = _find(buf, [b"stsd"], *stbl)
    codec = buf[s + 12: s + 16]
    width, height = struct.unpack_from(">HH", buf, s + 8 + 8 + 24)
    assert count == n and codec == b"mp4v"
    frames = []
    for sz in sizes:
        frames.append(Image.open(_io.BytesIO(buf[off: off + sz])).convert("RGB"))
        off += sz
    return frames, timescale / delta, (width, height)
