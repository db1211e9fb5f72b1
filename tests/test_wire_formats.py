"""the stage-2 artefacts stage 3 reads back [R infinicube/inference/scene_gaussian_generation.py:
286-320].  webdataset / imageio are absent from the build container, so the writers restate webdataset's published tar
framing (wire_formats.py header: parity unpinned for the framing); what IS checkable without them: every member
round-trips through the standard library, numpy payloads are byte-identical to numpy's own serialisation, names /
order / modes follow the reference's keys, and the uint16 quantisation is the caller's `(x * 100).astype(uint16)`."""
import io
import os
import tarfile

import numpy as np
import pytest
import torch

from infinicube_amd.utils import wire_formats as wf


def _scene(n=5, h=48, w=64):
    g = np.random.default_rng(0)
    depth = (g.random((n, h, w), dtype=np.float32) * 120).astype(np.float32)
    depth[:, :6] = 0.0
    depth[0, 10, 10] = 655.3599                # just under the uint16 range after x100
    inst = g.integers(0, 5, (n, h, w)).astype(np.int32)
    inst[1, 20:30, 20:30] = 2 ** 15 + 3
    poses = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    poses[:, 2, 3] = np.arange(n) * 0.5
    sem = g.integers(0, 255, (n, h, w, 3), dtype=np.uint8)
    co = g.integers(0, 255, (n, h, w, 3), dtype=np.uint8)
    return depth, inst, poses, sem, co


def test_tar_members_names_order_and_roundtrip(tmp_path):
    depth, inst, poses, sem, co = _scene()
    du16 = (depth * 100).astype(np.uint16)                      # the caller's conversion, on the host for this CPU test
    intr = np.array([900.0, 900.0, 32.0, 24.0, 64, 48])
    files = wf.write_guidance_buffer_artifacts(tmp_path, "clip0007", depth, inst, poses, intr, list(sem), list(co),
                                               resolution="480p", depth_u16=du16,
                                               depth_colorizer=lambda d: np.repeat((np.clip(d, 0, 127) * 2).astype(np.uint8)[..., None], 3, -1))
    assert {p.name for p in files.values()} == {
        "voxel_depth_100_480p_front.tar", "instance_buffer_480p_front.tar", "pose.tar", "intrinsic.tar",
        "semantic_buffer_video_480p_front.mp4", "coordinate_buffer_video_480p_front.mp4", "depth_vis_video_480p_front.mp4"}
    with tarfile.open(files["depth"]) as t:
        members = t.getmembers()
        assert [m.name for m in members] == [f"clip0007.{i:06d}.voxel_depth_100.front.png" for i in range(5)]
        assert all(m.mode == 0o444 and m.uname == "bigdata" and m.gname == "bigdata" for m in members)
    d = wf.read_tar_sample(files["depth"])
    assert d["__key__"] == "clip0007"
    for i in range(5):
        got = d[f"{i:06d}.voxel_depth_100.front.png"]
        assert got.dtype == np.uint16 and np.array_equal(got, du16[i])
    assert int(d["000000.voxel_depth_100.front.png"][10, 10]) == 65535
    s = wf.read_tar_sample(files["instance"])
    assert np.array_equal(s["000001.instance_buffer.front.png"], inst[1].astype(np.uint16))
    p = wf.read_tar_sample(files["pose"])
    keys = sorted(k for k in p if "pose.front.npy" in k)      # how stage 3 enumerates them
    assert len(keys) == 5 and all(np.array_equal(p[k], poses[i]) and p[k].dtype == np.float32 for i, k in enumerate(keys))
    assert np.array_equal(wf.read_tar_sample(files["intrinsic"])["intrinsic.front.npy"], intr)
    # .npy payloads are exactly numpy's own serialisation (what webdataset's npy handler emits)
    with tarfile.open(files["pose"]) as t:
        raw = t.extractfile(t.getmembers()[2]).read()
    ref = io.BytesIO()
    np.lib.format.write_array(ref, poses[2])
    assert raw == ref.getvalue()
    for k in ("semantic_video", "coordinate_video", "depth_vis_video"):
        assert os.path.getsize(files[k]) > 1000


def test_write_to_tar_semantics(tmp_path):
    sample = {"b.json": {"x": [1, 2]}, "a.txt": "hello", "c.pyd": {"k": np.arange(3)}, "_private": b"skip"}
    wf.write_to_tar(sample, tmp_path / "sub" / "x.tar", __key__="k1")
    assert sample["__key__"] == "k1"                      # the reference mutates the caller's dict the same way
    with tarfile.open(tmp_path / "sub" / "x.tar") as t:
        assert [m.name for m in t.getmembers()] == ["k1.a.txt", "k1.b.json", "k1.c.pyd"]
    r = wf.read_tar_sample(tmp_path / "sub" / "x.tar")
    assert r["b.json"] == {"x": [1, 2]} and r["a.txt"] == b"hello" and np.array_equal(r["c.pyd"]["k"], np.arange(3))
    with pytest.raises(ValueError, match="__key__"):
        wf.write_to_tar({"a.txt": "x"}, tmp_path / "y.tar")
    with pytest.raises(ValueError, match="no encoder"):
        wf.write_to_tar({"a.bin": 3.5}, tmp_path / "z.tar", __key__="k")


def test_png16_roundtrip_and_video_inputs(tmp_path):
    a = np.arange(48 * 64, dtype=np.uint16).reshape(48, 64) * 17
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(wf.imageencoder_imageio_png(a)))), a)
    with pytest.raises(TypeError):
        wf.imageencoder_imageio_png(a.astype(np.float32))
    frames = {f"{i:04d}.png": wf.imageencoder_imageio_png(np.full((32, 48, 3), i * 40, np.uint8)) for i in range(3)}
    frames["__key__"] = "clip"
    wf.write_video_file(frames, tmp_path / "v", fps=10)       # dict of PNG bytes, suffix added
    assert os.path.getsize(tmp_path / "v.mp4") > 500
    assert wf.X264_PARAMS[:4] == ["-preset", "veryslow", "-crf", "23.5"] and wf.X264_PARAMS[-2:] == ["-movflags", "+faststart"]


@pytest.mark.gpu
def test_depth_quantisation_kernel_bit_exact():
    """icv_depth_to_u16 == numpy's (depth * 100).astype(uint16) at the full 93 x 480 x 832 size, including values that
    truncate, the 0 sky value and values beyond the uint16 range (numpy wraps modulo 2^16 on x86-64)."""
    g = torch.Generator().manual_seed(1)
    depth = torch.rand((93, 480, 832), generator=g) * 300
    depth[:, :50] = 0
    depth[3, 100, 100], depth[3, 100, 101], depth[3, 100, 102] = 655.35, 655.36, 1000.0
    want = (depth.numpy() * 100).astype(np.uint16)
    got = wf.depth_to_uint16_x100(depth)
    assert got.dtype == np.uint16 and np.array_equal(got, want)
    got2 = wf.depth_to_uint16_x100(depth.reshape(-1)[: 93 * 480 * 832 - 3].to("cuda:0"))     # ragged tail, device input
    assert np.array_equal(got2, want.reshape(-1)[:-3])
