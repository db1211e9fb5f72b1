"""SURVEY.md §8f row 4 (second half): voxel ray-cast of the guidance-buffer renderer.  fVDB is absent, so the oracle
(oracle/voxel_ref.py) restates its call-site semantics — parity with the reference is UNPINNED.  CPU: the two oracle
implementations (float32 cell walk vs float64 all-voxel brute force) agree on every ray that is not within rounding of a
decision boundary; voxelisation semantics.  GPU: the HIP kernel (brick skipping, through the C ABI) equals the
cell-walk oracle BIT FOR BIT (depth floats and voxel indices), and at the full 93 x 480 x 832 size holds the
size-independent properties (sky, monotonicity under camera advance, brick skipping on/off agreement via a sparse vs a
dense-bricked scene)."""
import numpy as np
import pytest
import torch

from oracle import voxel_ref as V


class Cam:
    def __init__(self, w, h, f):
        self.w, self.h = w, h
        u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
        r = np.stack([(u - w / 2) / f, (v - h / 2) / f, np.ones_like(u)], -1)
        self.rays = (r / np.linalg.norm(r, axis=-1, keepdims=True)).astype(np.float32)

    def get_rays(self):
        return torch.from_numpy(self.rays)


def _scene(seed=0, n_pts=6000):
    """Street canyon point cloud in a z-up world: ground, two walls, a few boxes; camera looks along +x."""
    g = np.random.default_rng(seed)
    ground = np.stack([g.uniform(0, 30, n_pts), g.uniform(-6, 6, n_pts), g.normal(0, 0.02, n_pts)], 1)
    wall_l = np.stack([g.uniform(0, 30, n_pts // 2), np.full(n_pts // 2, 6.0) + g.normal(0, 0.03, n_pts // 2), g.uniform(0, 5, n_pts // 2)], 1)
    wall_r = wall_l * np.array([1, -1, 1])
    box = np.stack([g.uniform(12, 14, 800), g.uniform(-1, 1, 800), g.uniform(0, 1.5, 800)], 1)
    thin = np.stack([np.full(60, 8.03), g.uniform(-0.5, 0.5, 60), g.uniform(0.5, 1.0, 60)], 1)     # a one-voxel-thick sheet (eps cases)
    pts = np.concatenate([ground, wall_l, wall_r, box, thin]).astype(np.float32)
    sem = np.concatenate([np.full(len(ground), 18), np.full(len(wall_l) * 2, 14), np.full(len(box), 1), np.full(len(thin), 10)]).astype(np.int32)
    sem[g.integers(0, len(sem), 200)] = 15                      # label noise: exercises the per-voxel mode
    inst = np.where(sem == 1, 7, 0).astype(np.int32)
    return pts, sem, inst


def _poses(n):
    # camera (x right, y down, z front) -> world (x front, y left, z up), moving forward, slight yaw
    base = np.array([[0, 0, 1, 0], [-1, 0, 0, 0], [0, -1, 0, 1.6], [0, 0, 0, 1]], np.float32)
    out = []
    for i in range(n):
        yaw = 0.05 * i
        rz = np.array([[np.cos(yaw), -np.sin(yaw), 0, 1.0 + 0.7 * i], [np.sin(yaw), np.cos(yaw), 0, 0.1 * i], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
        out.append(rz @ base)
    return np.stack(out).astype(np.float32)


def test_voxelisation_mode_and_rounding():
    pts = np.array([[0.09, 0.1, 0.1], [0.11, 0.1, 0.1], [0.19, 0.1, 0.1], [0.3, 0.1, 0.1], [0.3, 0.1, 0.1], [0.3, 0.11, 0.1], [-0.05, 0.1, 0.1]], np.float32)
    sem = np.array([5, 3, 3, 9, 2, 2, 4], np.int32)
    ijk, a = V.points_to_voxels(pts, {"semantics": sem}, (0.2, 0.2, 0.2))
    assert ijk.tolist() == [[-1, 0, 0], [0, 0, 0], [1, 0, 0]]               # voxel i spans [0.2 i, 0.2 (i+1)); ties round to even
    assert a["semantics"].tolist() == [4, 3, 2]                               # mode; (9,2,2) -> 2
    ijk2, a2 = V.points_to_voxels(np.array([[0.1, 0.1, 0.1]] * 2, np.float32), {"s": np.array([7, 3])}, (0.2, 0.2, 0.2))
    assert a2["s"].tolist() == [3]                                           # tie -> the smallest category
    from infinicube_amd.utils.voxel_render import points_to_voxels
    p, s, i = _scene()
    ti, ta = points_to_voxels(torch.from_numpy(p), {"semantics": torch.from_numpy(s), "instance": torch.from_numpy(i)})
    oi, oa = V.points_to_voxels(p, {"semantics": s, "instance": i}, (0.2, 0.2, 0.2))
    assert np.array_equal(ti.numpy(), oi) and np.array_equal(ta["semantics"].numpy(), oa["semantics"]) and np.array_equal(ta["instance"].numpy(), oa["instance"])


def test_oracle_walk_agrees_with_bruteforce():
    p, s, i = _scene(1, 1500)
    ijk, attrs = V.points_to_voxels(p, {"semantics": s}, (0.2, 0.2, 0.2))
    vol, vmin, dims = V.dense_volume(ijk)
    cam = Cam(24, 16, 20.0)
    poses = _poses(2)
    d, h = V.raycast_dda(vol, vmin, (0.2, 0.2, 0.2), cam.rays.reshape(-1, 3), poses)
    db, hb, amb = V.raycast_bruteforce(ijk, (0.2, 0.2, 0.2), cam.rays.reshape(-1, 3), poses)
    ok = ~amb
    assert ok.mean() > 0.9
    assert np.array_equal(h[ok], hb[ok]), f"{(h[ok] != hb[ok]).sum()} rays: first voxel differs between the cell walk and brute force"
    assert np.abs(d[ok] - db[ok]).max() < 1e-3
    assert (h >= 0).mean() > 0.2 and (d > 0).mean() > 0.2 and (d == 0).any()      # the scene has both hits and sky


@pytest.mark.gpu
def test_hip_raycast_edge_cases():
    """An empty world, a camera inside an occupied voxel, rays parallel to an axis, a world far off the view."""
    from infinicube_amd.utils.voxel_render import render_voxel_buffers
    cam = Cam(16, 12, 10.0)
    pose = torch.from_numpy(_poses(1)[0])
    d, s, i = render_voxel_buffers(cam, pose, torch.zeros((0, 3)), torch.zeros((0,), dtype=torch.int32))
    assert d.shape == (12, 16) and float(d.abs().max()) == 0 and int(s.abs().max()) == 0
    # a solid 2 m cube around the camera: every ray starts inside occupied space -> the run starts at t = 0 (depth 0 * z = 0),
    # the semantic map still reports the voxel the ray starts in
    g = np.random.default_rng(0)
    cube = torch.from_numpy((g.uniform(-1, 1, (200000, 3)) + np.array([1.0, 0.0, 1.6])).astype(np.float32))
    d, s, i = render_voxel_buffers(cam, pose, cube, torch.full((len(cube),), 14, dtype=torch.int32))
    assert float(d.abs().max()) == 0 and bool((s == 14).all())
    # axis-parallel rays (identity rotation, rays along +z of a z-front camera placed in world axes) hit a slab exactly
    eye = torch.eye(4)
    one = Cam(1, 1, 1.0)
    one.rays = np.array([[[0.0, 0.0, 1.0]]], np.float32)
    slab = torch.from_numpy(np.stack(np.meshgrid(np.arange(-1, 1, 0.05), np.arange(-1, 1, 0.05), [5.05]), -1).reshape(-1, 3).astype(np.float32))
    d, s, i = render_voxel_buffers(one, eye, slab, torch.full((len(slab),), 18, dtype=torch.int32))
    assert abs(float(d[0, 0]) - 5.0) < 1e-5 and int(s[0, 0]) == 18           # the voxel [5.0, 5.2) starts at z = 5.0
    far = slab + torch.tensor([500.0, 0.0, 0.0])
    d, s, i = render_voxel_buffers(one, eye, far, torch.full((len(far),), 18, dtype=torch.int32))
    assert float(d[0, 0]) == 0 and int(s[0, 0]) == 0


@pytest.mark.gpu
def test_hip_raycast_bit_exact_vs_oracle():
    from infinicube_amd.utils.voxel_render import VoxelVolume, points_to_voxels
    p, s, i = _scene(2)
    ijk, attrs = points_to_voxels(torch.from_numpy(p).cuda(), {"semantics": torch.from_numpy(s).cuda(), "instance": torch.from_numpy(i).cuda()})
    volume = VoxelVolume.build(ijk)
    cam = Cam(64, 48, 50.0)
    poses = _poses(3)
    depth, sem, inst, idx = volume.raycast(cam.get_rays(), torch.from_numpy(poses), attrs["semantics"], attrs["instance"], want_index=True)
    ijk_np = ijk.cpu().numpy()
    vol, vmin, dims = V.dense_volume(ijk_np)
    assert np.array_equal(vmin, volume.vol_min) and np.array_equal(dims, volume.dims) and np.array_equal(vol, volume.vol.cpu().numpy())
    d, h = V.raycast_dda(vol, vmin, (0.2, 0.2, 0.2), cam.rays.reshape(-1, 3), poses)
    got_h = idx.cpu().numpy().reshape(3, -1)
    assert np.array_equal(got_h, h), f"{(got_h != h).sum()} rays hit a different voxel than the oracle's cell walk"
    assert np.array_equal(depth.cpu().numpy().reshape(3, -1), d), "z-depth differs from the oracle bit for bit"
    sa = attrs["semantics"].cpu().numpy()
    assert np.array_equal(sem.cpu().numpy().reshape(3, -1), np.where(h >= 0, sa[np.maximum(h, 0)], 0))
    assert set(np.unique(inst.cpu().numpy())) <= {0, 7}
    # the thin sheet (one voxel thick, 0.2 m): visible in the semantic map (eps 0.01) AND as depth (0.2 > 0.1)
    assert (sem.cpu().numpy() == 10).any()


@pytest.mark.gpu
def test_hip_raycast_full_size_properties_and_rate():
    """93 x 480 x 832 rays over a 60 m street: one launch; reports rays/s (profiles/r02/voxel_raycast.md)."""
    import time
    from infinicube_amd.utils.voxel_render import render_voxel_buffers
    p, s, i = _scene(3, 400000)
    cam = Cam(832, 480, 700.0)
    poses = torch.from_numpy(_poses(93) * np.array([[1, 1, 1, 0.3], [1, 1, 1, 0.3], [1, 1, 1, 1], [1, 1, 1, 1]], np.float32))
    args = (cam, poses, torch.from_numpy(p), torch.from_numpy(s), torch.from_numpy(i))
    render_voxel_buffers(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    depth, sem, inst = render_voxel_buffers(*args)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert depth.shape == (93, 480, 832) and sem.dtype == torch.int32
    assert bool(((depth == 0) == (sem == 0)).float().mean() > 0.99)         # sky <-> no depth (up to the eps difference)
    assert float(depth.max()) < 80 and float(depth[depth > 0].min()) > 0.2
    assert 0.02 < float((sem == 0).float().mean()) < 0.9                      # the canyon has open sky and solid walls / ground
    assert {14, 18} <= set(torch.unique(sem).tolist())                       # BUILDING and ROAD both visible
    # 4000 sampled rays of frame 40 against the oracle's cell walk on the same voxel world: bit for bit
    from infinicube_amd.utils.voxel_render import points_to_voxels
    ijk, vattrs = points_to_voxels(torch.from_numpy(p), {"semantics": torch.from_numpy(s)})
    vol, vmin, dims = V.dense_volume(ijk.numpy())
    g = np.random.default_rng(0)
    pix = np.sort(g.choice(480 * 832, 4000, replace=False))
    d_o, h_o = V.raycast_dda(vol, vmin, (0.2, 0.2, 0.2), cam.rays.reshape(-1, 3)[pix], poses[40:41].numpy())
    assert np.array_equal(depth[40].reshape(-1)[torch.from_numpy(pix).cuda()].cpu().numpy(), d_o[0])
    sa = vattrs["semantics"].numpy()
    assert np.array_equal(sem[40].reshape(-1)[torch.from_numpy(pix).cuda()].cpu().numpy(), np.where(h_o[0] >= 0, sa[np.maximum(h_o[0], 0)], 0))
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    open("gpurun_out/voxel_raycast.txt", "w").write(f"{93 * 480 * 832 / dt / 1e6:.1f} M rays/s, {dt * 1e3:.1f} ms for 93 x 480 x 832 rays incl. voxelisation of {len(p)} points\n")
    print(f"\\nvoxel ray-cast: {93 * 480 * 832 / dt / 1e6:.1f} M rays/s incl. voxelisation of {len(p)} points ({dt * 1e3:.1f} ms for 93 x 480 x 832)")
