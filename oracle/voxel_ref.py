"""ORACLE (test infrastructure) — CPU restatement of the voxel part of the guidance-buffer renderer
(SURVEY.md §8f row 4).  Only tests/ may import this.

** PARITY UNPINNED **: the reference renders through fVDB 0.2.0 (`fvdb.gridbatch_from_points`, `GridBatch.
segments_along_rays`, `GridBatch.voxels_along_rays`) [R infinicube/utils/fvdb_utils.py:139-215,572-605;
infinicube/camera/base.py:520-619], an un-vendored CUDA wheel (pyproject.toml:69) absent from /root/reference and this
container, and no golden buffers exist.  Restated from the call sites and fVDB's published semantics:
  * `points_to_voxels`: voxel (i,j,k) = round((p - origin) / voxel_size), origin = voxel_size / 2 [R fvdb_utils.py:591,
    153-156]; per-voxel attribute = the most frequent category among the voxel's points, ties -> the smallest category
    ("argmax-category": argmax over the sorted unique categories) [R fvdb_utils.py:105-109,170-193];
  * depth = start of the first run of consecutive occupied voxels along the ray whose length exceeds eps (1e-1)
    [`segments_along_rays(o, d, 1, eps=1e-1)`, R camera/base.py:541-549], times the camera ray's z [R base.py:350-361];
  * semantic / instance = attribute of the first occupied voxel whose in-voxel ray length exceeds eps (1e-2), else the
    background value [`voxels_along_rays(o, d, 1, eps=1e-2)`, R camera/base.py:600-614].
Two implementations: `raycast_dda` mirrors the kernel's float32 arithmetic operation for operation (plain cell-by-cell
walk, no brick skipping: it also checks the kernel's skip logic) and must match bit for bit; `raycast_bruteforce`
intersects every occupied voxel with every ray in float64 (no traversal at all) and validates the DDA semantics up to
rays that graze a voxel within rounding."""
from __future__ import annotations

import numpy as np

F = np.float32


def points_to_voxels(points: np.ndarray, attrs: dict, voxel_size, origin=None):
    """points [P,3] float32, attrs {name: int array [P]} -> (ijk int32 [M,3] sorted by (k,j,i), {name: int array [M]})."""
    vs = np.asarray(voxel_size, F)
    org = vs / F(2) if origin is None else np.asarray(origin, F)
    ijk = np.rint((points.astype(F) - org) / vs).astype(np.int64)          # torch.round == round-half-even
    order = np.lexsort((ijk[:, 0], ijk[:, 1], ijk[:, 2]))
    s = ijk[order]
    new = np.ones(len(s), bool)
    new[1:] = (s[1:] != s[:-1]).any(1)
    vid = np.cumsum(new) - 1
    uniq = s[new]
    out = {}
    for name, a in attrs.items():
        a = np.asarray(a)[order]
        res = np.zeros(len(uniq), a.dtype)
        for v in range(len(uniq)):
            vals, cnt = np.unique(a[vid == v], return_counts=True)      # sorted categories: argmax = first max = smallest
            res[v] = vals[np.argmax(cnt)]
        out[name] = res
    return uniq.astype(np.int32), out


def dense_volume(ijk: np.ndarray, pad: int = 8):
    lo = ijk.min(0) - pad
    lo = (np.floor(lo / 8.0) * 8).astype(np.int64)
    dims = ((ijk.max(0) + pad + 1 - lo + 7) // 8 * 8).astype(np.int64)
    vol = np.full((dims[2], dims[1], dims[0]), -1, np.int32)
    r = ijk - lo
    vol[r[:, 2], r[:, 1], r[:, 0]] = np.arange(len(ijk), dtype=np.int32)
    return vol, lo.astype(np.int32), dims.astype(np.int32)


def raycast_dda(vol, vol_min, voxel_size, rays_cam, poses, eps_depth=1e-1, eps_voxel=1e-2):
    """-> (zdepth f32 [N,HW], hit voxel index i32 [N,HW]); float32 arithmetic in the kernel's order, vectorised over rays."""
    vs = np.asarray(voxel_size, F)
    Dz, Dy, Dx = vol.shape
    D = np.array([Dx, Dy, Dz])
    N, HW = poses.shape[0], rays_cam.shape[0]
    r = rays_cam.astype(F)
    m = poses.astype(F)
    d = np.empty((N, HW, 3), F)
    for i in range(3):
        d[..., i] = ((m[:, i, 0][:, None] * r[None, :, 0]) + (m[:, i, 1][:, None] * r[None, :, 1])) + (m[:, i, 2][:, None] * r[None, :, 2])
    grid_lo = (vol_min.astype(np.float64) * vs.astype(np.float64)).astype(F)       # low corner of cell (0,0,0): origin - vs/2 = ijk_min * vs
    iv = (F(1.0) / vs).astype(F)
    o = np.empty((N, HW, 3), F)
    for i in range(3):
        o[..., i] = np.broadcast_to(((m[:, i, 3] - grid_lo[i]) * iv[i])[:, None], (N, HW))
    dg = (d * iv[None, None, :]).astype(F)
    step = np.sign(dg).astype(np.int64)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.where(step != 0, (F(1.0) / dg).astype(F), F(0.0)).astype(F)
    o, dg, step, inv = (x.reshape(-1, 3) for x in (o, dg, step, inv))
    rz = np.broadcast_to(r[None, :, 2], (N, HW)).reshape(-1)
    n = o.shape[0]
    t0 = np.zeros(n, F)
    t1 = np.full(n, np.inf, F)
    miss = np.zeros(n, bool)
    for i in range(3):
        z = step[:, i] == 0
        miss |= z & ((o[:, i] < 0) | (o[:, i] >= F(D[i])))
        with np.errstate(invalid="ignore"):
            ta = ((F(0.0) - o[:, i]) * inv[:, i]).astype(F)
            tb = ((F(D[i]) - o[:, i]) * inv[:, i]).astype(F)
        t0 = np.where(z, t0, np.maximum(t0, np.minimum(ta, tb))).astype(F)
        t1 = np.where(z, t1, np.minimum(t1, np.maximum(ta, tb))).astype(F)
    active = ~miss & (t0 < t1)
    c = np.empty((n, 3), np.int64)
    for i in range(3):
        c[:, i] = np.clip(np.floor((o[:, i] + (t0 * dg[:, i]).astype(F)).astype(F)), 0, D[i] - 1).astype(np.int64)
    t_cur = t0.copy()
    run_start = np.zeros(n, F)
    in_run = np.zeros(n, bool)
    depth_done = np.zeros(n, bool)
    hit_done = np.zeros(n, bool)
    depth = np.zeros(n, F)
    hit = np.full(n, -1, np.int32)
    ed, ev = F(eps_depth), F(eps_voxel)

    def face(i, cc):
        with np.errstate(invalid="ignore", over="ignore"):
            t = ((np.where(step[:, i] > 0, cc + 1, cc).astype(F) - o[:, i]) * inv[:, i]).astype(F)
        return np.where(step[:, i] == 0, F(np.inf), t).astype(F)

    for _ in range(int(D.sum()) + 8):
        inside = active & (c >= 0).all(1) & (c < D[None, :]).all(1)
        leaving = active & ~inside
        fin = leaving & in_run & ~depth_done & ((t_cur - run_start) > ed)
        depth[fin] = run_start[fin]
        active = inside
        if not active.any():
            break
        cc = np.clip(c, 0, D[None, :] - 1)
        idx = np.where(active, vol[cc[:, 2], cc[:, 1], cc[:, 0]], -1)
        tx, ty, tz = face(0, c[:, 0]), face(1, c[:, 1]), face(2, c[:, 2])
        a = np.zeros(n, np.int64)
        t_out = tx.copy()
        m1 = ty < t_out
        a[m1] = 1
        t_out = np.where(m1, ty, t_out)
        m2 = tz < t_out
        a[m2] = 2
        t_out = np.where(m2, tz, t_out).astype(F)
        occ = active & (idx >= 0)
        emp = active & (idx < 0)
        h = occ & ~hit_done & ((t_out - t_cur) > ev)
        hit[h] = idx[h]
        hit_done |= h
        st = occ & ~in_run
        run_start[st] = t_cur[st]
        in_run |= st
        close = emp & in_run
        dd = close & ~depth_done & ((t_cur - run_start) > ed)
        depth[dd] = run_start[dd]
        depth_done |= dd
        in_run &= ~close
        active &= ~(depth_done & hit_done)
        mv = active
        c[mv, a[mv]] += step[mv, a[mv]]
        t_cur = np.where(mv, t_out, t_cur).astype(F)
    return (depth * rz).astype(F).reshape(N, HW), hit.reshape(N, HW)


def raycast_bruteforce(ijk, voxel_size, rays_cam, poses, eps_depth=1e-1, eps_voxel=1e-2, tol=1e-6):
    """float64, no traversal: every ray against every occupied voxel's box.  Returns (zdepth, hit index, ambiguous mask):
    `ambiguous` marks rays whose decision depends on a length within `tol` of an eps or on a near-tie."""
    vs = np.asarray(voxel_size, np.float64)
    lo = ijk.astype(np.float64) * vs                      # voxel i spans [i*vs, (i+1)*vs) (origin = vs/2)
    hi = lo + vs
    N, HW = poses.shape[0], rays_cam.shape[0]
    depth = np.zeros((N, HW))
    hit = np.full((N, HW), -1, np.int64)
    amb = np.zeros((N, HW), bool)
    r = rays_cam.astype(np.float64)
    for n in range(N):
        R, t = poses[n, :3, :3].astype(np.float64), poses[n, :3, 3].astype(np.float64)
        d = r @ R.T
        for px in range(HW):
            dd = d[px]
            with np.errstate(divide="ignore", invalid="ignore"):
                ta = (lo - t) / dd
                tb = (hi - t) / dd
            tmin = np.where(dd != 0, np.minimum(ta, tb), np.where((t >= lo) & (t < hi), -np.inf, np.inf)).max(1)
            tmax = np.where(dd != 0, np.maximum(ta, tb), np.where((t >= lo) & (t < hi), np.inf, -np.inf)).min(1)
            tmin = np.maximum(tmin, 0.0)
            ok = tmax > tmin
            if not ok.any():
                continue
            idx = np.nonzero(ok)[0]
            order = idx[np.argsort(tmin[idx], kind="stable")]
            tin, tout = tmin[order], tmax[order]
            ln = tout - tin
            if (np.abs(ln - eps_voxel) < tol).any() or (len(tin) > 1 and (np.diff(tin) < tol).any()):
                amb[n, px] = True
            first = np.nonzero(ln > eps_voxel)[0]
            if len(first):
                hit[n, px] = order[first[0]]
            # runs of voxels whose intervals touch
            s, e = tin[0], tout[0]
            found = False
            for k in range(1, len(tin) + 1):
                if k < len(tin) and tin[k] <= e + tol:
                    e = max(e, tout[k])
                    continue
                if abs((e - s) - eps_depth) < tol:
                    amb[n, px] = True
                if e - s > eps_depth:
                    depth[n, px] = s * r[px, 2]
                    found = True
                    break
                if k < len(tin):
                    s, e = tin[k], tout[k]
            if not found:
                depth[n, px] = 0.0
    return depth, hit, amb
