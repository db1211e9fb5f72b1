"""ICV_WORLD=N: the unchanged single-process caller [R infinicube/inference/guidance_buffer_generation.py:755-782]
gets N ranks from inside the constructor (multigpu.WorkerPool): N fresh worker processes, rank 0 included; the caller's
process is their client (no process group, no environment writes).  Real processes over gloo on CPU; the whole path goes
through WanVideoGenerator.generate -> WorkerPool.generate -> WanVideoPipeline.__call__ on every rank."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))


def _checkpoint(tmp_path):
    import mgpu_factory as F
    from infinicube_amd.videogen import synthetic as syn
    bsd = syn.make_buffer_embedder_state_dict(F.CFG)
    path = str(tmp_path / "step-1.safetensors")
    save_file({"buffer_embedder." + k: v for k, v in bsd.items()}, path)
    return path


def _run(path, tmp_path, out_name):
    import mgpu_factory as F
    from infinicube.videogen import WanVideoGenerator
    from infinicube_amd.videogen import synthetic as syn
    sem, co = syn.make_dummy_buffers(F.GRID)
    co[:, :, : F.GRID.width // 2] //= 2
    out = str(tmp_path / out_name)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        g = WanVideoGenerator(path, device="cpu", use_wan_1pt3b=True, pipeline_factory=F.factory)
        frames = g.generate(sem, co, seed=3, output_path=out)
        frames2 = g.generate(sem, co, seed=4)
    return g, np.stack([np.asarray(f) for f in frames]), np.stack([np.asarray(f) for f in frames2]), out, buf.getvalue()


@pytest.mark.parametrize("world,parallelism,kv_exchange", [(2, "auto", "allgather"), (3, "sp", "p2p")])
def test_generator_spawns_workers_and_matches_single_process(tmp_path, monkeypatch, world, parallelism, kv_exchange):
    path = _checkpoint(tmp_path)
    _, ref1, ref2, out_single, log_single = _run(path, tmp_path, "single.mp4")
    assert os.path.getsize(out_single) > 0
    monkeypatch.setenv("ICV_WORLD", str(world))
    monkeypatch.setenv("ICV_DIST_BACKEND", "gloo")
    monkeypatch.setenv("ICV_WORKER_FACTORY", "mgpu_factory:factory")
    monkeypatch.setenv("ICV_PARALLELISM", parallelism)
    monkeypatch.setenv("ICV_KV_EXCHANGE", kv_exchange)
    monkeypatch.setenv("ICV_WORLD_TIMEOUT_S", "300")
    monkeypatch.setenv("PYTHONPATH", os.pathsep.join([os.path.dirname(HERE), HERE, os.environ.get("PYTHONPATH", "")]))
    g = None
    env_before = dict(os.environ)
    try:
        g, got1, got2, out_multi, log_multi = _run(path, tmp_path, "multi.mp4")
        import torch.distributed as dist
        # the caller's process is a CLIENT: it joined no process group, its environment was not written, and every rank - 0
        # included - is a fresh process whose runtime initialised under the composed environment
        assert not dist.is_initialized() and g._pool is not None and g._pool.world == world
        assert dict(os.environ) == env_before, "the pool must not write the caller's environment"
        rec = g._pool.plan_record()
        assert rec["client"]["pid"] == os.getpid() and rec["client"]["joined_process_group"] is False
        assert [r["rank"] for r in rec["ranks"]] == list(range(world)) and len({r["pid"] for r in rec["ranks"]} | {os.getpid()}) == world + 1
        for r in rec["ranks"]:
            assert r["GPU_MAX_HW_QUEUES"] == "16" and r["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and r["hip_initialised_at_start"] is False
            assert r["NCCL_MAX_NCHANNELS"] == "8"          # the RCCL channel cap lives in the ranks' environments only
        assert rec["plan"] == [g.pipe.parallelism, g.pipe.kv_exchange]
        assert not hasattr(g.pipe, "dit") or g.pipe.dit is None, "the client loads no weights"
        assert log_multi == log_single.replace("single.mp4", "multi.mp4"), "the caller-visible progress lines must not change"
        assert os.path.getsize(out_multi) > 0
        # sharded attention reorders fp32 sums (and bf16 storage roundings can flip): frames agree to rounding
        for a, b in ((ref1, got1), (ref2, got2)):
            assert a.shape == b.shape
            d = np.abs(a.astype(np.int16) - b.astype(np.int16))
            assert d.max() <= 2 and (d > 0).mean() < 0.02, f"N-rank frames differ from the single-process frames: max {d.max()}, {100 * (d > 0).mean():.2f} % pixels"
        assert not np.array_equal(got1, got2), "the seed of the second request must reach every rank"
        # a SECOND generator built with the same arguments reuses the live pool (its workers sit in their serve loop: the
        # constructor must not issue a barrier they would never answer) and serves requests like the first
        import mgpu_factory as F
        from infinicube.videogen import WanVideoGenerator
        from infinicube_amd.videogen import synthetic as syn
        with contextlib.redirect_stdout(io.StringIO()):
            g2 = WanVideoGenerator(path, device="cpu", use_wan_1pt3b=True, pipeline_factory=F.factory)
            assert g2._pool is g._pool
            sem, co = syn.make_dummy_buffers(F.GRID)
            co[:, :, : F.GRID.width // 2] //= 2
            again = np.stack([np.asarray(f) for f in g2.generate(sem, co, seed=3)])
            unseeded = [np.stack([np.asarray(f) for f in g2.generate(sem, co, seed=None)]) for _ in range(2)]
        assert np.array_equal(again, got1), "the reused pool must reproduce the first generator's frames for the same request"
        # seed=None: ONE drawn seed for all ranks (shards of one latent) - finite frames, and a different video per call
        assert all(u.shape == got1.shape for u in unseeded) and not np.array_equal(unseeded[0], unseeded[1])
    finally:
        if g is not None and g._pool is not None:
            g._pool.close()
    import torch.distributed as dist
    assert not dist.is_initialized()


def test_requested_world_parsing(monkeypatch):
    from infinicube_amd.videogen import multigpu
    monkeypatch.delenv("ICV_WORLD", raising=False)
    assert multigpu.requested_world() == 1
    monkeypatch.setenv("ICV_WORLD", "4")
    assert multigpu.requested_world() == 4
    monkeypatch.setenv("ICV_WORKER_RANK", "2")       # inside a worker: never recurse
    assert multigpu.requested_world() == 1
    monkeypatch.delenv("ICV_WORKER_RANK")
    monkeypatch.setenv("ICV_WORLD", "0")
    with pytest.raises(ValueError):
        multigpu.requested_world()


def test_device_literal_maps_to_local_rank(monkeypatch):
    from infinicube_amd.videogen.pipeline import WanVideoPipeline as P
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert P.resolve_device("cuda:0") == "cuda:0" and P.resolve_device("cuda") == "cuda:0" and P.resolve_device("cpu") == "cpu"
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert P.resolve_device("cuda:0") == "cuda:0"        # LOCAL_RANK alone (world 1) changes nothing
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert P.resolve_device("cuda:0") == "cuda:5" and P.resolve_device("cuda") == "cuda:5" and P.resolve_device("cpu") == "cpu"


def test_worker_that_dies_while_loading_is_reported(tmp_path, monkeypatch):
    """A worker that fails after joining the process group (missing checkpoint, out of memory, ...) must surface as an error
    with ITS traceback in the caller within seconds, not as a collective timeout."""
    import time
    import mgpu_factory as F
    from infinicube.videogen import WanVideoGenerator
    from infinicube_amd.videogen import multigpu
    path = _checkpoint(tmp_path)
    monkeypatch.setenv("ICV_WORLD", "2")
    monkeypatch.setenv("ICV_DIST_BACKEND", "gloo")
    monkeypatch.setenv("ICV_WORKER_FACTORY", "mgpu_factory:failing_factory")
    monkeypatch.setenv("ICV_WORLD_TIMEOUT_S", "300")
    monkeypatch.setenv("PYTHONPATH", os.pathsep.join([os.path.dirname(HERE), HERE, os.environ.get("PYTHONPATH", "")]))
    t0 = time.time()
    try:
        with pytest.raises(RuntimeError, match="synthetic worker failure"):
            with contextlib.redirect_stdout(io.StringIO()):
                WanVideoGenerator(path, device="cpu", use_wan_1pt3b=True, pipeline_factory=F.factory)
        assert time.time() - t0 < 120
    finally:
        if multigpu._ACTIVE_POOL is not None:
            multigpu._ACTIVE_POOL.close()
    import torch.distributed as dist
    assert not dist.is_initialized()


@pytest.mark.parametrize("inject,want_plan,n_failed", [
    ("0:1:raise", ("sp", "allgather"), 1),               # world 3 + p2p requested: plan 0 = (sp, p2p) fails on a worker -> (sp, allgather)
    ("0:0:hang,1:2:raise", None, 2),                     # rank 0 hangs in plan 0, rank 2 raises in plan 1 -> ONE GPU, in the caller's process
])
def test_pool_start_falls_back_plan_by_plan(tmp_path, monkeypatch, capfd, inject, want_plan, n_failed):
    """The staged start of the worker pool (the mirror of bench.py's launch guard): every plan is probed on all ranks before any
    weights load; a rank that raises or hangs in the probe abandons that process group (workers killed, group destroyed) and the
    next plan starts from fresh processes; when nothing is left the caller's generator simply runs on one GPU.  Frames must equal
    the single-process run either way, and stderr names what failed."""
    import torch.distributed as dist
    path = _checkpoint(tmp_path)
    _, ref1, _, _, _ = _run(path, tmp_path, "single.mp4")
    monkeypatch.setenv("ICV_WORLD", "3")
    monkeypatch.setenv("ICV_DIST_BACKEND", "gloo")
    monkeypatch.setenv("ICV_WORKER_FACTORY", "mgpu_factory:factory")
    monkeypatch.setenv("ICV_PARALLELISM", "auto")
    monkeypatch.setenv("ICV_KV_EXCHANGE", "p2p")
    monkeypatch.setenv("ICV_WORLD_TIMEOUT_S", "300")
    monkeypatch.setenv("ICV_WORLD_PROBE_TIMEOUT_S", "8")
    monkeypatch.setenv("ICV_TEST_POOL_INJECT", inject)
    monkeypatch.setenv("ICV_TEST_HOOKS", "1")
    monkeypatch.setenv("PYTHONPATH", os.pathsep.join([os.path.dirname(HERE), HERE, os.environ.get("PYTHONPATH", "")]))
    g = None
    try:
        g, got1, _, _, _ = _run(path, tmp_path, "multi.mp4")
        err = capfd.readouterr().err
        assert err.count("multi-GPU start with plan") == n_failed
        if want_plan is None:
            assert g._pool is None and "continuing on ONE GPU" in err
            assert np.array_equal(got1, ref1)
        else:
            assert g._pool is not None and g._pool.plan == want_plan and len(g._pool.failed_plans) == n_failed
            assert not dist.is_initialized() and g._pool.world == 3 and len(g._pool.plan_record()["ranks"]) == 3
            assert (g.pipe.parallelism, g.pipe.kv_exchange) == want_plan
            d = np.abs(ref1.astype(np.int16) - got1.astype(np.int16))
            assert d.max() <= 2
    finally:
        if g is not None and g._pool is not None:
            g._pool.close()
        from infinicube_amd.videogen import multigpu
        multigpu._DEGRADED = None
    assert not dist.is_initialized()


def test_a_rank_that_fails_in_the_middle_of_a_request_ends_the_pool_with_its_traceback(tmp_path, monkeypatch):
    """One rank raises inside a request while its peers sit in that request's collectives: the client must get THAT rank's traceback
    within seconds (not a collective timeout), tear the whole pool down (the survivors can never complete), refuse further requests
    cleanly, and leave no worker process and no tmpfs blob behind."""
    import glob
    import time
    import mgpu_factory as F
    from infinicube.videogen import WanVideoGenerator
    from infinicube_amd.videogen import synthetic as syn
    path = _checkpoint(tmp_path)
    monkeypatch.setenv("ICV_WORLD", "2")
    monkeypatch.setenv("ICV_DIST_BACKEND", "gloo")
    monkeypatch.setenv("ICV_WORKER_FACTORY", "mgpu_factory:flaky_factory")
    monkeypatch.setenv("ICV_WORLD_TIMEOUT_S", "300")
    monkeypatch.setenv("PYTHONPATH", os.pathsep.join([os.path.dirname(HERE), HERE, os.environ.get("PYTHONPATH", "")]))
    sem, co = syn.make_dummy_buffers(F.GRID)
    blobs_before = set(glob.glob("/dev/shm/icv_pool_*"))
    g = None
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            g = WanVideoGenerator(path, device="cpu", use_wan_1pt3b=True, pipeline_factory=F.factory)
            assert len(g.generate(sem, co, seed=3)) == F.GRID.num_frames
            pids = [r["pid"] for r in g._pool.plan_record()["ranks"]]
            t0 = time.time()
            with pytest.raises(RuntimeError, match="synthetic failure in the middle of request 2 on rank 1"):
                g.generate(sem, co, seed=4)
            assert time.time() - t0 < 60
            assert g._pool._closed and not g._pool.procs
            with pytest.raises(RuntimeError, match="pool is closed"):
                g.generate(sem, co, seed=5)
        for pid in pids:
            assert not os.path.exists(f"/proc/{pid}") or open(f"/proc/{pid}/stat").read().split()[2] == "Z", f"rank process {pid} survived the pool"
        assert set(glob.glob("/dev/shm/icv_pool_*")) == blobs_before, "the request / frames blobs must go with the pool"
    finally:
        if g is not None and g._pool is not None:
            g._pool.close()
