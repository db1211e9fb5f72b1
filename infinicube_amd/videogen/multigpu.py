"""Multi-GPU generation behind the UNCHANGED single-process caller.

InfiniCube's stage 2 builds ONE ``WanVideoGenerator`` in its own process with the literal ``device="cuda:0"`` and never
touches ``torch.distributed`` [R infinicube/inference/guidance_buffer_generation.py:755-782]; launching that script N
times under ``torch.distributed.run`` would render the voxel world N times and is a change of the caller's contract.
So the N-GPU mode lives behind the constructor: with ``ICV_WORLD=N`` (or ``auto``) in the environment the caller's
process becomes rank 0 on GPU 0 and ``WorkerPool`` starts N-1 persistent worker processes (one per further GPU, this
module's ``__main__``), each of which builds the same generator on ``cuda:<rank>``, joins one process group
(``nccl`` = RCCL over xGMI; the K/V exchange and the velocity swap of seqpar.py run in it) and then serves
``generate`` commands.  Per call rank 0 broadcasts the request (prompt, seed, sampling settings and the two uint8
buffers, over a gloo control group on the host), every rank runs ``WanVideoPipeline.__call__`` on its token shard /
CFG branch, and only rank 0 decodes the latent, returns the frames and writes the mp4.

Not a scheduler or a serving layer: one generator, one request at a time, exactly the reference's usage.
"""
from __future__ import annotations

import atexit
import datetime
import importlib
import os
import pickle
import socket
import subprocess
import sys
import tempfile
import time
from typing import Optional

import numpy as np
import torch

# pipeline attributes rank 0 may have changed since construction; sent with every request so the ranks cannot drift
_PIPE_SETTINGS = ("num_inference_steps", "cfg_scale", "sigma_shift", "parallelism", "sp_chunks", "kv_exchange",
                  "reference_rounding", "gemm_dtype", "attn_dtype")


_ACTIVE_POOL = None   # the one WorkerPool of this process (the reference's caller keeps ONE generator per process)


def pool_for(world: int, ctor_kwargs: dict) -> "WorkerPool":
    """The process-wide pool: created on first use, reused by a later generator built with the same arguments; a
    generator with DIFFERENT arguments cannot share the workers (they hold the first one's weights) and is refused."""
    global _ACTIVE_POOL
    if _ACTIVE_POOL is not None and not _ACTIVE_POOL._closed:
        if _ACTIVE_POOL.ctor_kwargs == ctor_kwargs and _ACTIVE_POOL.world == world:
            return _ACTIVE_POOL
        raise RuntimeError("ICV_WORLD: this process already drives a worker pool built for another WanVideoGenerator "
                           f"({_ACTIVE_POOL.ctor_kwargs}); close it (generator._pool.close()) before building a different one")
    _ACTIVE_POOL = WorkerPool(world, ctor_kwargs)
    return _ACTIVE_POOL


def layout_cache():
    """The live pool's parallel-layout cache (process groups are a resource of the pool's process group), or None."""
    return _ACTIVE_POOL.layouts if (_ACTIVE_POOL is not None and not _ACTIVE_POOL._closed) else None


def requested_world() -> int:
    """ICV_WORLD = N | auto (= every visible GPU).  1 inside a worker, when unset, or when the process already is a
    rank of somebody else's job (torch.distributed initialised / launched by torch.distributed.run)."""
    v = os.environ.get("ICV_WORLD", "").strip().lower()
    if not v or os.environ.get("ICV_WORKER_RANK") is not None:
        return 1
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (_ACTIVE_POOL is None or _ACTIVE_POOL._closed):
        return 1          # somebody else's process group (e.g. torch.distributed.run): this process is already one rank of it
    if v == "auto":
        return max(1, torch.cuda.device_count())
    n = int(v)
    if n < 1:
        raise ValueError(f"ICV_WORLD must be a positive integer or 'auto', got {v!r}")
    return n


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _resolve(spec: str):
    mod, _, fn = spec.partition(":")
    return getattr(importlib.import_module(mod), fn)


class WorkerPool:
    """Rank 0's handle on the N-1 worker processes and the process group they share."""

    def __init__(self, world: int, ctor_kwargs: dict, backend: Optional[str] = None):
        import torch.distributed as dist
        self.world, self.dist, self.ctor_kwargs = world, dist, dict(ctor_kwargs)
        self._closed = True
        self._ready = False          # set by the first wait_ready(): the workers then sit in their serve loop
        self.layouts = {}            # seqpar.ParallelLayout per (world, rank, mode, cfg): shared by every pipeline of this process
        self.backend = backend or os.environ.get("ICV_DIST_BACKEND", "nccl")
        self.timeout_s = float(os.environ.get("ICV_WORLD_TIMEOUT_S", "3600"))
        port = _free_port()
        init_method = f"tcp://127.0.0.1:{port}"
        self._dir = tempfile.mkdtemp(prefix="icv_world_")
        spec = dict(ctor=ctor_kwargs, backend=self.backend, init_method=init_method, timeout_s=self.timeout_s,
                    factory=os.environ.get("ICV_WORKER_FACTORY"))
        spec_path = os.path.join(self._dir, "spec.pkl")
        with open(spec_path, "wb") as f:
            pickle.dump(spec, f)
        self.procs = []
        for r in range(1, world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), ICV_WORKER_RANK=str(r),
                       ICV_WORKER_SPEC=spec_path, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            env.pop("ICV_WORLD", None)
            log = open(os.path.join(self._dir, f"worker{r}.log"), "wb")
            self.procs.append((subprocess.Popen([sys.executable, "-m", "infinicube_amd.videogen.multigpu"], env=env,
                                                stdout=log, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL), log))
        # this process is rank 0 on GPU 0 ("cuda:0" already means that; LOCAL_RANK makes it explicit for resolve_device)
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        try:
            dist.init_process_group(self.backend, init_method=init_method, rank=0, world_size=world,
                                    timeout=datetime.timedelta(seconds=self.timeout_s))
            self.ctrl = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=self.timeout_s))
        except Exception:
            self._kill()
            raise
        self._closed = False
        atexit.register(self.close)
        print(f"[icvideo] {world} ranks: this process + {world - 1} workers (backend {self.backend}; logs in {self._dir})", file=sys.stderr)

    # -- rank 0 side -------------------------------------------------------------------------------
    def _check_alive(self):
        for i, (p, _) in enumerate(self.procs):
            rc = p.poll()
            if rc is not None:
                tail = ""
                try:
                    with open(os.path.join(self._dir, f"worker{i + 1}.log"), "rb") as f:
                        tail = f.read()[-2000:].decode(errors="replace")
                except OSError:
                    pass
                raise RuntimeError(f"multi-GPU worker rank {i + 1} exited with code {rc}:\n{tail}")

    def _blame_dead_worker(self, grace_s: float = 10.0):
        """A transport error usually means a worker is going down: give it a moment to exit, then report ITS log."""
        deadline = time.time() + grace_s
        while time.time() < deadline:
            self._check_alive()
            time.sleep(0.2)

    def wait_ready(self):
        """Every rank has built its generator (weights resident).  A second generator built with the same arguments
        reuses the pool (pool_for): its workers are already parked in their serve loop, where a barrier would pair with a
        broadcast and hang, so only the first call synchronises."""
        self._check_alive()
        if self._ready:
            return
        try:
            work = self.dist.barrier(group=self.ctrl, async_op=True)
            while not work.is_completed():      # a worker that dies while loading must not cost the whole collective timeout
                self._check_alive()
                time.sleep(0.2)
            work.wait()
            self._ready = True
        except RuntimeError as e:
            if "multi-GPU worker rank" not in str(e):
                self._blame_dead_worker()       # prefer the dead worker's own log to the transport's error
            raise

    def generate(self, semantic: np.ndarray, coordinate: np.ndarray, call_kwargs: dict, pipe) -> None:
        """Hand one request to the workers; the caller then runs its own share through ``pipe(...)``."""
        self._check_alive()
        if call_kwargs.get("seed") is None:
            raise ValueError("WorkerPool.generate: the caller resolves seed=None to one drawn integer for all ranks (WanVideoGenerator.generate does)")
        msg = dict(cmd="generate", shape=tuple(semantic.shape), call=call_kwargs,
                   settings={k: getattr(pipe, k) for k in _PIPE_SETTINGS if hasattr(pipe, k)})
        try:
            self.dist.broadcast_object_list([msg], src=0, group=self.ctrl)
            for arr in (semantic, coordinate):
                self.dist.broadcast(torch.from_numpy(np.ascontiguousarray(arr)), src=0, group=self.ctrl)
        except Exception:
            self._blame_dead_worker()
            raise

    def close(self):
        if getattr(self, "_closed", True):
            return
        self._closed = True
        try:
            # libicvideo's own communicators (seqpar._NativeComm) were built on this pool's process groups and go first, while
            # the peers are still alive (each worker closes its own on "exit"): a communicator that survived the pool would
            # pair this rank with dead peers when the next pool rendezvouses under the same cache key
            from .seqpar import _NativeComm
            _NativeComm.close_all()
        except Exception:
            pass
        try:
            if all(p.poll() is None for p, _ in self.procs):
                self.dist.broadcast_object_list([dict(cmd="exit")], src=0, group=self.ctrl)
            deadline = time.time() + 30
            for p, _ in self.procs:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    pass
        finally:
            self._kill()
            self.layouts.clear()     # the sub-groups go with the process group
            try:
                if self.dist.is_initialized():
                    self.dist.destroy_process_group()
            except Exception:
                pass
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                os.environ.pop(k, None)

    def _kill(self):
        for p, log in self.procs:
            if p.poll() is None:
                p.kill()          # exactly the PIDs this pool started
                p.wait()
            log.close()


# -- worker side ---------------------------------------------------------------------------------------
def _exit_when_parent_dies(parent_pid: int) -> None:
    """A worker blocked in a collective would outlive a crashed caller and keep its GPU: poll the parent and leave."""
    import threading

    def watch():
        while True:
            time.sleep(5.0)
            if os.getppid() != parent_pid:
                print("[worker] parent process is gone: exiting", flush=True)
                os._exit(3)

    threading.Thread(target=watch, daemon=True).start()


def worker_main() -> int:
    import torch.distributed as dist
    _exit_when_parent_dies(os.getppid())
    with open(os.environ["ICV_WORKER_SPEC"], "rb") as f:
        spec = pickle.load(f)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    to = datetime.timedelta(seconds=spec["timeout_s"])
    dist.init_process_group(spec["backend"], init_method=spec["init_method"], rank=rank, world_size=world, timeout=to)
    ctrl = dist.new_group(backend="gloo", timeout=to)
    from .inference import WanVideoGenerator
    factory = _resolve(spec["factory"]) if spec.get("factory") else None
    gen = WanVideoGenerator(**spec["ctor"], pipeline_factory=factory)   # device "cuda:0" resolves to cuda:LOCAL_RANK
    dist.barrier(group=ctrl)
    print(f"[worker {rank}] ready", flush=True)
    while True:
        box = [None]
        dist.broadcast_object_list(box, src=0, group=ctrl)
        msg = box[0]
        if msg["cmd"] == "exit":
            break
        bufs = []
        for _ in range(2):
            t = torch.empty(msg["shape"], dtype=torch.uint8)
            dist.broadcast(t, src=0, group=ctrl)
            bufs.append(t.numpy())
        for k, v in msg["settings"].items():
            setattr(gen.pipe, k, v)
        n, h, w, _ = msg["shape"]
        c = msg["call"]
        gen.pipe(prompt=c["prompt"], negative_prompt=c["negative_prompt"], semantic_buffer_video=gen._ndarray_to_pil_list(bufs[0]),
                 coordinate_buffer_video=gen._ndarray_to_pil_list(bufs[1]), height=h, width=w, num_frames=n, seed=c["seed"],
                 tiled=c["tiled"], return_latents=True, join_decode=True)   # rank 0 alone blends the decoded tiles (this rank
        #                                                             computes its share of them), returns frames, writes the mp4
        print(f"[worker {rank}] request done", flush=True)
    from .seqpar import _NativeComm
    _NativeComm.close_all()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(worker_main())
