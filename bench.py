#!/usr/bin/env python3
"""bench.py — denoise-steps/s of the buffer-conditioned Wan2.1 DiT loop on MI355X.

Metric (BASELINE.json): "denoise steps/sec ... for 93-frame 480p Wan2.1, 1/2/4/8 GPU".
One STEP = one scheduler step = 2 DiT forwards (cond + uncond, cfg 5) + fused unpatchify/CFG/Euler
over synthetic 93x480x832 latents (16x24x60x104, S = 37,440 tokens), random-init weights of the
named architecture, non-zero guidance-buffer tokens, inputs resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model 1.3b|14b]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

N > 1 = token-sequence parallelism (strong scaling: the same 37,440-token video is split over the
ranks; per-layer RCCL all-gather of K/V).  Rank 0 prints ONE JSON line — ALWAYS: for N > 1 every rank the
launcher starts is a supervisor (infinicube_amd/videogen/launch_guard.py) that runs the real rank as a child,
watches its phases (init / groups / setup / autotune / warmup / timed / report) against a budget, and on a raised
error or a hung phase moves all ranks to the next plan:  requested layout + measured K|V transport  ->  the same
layout with plain all_gather  ->  `sp` on the world group with all_gather (no sub-groups at all)  ->  a JSON line
with "error", the plan / rank / phase that failed and the worker's message, exit code 1.  `multi_gpu` records which
plan ran and what failed before it.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

# the host driver only supports dmabuf IPC: without this RCCL peer access fails with hipIpcGetMemHandle errors
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("NCCL_DEBUG", "WARN")   # RCCL problems of a first multi-GPU contact surface on stderr

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from infinicube_amd.videogen import launch_guard as guard  # noqa: E402
from infinicube_amd.videogen import synthetic as syn  # noqa: E402
from infinicube_amd.videogen.config import GRID_480P, TokenGrid, dit_forward_flops, preset  # noqa: E402
from infinicube_amd.videogen.dit import WanDiT  # noqa: E402
from infinicube_amd.videogen.ops import HipOps  # noqa: E402
from infinicube_amd.videogen.scheduler import FlowMatchScheduler  # noqa: E402
from infinicube_amd.videogen.seqpar import BranchExchange, ParallelLayout, autotune_kv_exchange  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
XGMI_LINKS, XGMI_LINK_GBS = 7, 153.0     # per GPU, point to point (task statement / MI355X guide)
PEAK_FP8_TFLOPS = 5000.0    # dense fp8 (K=64/128 scaled) MFMA peak, same guide
CFG_SCALE = 5.0


def cpu_baseline(cfg, grid, threads):
    """The CPU oracle (oracle/wan_ref.py, kind='port': the reference has no separate CPU DiT and its
    diffsynth dependency cannot travel) timed on this box's host cores on a BOUNDED sample of the
    same workload: ONE of the L layers; its token-local ops (LN/modulate, projections, RMSNorm+RoPE,
    cross-attention, FFN) on a 4096-token slice, its self-attention on a 512-query slice against all
    S keys; both scaled to S tokens, then x L layers x 2 forwards."""
    import dataclasses
    import torch.nn.functional as F
    from oracle import wan_ref as R
    torch.set_num_threads(threads)
    S, d, H = grid.S, cfg.dim, cfg.num_heads
    ts, qs = min(4096, S), min(512, S)
    one = dataclasses.replace(cfg, num_layers=1)
    sd = syn.make_dit_state_dict(one, seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((ts, d), generator=g)
    ctx = torch.randn((cfg.text_len, d), generator=g)
    t_mod = torch.randn((6, d), generator=g) * 0.1
    kf, vf = torch.randn((S, d), generator=g), torch.randn((S, d), generator=g)
    freqs = R.rope_freqs_3d(cfg.head_dim, grid.T, grid.Hp, grid.Wp)[:ts]
    p = "blocks.0"
    f32 = torch.float32
    t0 = time.perf_counter()
    mod = sd[f"{p}.modulation"].reshape(6, d) + t_mod
    h = R.modulate(R.layer_norm(x, None, None, cfg.eps), mod[0], mod[1])
    q = R.rope_apply(R.rms_norm(R._lin(sd, f"{p}.self_attn.q", h, f32), sd[f"{p}.self_attn.norm_q.weight"], cfg.eps), freqs, H)
    k = R.rope_apply(R.rms_norm(R._lin(sd, f"{p}.self_attn.k", h, f32), sd[f"{p}.self_attn.norm_k.weight"], cfg.eps), freqs, H)
    v = R._lin(sd, f"{p}.self_attn.v", h, f32)
    x = x + mod[2] * R._lin(sd, f"{p}.self_attn.o", v + k, f32)   # attention output stand-in (timed below)
    hh = R.layer_norm(x, sd[f"{p}.norm3.weight"], sd[f"{p}.norm3.bias"], cfg.eps)
    q2 = R.rms_norm(R._lin(sd, f"{p}.cross_attn.q", hh, f32), sd[f"{p}.cross_attn.norm_q.weight"], cfg.eps)
    k2 = R.rms_norm(R._lin(sd, f"{p}.cross_attn.k", ctx, f32), sd[f"{p}.cross_attn.norm_k.weight"], cfg.eps)
    v2 = R._lin(sd, f"{p}.cross_attn.v", ctx, f32)
    x = x + R._lin(sd, f"{p}.cross_attn.o", R.attention(q2, k2, v2, H), f32)
    h = R.modulate(R.layer_norm(x, None, None, cfg.eps), mod[3], mod[4])
    x = x + mod[5] * R._lin(sd, f"{p}.ffn.2", F.gelu(R._lin(sd, f"{p}.ffn.0", h, f32), approximate="tanh"), f32)
    t1 = time.perf_counter()
    a = R.attention(q[:qs], kf, vf, H)
    t2 = time.perf_counter()
    assert torch.isfinite(x).all() and torch.isfinite(a).all()
    t_local, t_attn = (t1 - t0) * (S / ts), (t2 - t1) * (S / qs)
    t_step = 2.0 * cfg.num_layers * (t_local + t_attn)
    return {
        "value": 1.0 / t_step, "unit": "denoise steps/s", "cores": threads, "kind": "port",
        "sample": (f"oracle/wan_ref.py fp32, 1 of {cfg.num_layers} layers: token-local ops on {ts} tokens "
                   f"({t1 - t0:.1f}s) and self-attention of {qs} queries x {S} keys ({t2 - t1:.1f}s), each scaled to "
                   f"S={S}, x{cfg.num_layers} layers x2 forwards = {t_step:.0f} s/step; host cpu_count={os.cpu_count()}"),
    }


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without torch.distributed.run: start N copies of this script, one per GPU, with the
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment torch.distributed.run would give them; rank 0 inherits this
    process's stdout (its ONE JSON line is the output), the other ranks' stdout goes to stderr.  A rank that fails takes
    the others down (exactly the PIDs started here) and its exit code is returned."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("ICV_BENCH_SHARE_GPU", "0") != "1":
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        env.setdefault("NCCL_DEBUG", "WARN")          # RCCL problems of a first multi-GPU contact surface on stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr, stdin=subprocess.DEVNULL))
    rc = 0
    try:
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                    for o in alive:
                        procs[o].terminate()
            time.sleep(0.2)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
                pr.wait()
    return rc


METRIC = "denoise steps/sec, 93-frame 480p Wan2.1 buffer-conditioned DiT loop"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    # default = Wan2.1-14B: BASELINE.json quotes its metric/target (>= 1.0 step/s on 8 GPUs, >= 6x scaling) and
    # its multi-GPU config on the 14B model, and 14B (28 GB bf16) fits one 288 GB MI355X
    ap.add_argument("--model", default=os.environ.get("ICV_BENCH_MODEL", "14b"),
                    choices=["1.3b", "14b", "14b-i2v", "small", "tiny", "tiny-i2v"])
    # bf16 is the metric's dtype; fp8 = BASELINE.json config #5's "fp8 MFMA weights" mode (e4m3 projections,
    # bf16 attention) and is reported as such, never as the headline number
    ap.add_argument("--gemm-dtype", default=os.environ.get("ICV_BENCH_GEMM_DTYPE", "bf16"), choices=["bf16", "fp8"])
    ap.add_argument("--attn-dtype", default=os.environ.get("ICV_BENCH_ATTN_DTYPE", "bf16"), choices=["bf16", "fp8"],
                    help="fp8: e4m3 self-attention (single-GPU runs of the fp8 mode only; reported as such)")
    ap.add_argument("--frames", type=int, default=GRID_480P.num_frames)
    ap.add_argument("--height", type=int, default=GRID_480P.height)
    ap.add_argument("--width", type=int, default=GRID_480P.width)
    ap.add_argument("--sp-chunks", type=int, default=int(os.environ.get("ICV_SP_CHUNKS", "4")),
                    help="N>1: the per-layer K/V all-gather is pipelined with attention in this many chunks")
    ap.add_argument("--parallelism", default=os.environ.get("ICV_PARALLELISM", "auto"), choices=["auto", "sp", "cfg+sp"],
                    help="N>1: 'sp' = token shards over all N ranks, both CFG forwards on every rank; 'cfg+sp' = cond / "
                         "uncond forwards on two groups of N/2 ranks, token shards inside a group (auto when N is even)")
    ap.add_argument("--kv-exchange", default=os.environ.get("ICV_KV_EXCHANGE", "auto"),
                    choices=["auto"] + [m + sfx for m in ("allgather", "p2p", "native", "ipc") for sfx in ("", "+arrival")],
                    help="N>1: K|V rows travel by all_gather_into_tensor (RCCL's schedule) or by grouped send/recv to every "
                         "peer (the direct, fully-connected schedule), or by libicvideo's own RCCL communicator (icv_allgather_kv; "
                         "seqpar.KVGather); 'auto' (default) = a start-up autotune times two real layers with each transport x "
                         "{4, 2} chunks on the ranks of the run and keeps the fastest (the table goes into multi_gpu.autotune); "
                         "'<transport>+arrival' = ONE arrival-gated attention launch per layer over the K|V pieces (csrc/attn7p.hip) "
                         "instead of one carried-state launch per row chunk")
    ap.add_argument("--rccl-max-channels", type=int, default=-1,
                    help="N>1: cap RCCL's channel count (NCCL_MAX_NCHANNELS; one channel = one resident workgroup beside the attention "
                         "kernel while a transfer runs): -1 (default) = 8 unless NCCL_MAX_NCHANNELS is already set - derived from the "
                         "measured slow-down of the shard-shape attention under k co-running copy work-groups, "
                         "profiles/r05/kv_contention.md; 0 = RCCL's own choice")
    ap.add_argument("--no-fallback", action="store_true",
                    help="N>1: run only the requested plan; a failure is reported instead of trying the simpler layouts")
    ap.add_argument("--native-forward", action="store_true",
                    help="drive each forward with ONE icv_dit_forward call (bf16, single GPU) instead of the per-op entry points; bit-identical")
    ap.add_argument("--share-stem", action="store_true",
                    help="let the uncond forward reuse the context-free stem (patch embed + layer 0's self-attention block) of the "
                         "cond forward, as the product pipeline does (bit-identical result, 1/80 less attention/QKV/O work). OFF by "
                         "default: the metric's step is two FULL forwards")
    ap.add_argument("--e2e", type=int, nargs="?", const=50, default=0, metavar="STEPS",
                    help="N=1: after the timed region ALSO run one whole WanVideoGenerator.generate() (93 frames 480p, tiled Wan-VAE, "
                         "UMT5, mp4 written; random-init weights of the real architectures) with STEPS denoising steps (default 50) and "
                         "add its wall-clock + stage table as `e2e` - the 'wall-clock' half of BASELINE.json's metric; ~3 min at 14B, so "
                         "off unless asked for (tools/e2e_wallclock.py is the same measurement on its own)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args(argv)


def error_record(args, world, message, failed=None, phase=None):
    """The line a run prints when it could not measure: same keys a reader expects, value null, the reason spelled out."""
    return {"metric": METRIC, "value": None, "unit": "denoise steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"Wan2.1-{args.model.upper()} DiT, {args.frames} frames {args.height}x{args.width}", "model": args.model},
            "error": message, "failed_phase": phase, "failed_attempts": failed or []}


def emit(obj, stdout_fd=None):
    """THE one line on stdout (fd 1 may have been pointed at stderr for the run: see run_rank)."""
    sys.stdout.flush()
    if stdout_fd is not None:
        os.dup2(stdout_fd, 1)
    print(json.dumps(obj), flush=True)
    if stdout_fd is not None:
        os.dup2(2, 1)


def plan_attempts(args, world):
    """The staged fallback of an N-rank run, most capable plan first (launch_guard.Attempt list)."""
    def resolved(mode):
        return ("cfg+sp" if world % 2 == 0 else "sp") if mode == "auto" else mode
    plans = [(resolved(args.parallelism), args.kv_exchange, args.sp_chunks)]
    if not args.no_fallback:
        if args.kv_exchange != "allgather":
            plans.append((resolved(args.parallelism), "allgather", args.sp_chunks))
        plans.append(("sp", "allgather", 4))          # no sub-groups, RCCL's own all-gather on the world group
    seen, out = set(), []
    for mode, kv, ch in plans:
        if (mode, kv, ch) in seen:
            continue
        seen.add((mode, kv, ch))
        out.append(guard.Attempt(f"{mode} / kv-exchange {kv} / {ch} chunks",
                                 {"ICV_BENCH_PLAN": json.dumps(dict(parallelism=mode, kv_exchange=kv, sp_chunks=ch))}))
    return out


def supervise(args, world, rank):
    """This process is one of the N the launcher started: run the real rank as a child under launch_guard, plan by plan."""
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)                       # nothing but rank 0's one line may reach stdout
    printed = {"done": False}

    def last_words():                   # the launcher is taking the ranks down (another rank's supervisor died)
        if rank == 0 and not printed["done"]:
            printed["done"] = True
            emit(error_record(args, world, "terminated by the launcher (SIGTERM) before a result existed", phase="supervisor"), stdout_fd)

    guard.install_sigterm(last_words)
    # per-step bound for the timed / warm-up budgets: generous (a 14B step on ONE GPU is 3.4 s)
    per_step = float(os.environ.get("ICV_BENCH_STEP_BUDGET_S", "8"))        # N >= 2: a 14B step is <= 1.7 s
    budgets = {"warmup": 90.0 + per_step * args.warmup, "timed": 60.0 + per_step * (args.steps + 1)}
    try:
        sup = guard.Supervisor(rank, world, plan_attempts(args, world), [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], budgets=budgets)
        res = sup.run()
    except BaseException as e:  # noqa: BLE001 - whatever happens, rank 0 says so on stdout
        if isinstance(e, SystemExit) and printed["done"]:
            raise
        if rank == 0 and not printed["done"]:
            printed["done"] = True
            emit(error_record(args, world, f"supervisor failed: {type(e).__name__}: {e}", phase="supervisor"), stdout_fd)
        return 1
    if rank == 0:
        printed["done"] = True
        if res["ok"]:
            out = res["result"]
            out.setdefault("multi_gpu", {})["plan"] = res["plan"]
            out["multi_gpu"]["failed_attempts"] = res["failed"]      # what was tried before the plan that ran (empty = first plan)
            emit(out, stdout_fd)
        else:
            last = res["failed"][-1] if res["failed"] else {}
            emit(error_record(args, world, f"every plan failed; last: rank {last.get('rank')} in phase '{last.get('phase')}': {last.get('reason', '')[-600:]}",
                              failed=res["failed"], phase=last.get("phase")), stdout_fd)
    # rank 0's line first, THEN anybody leaves: a launcher that sees a rank exit non-zero takes the others down at once
    try:
        if rank == 0:
            sup.store.set("printed", "1")
        else:
            t_end = time.time() + 30.0
            while time.time() < t_end and not sup.store.check(["printed"]):
                time.sleep(0.1)
    except Exception:
        pass
    return 0 if res["ok"] else 1


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N` (no launcher): become the launcher — one rank per GPU, exactly the processes
        # torch.distributed.run would start — and hand rank 0's JSON line through
        rc = self_launch(args.gpus)
        if rc == 2 and torch.cuda.device_count() < args.gpus:
            emit(error_record(args, args.gpus, f"--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) visible", phase="launch"))
        raise SystemExit(rc)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        # read when HIP / HSA initialise, i.e. at this process's FIRST HIP call (torch.cuda.is_available() is one): set here, before
        # anything touches the device, and in the supervised rank's environment before it exists (launch_guard._run_attempt).
        # The copy-engine K|V transport keeps one pull stream per peer next to the launch stream, and a pull that waits for its
        # peer's flag is a spinning kernel that blocks its HARDWARE queue (profiles/r05/kv_contention.md, "pending waits"): enough
        # queues that neither the launch stream nor another peer's copy ever sits behind one (7 pulls + launch + torch / RCCL streams)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 and not guard.supervised() and os.environ.get("ICV_BENCH_GUARD", "1") == "1":
        raise SystemExit(supervise(args, world, rank))
    if world != args.gpus:
        args.gpus = world        # the launcher's world size is authoritative
    plan_env = os.environ.get("ICV_BENCH_PLAN")
    if plan_env:                 # the supervisor's plan for this attempt overrides the command line
        for k, v in json.loads(plan_env).items():
            setattr(args, k, v)
    phase = guard.PhaseReporter(rank)
    stdout_fd = None
    try:
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        run_rank(args, world, rank, phase, stdout_fd)
    except BaseException as e:  # noqa: BLE001
        if guard.supervised() or isinstance(e, (SystemExit, KeyboardInterrupt)) and not isinstance(e, SystemExit):
            raise                # under a supervisor the exit code + log ARE the report
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        import traceback
        traceback.print_exc()
        if rank == 0:
            emit(error_record(args, world, f"{type(e).__name__}: {e}", phase=phase.current), stdout_fd)
        raise SystemExit(1)


def run_rank(args, world, rank, phase, stdout_fd):
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE JSON line and nothing else: native libraries write there too (RCCL's version banner and NCCL_DEBUG
    # output, gloo's "[Gloo] Rank ..." lines), so file descriptor 1 points at stderr for the whole run and is restored for
    # the final print
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible to PyTorch-ROCm: the denoising loop has no CPU path")
    # one process per GPU.  ICV_BENCH_SHARE_GPU=1 (+ ICV_DIST_BACKEND=gloo) lets several ranks share the only GPU of a
    # development box so the N>1 code path can be exercised there; it is never a measurement mode.
    share = os.environ.get("ICV_BENCH_SHARE_GPU", "0") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        phase("init")
        from infinicube_amd.videogen.seqpar import apply_rccl_channel_cap
        apply_rccl_channel_cap(args.rccl_max_channels)                          # read when the communicators are created
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ICV_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm
        # an explicit collective timeout: a wedged exchange must end THIS attempt (the supervisor then moves every rank to
        # the next plan), not sit out torch's 10-minute default
        dist_timeout = datetime.timedelta(seconds=float(os.environ.get("ICV_BENCH_DIST_TIMEOUT_S", "150")))
        # no device_id: communicators are created lazily on first use (the classic unique-id path).  With device_id
        # PyTorch initialises eagerly and builds the sub-groups of the cfg+sp layout with ncclCommSplit, which this
        # build could not exercise on RCCL; torch.cuda.set_device above already binds the rank to its GPU.
        dist.init_process_group(backend=backend, timeout=dist_timeout)
        on_dev = dist.get_backend() == "nccl"
        one = torch.ones(1, device=device if on_dev else "cpu")
        dist.all_reduce(one)                         # the world communicator exists (and works) before anything else is built
        assert int(one.item()) == world

    cfg = preset(args.model)
    grid = TokenGrid(args.frames, args.height, args.width)
    if world > 1:
        phase("groups")
    layout = ParallelLayout.make(world, rank, args.parallelism, use_cfg=True)
    plan = layout.shard_plan(grid.S)
    if world > 1:
        # first collective on every group this rank will use, inside the watched phase: RCCL builds a communicator lazily
        # at its first use, and that is where a first contact with a new node goes wrong
        for g in (layout.sp_group, layout.pair_group):
            if g is not None:
                m = dist.get_world_size(g)
                src = torch.full((4,), float(rank), device=device if on_dev else "cpu")
                dst = torch.empty((4 * m,), device=src.device)
                dist.all_gather_into_tensor(dst, src, group=g)
                assert dst.view(m, 4)[:, 0].tolist() == [float(r) for r in dist.get_process_group_ranks(g)], "group smoke test: wrong ranks"
        phase("setup")
    ops = HipOps(device)

    # ---- synthetic weights / inputs, resident in HBM before timing (SURVEY.md §8d recipe) ----
    sd = syn.make_dit_state_dict(cfg, seed=0, device=device, dtype=torch.bfloat16)
    bsd = syn.make_buffer_embedder_state_dict(cfg, device=device, dtype=torch.bfloat16)
    model = WanDiT(cfg, sd, ops, bsd, gemm_dtype=args.gemm_dtype, attn_dtype=args.attn_dtype)
    del sd, bsd
    # graphs off: the bench times individual attention launches with events (at the metric's size the loop is GPU-bound
    # and "auto" would not capture anyway)
    kv_auto = args.kv_exchange == "auto"
    model.prepare(grid, plan, sp_chunks=args.sp_chunks, group=layout.sp_group, graphs=False,
                  kv_exchange="allgather" if kv_auto else args.kv_exchange)
    model.share_stem = bool(args.share_stem)
    if args.native_forward:   # one icv_dit_forward call per forward instead of ~530 per-op calls (bit-identical)
        model.native_forward = True
        if not model._native_eligible():
            raise SystemExit("--native-forward covers the bf16 single-GPU path only")
    clip = syn.make_clip_features(cfg) if cfg.has_image_input else None
    ctx_c = model.encode_context(syn.make_text_context(cfg, 1), clip)
    ctx_u = model.encode_context(syn.make_text_context(cfg, 2), clip)
    buf = model.embed_buffers(syn.make_buffer_latents(cfg, grid))
    if cfg.has_image_input:   # i2v: first-frame conditioning latent folded into the cached additive tokens
        buf = model.embed_cond_latents(syn.make_cond_latents(cfg, grid), add_to=buf)
    latent = syn.make_latent_noise(grid, seed=0).to(device)
    total_steps = 50
    sched = FlowMatchScheduler(total_steps)

    # ---- per-launch timing of the dominant kernel (self-attention) with events on the launch stream
    attn_events = []
    record = {"on": False}
    raw_attention = ops.attention

    def timed_attention(q, k, v, o, heads, scale):
        if record["on"] and k.shape[0] == grid.S:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            raw_attention(q, k, v, o, heads, scale)
            e1.record()
            attn_events.append((e0, e1))
        else:
            raw_attention(q, k, v, o, heads, scale)

    ops.attention = timed_attention
    raw_attention8 = ops.attention_fp8

    def timed_attention8(q, k, v, o, heads, ws):     # fp8 mode: prepare + forward timed together
        if record["on"]:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            raw_attention8(q, k, v, o, heads, ws)
            e1.record()
            attn_events.append((e0, e1))
        else:
            raw_attention8(q, k, v, o, heads, ws)

    ops.attention_fp8 = timed_attention8
    raw_chunk = ops.attention_chunk
    chunk_events = []

    def timed_chunk(q, k, v, o, acc, ml, heads, scale, first, last):
        if record["on"]:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            raw_chunk(q, k, v, o, acc, ml, heads, scale, first, last)
            e1.record()
            chunk_events.append((e0, e1, k.shape[0]))
        else:
            raw_chunk(q, k, v, o, acc, ml, heads, scale, first, last)

    ops.attention_chunk = timed_chunk
    raw_pieces = ops.attention_pieces
    piece_events = []

    def timed_pieces(q, pieces, o, heads, scale, **kw):     # arrival-driven: ONE launch per layer over all S keys
        if record["on"]:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            raw_pieces(q, pieces, o, heads, scale, **kw)
            e1.record()
            piece_events.append((e0, e1, sum(int(p[0].shape[0]) for p in pieces)))
        else:
            raw_pieces(q, pieces, o, heads, scale, **kw)

    ops.attention_pieces = timed_pieces

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier(device_ids=[dev_index]) if dist.get_backend() == "nccl" else dist.barrier()
            torch.cuda.synchronize(device)

    xchg = BranchExchange(layout) if layout.mode == "cfg+sp" else None

    # ---- N > 1: which K|V transport / chunking?  measured on these ranks, not assumed (seqpar.autotune_kv_exchange)
    autotune = None
    if world > 1 and kv_auto:
        if layout.sp_world > 1:
            phase("autotune")
            own_ctx = ctx_u if layout.branch == 1 else ctx_c

            def two_layers():
                model.forward_tokens(latent, own_ctx, 500.0, buf, model.head_own, num_layers=min(2, cfg.num_layers))

            def reduce_max(vals):
                t = torch.tensor(vals, dtype=torch.float64, device=device if on_dev else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return t.tolist()

            t_tune = time.perf_counter()
            # per transport: the arrival-driven attention (bf16: one launch per layer; e4m3 blobs on the wire: the chunk launches gate on their
            # blobs inside the kernel) and the host-waited chunk launches
            arrival = [("+arrival", args.sp_chunks)] if ((args.attn_dtype == "bf16" or getattr(model, "fp8_wire", False))
                                                          and os.environ.get("ICV_ATTN_ARRIVAL") != "0") else []
            cands = [(m + sfx, c) for m in ("allgather", "p2p", "native", "ipc")
                     for sfx, c in arrival + [("", c) for c in sorted({args.sp_chunks, 2}, reverse=True)]]
            if share:      # several ranks on ONE GPU (development boxes): RCCL refuses duplicate devices in a communicator
                cands = [mc for mc in cands if not mc[0].startswith("native")]
            peers = layout.sp_world - 1
            link_peak = XGMI_LINK_GBS * min(peers, XGMI_LINKS)          # what the links of THIS rank can deliver to it (one per peer)

            def attn_probe(mode, chunks):
                """First contact made self-explaining, per candidate on the ranks of the run:
                  attn_under_exchange_ms / attn_from_memory_ms - one layer's self-attention (all its launches) with the REAL exchange in
                    flight in front of it (issued right before: the transfer hides under nothing else) vs the same launches over rows that
                    are already there: the difference is what the transfer exposes PLUS what its data movement costs the kernel beside it;
                  ipc_peer_copy - does a pull from the right-hand neighbour need compute units (icv_ipc_probe_copy: blit kernel) or not
                    (copy engine)?  Nothing in a timing table would tell."""
                if model.attn_fp8 and not getattr(model, "fp8_wire", False):
                    return {}
                H, q = cfg.num_heads, model.qkv[0]
                out = {}
                for key, under in (("attn_under_exchange_ms", True), ("attn_from_memory_ms", False)):
                    ms = []
                    for rep in range(3):
                        sync()
                        if under:
                            model._sp_acquire()
                            handles, bufs = model._sp_start_gather()
                        else:
                            handles, bufs = model._sp_start_gather(start=False)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        model._sp_attention(q, handles, bufs, H, model.attn_scale, from_memory=not under)
                        e1.record()
                        torch.cuda.synchronize(device)
                        if rep:
                            ms.append(e0.elapsed_time(e1))
                    out[key] = sum(ms) / len(ms)
                out["attn_under_exchange_ms"], out["attn_from_memory_ms"] = reduce_max([out["attn_under_exchange_ms"], out["attn_from_memory_ms"]])
                heap = getattr(model.kv_gather, "_heap", None)
                if heap is not None and peers > 0:
                    # the ranks take turns: the probe fills its device with an occupier kernel, which ranks that SHARE a GPU (the 1-GPU
                    # rehearsal) would do to each other, and on a real node a neighbour's concurrent pull would muddy the copy's timing
                    for turn in range(world):
                        sync()
                        if turn == rank:
                            kind, ms_copy = heap.probe_copy((layout.sp_rank + 1) % layout.sp_world)
                            out["ipc_peer_copy"] = kind
                            out["ipc_peer_copy_8mib_ms"] = ms_copy
                    sync()
                return out

            def exchange_only():          # one layer's K|V exchange with nothing to hide under: the raw transfer
                handles, _ = model._sp_start_gather()
                for h in handles:
                    model.kv_gather.wait(h)
                    if hasattr(model.kv_gather, "consumed"):
                        model.kv_gather.consumed(h)

            (args.kv_exchange, args.sp_chunks), table = autotune_kv_exchange(
                model, two_layers, sync, cands, reps=2, reduce_max=reduce_max, exchange_only=exchange_only, probe=attn_probe,
                on_candidate=lambda m, c: phase(f"autotune:{m}/{c}"),          # one supervisor phase (its own budget) per candidate
                log=(lambda m: print(f"[bench] {m}", file=sys.stderr, flush=True)) if rank == 0 else None)
            recv = 2 * 2 * plan.n_tok * cfg.dim * (layout.sp_world - 1)         # bytes each rank RECEIVES per layer exchange (k | v rows, bf16)
            for row in table:
                if row.get("exchange_ms"):
                    row["recv_gb_per_s_per_rank"] = recv / (row["exchange_ms"] * 1e-3) / 1e9
                    row["frac_of_xgmi_links"] = row["recv_gb_per_s_per_rank"] / link_peak if (link_peak and not share) else None
                if row.get("attn_under_exchange_ms") and row.get("attn_from_memory_ms"):
                    row["attn_slowdown_under_exchange"] = row["attn_under_exchange_ms"] / row["attn_from_memory_ms"]
            autotune = {"seconds": time.perf_counter() - t_tune, "layers_timed": min(2, cfg.num_layers), "table": table,
                        "bytes_received_per_rank_per_layer_exchange": recv,
                        "xgmi_peak_gb_per_s_into_one_rank": link_peak, "xgmi_note": f"{XGMI_LINKS} links x {XGMI_LINK_GBS:.0f} GB/s per GPU, point to point: one link per peer"
                        + (" (ranks SHARE one GPU here: no link is involved)" if share else ""),
                        "columns": {"ms": "two real layers, max over ranks (the choice is made on this)", "exchange_ms": "one layer's exchange with nothing to hide under",
                                    "attn_under_exchange_ms": "one layer's self-attention launches with the real exchange issued right in front of them",
                                    "attn_from_memory_ms": "the same launches over rows that are already there",
                                    "ipc_peer_copy": "icv_ipc_probe_copy: does a pull need compute units (blit kernel) or not (copy engine)"},
                        "chosen": {"kv_exchange": args.kv_exchange, "sp_chunks": args.sp_chunks}}
        else:
            args.kv_exchange = "allgather"          # cfg+sp at N = 2: no K|V exchange at all
    phase("warmup")

    def run_steps(first, count):
        model.denoise(latent, ctx_c if layout.branch in (None, 0) else None, ctx_u if layout.branch in (None, 1) else None,
                      buf, sched, CFG_SCALE, steps=[(first + i) % total_steps for i in range(count)], branch_exchange=xchg)

    from infinicube_amd import native
    run_steps(0, args.warmup)
    sync()
    phase("timed")
    record["on"] = True
    if args.native_forward:
        model.native_profile(True)
        if model._twin is not None:      # dual-stream CFG: the second forward runs on a twin engine with its own context
            model._twin[0].native_profile(True)
    if model.kv_gather is not None:
        model.kv_gather.timing = []          # (event before, event after) each wait on a K|V chunk: exposed transfer time
        model.kv_gather.n_collectives = 0
    calls0 = native.N_CALLS[0]
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps)
    t_in_loop = time.perf_counter() - t0     # host time inside the timed loop: issue time PLUS blocking on a full HIP queue
    sync()
    elapsed = time.perf_counter() - t0
    record["on"] = False
    phase("report")
    abi_calls = native.N_CALLS[0] - calls0
    kv_waits = list(model.kv_gather.timing or []) if model.kv_gather is not None else []
    n_coll = model.kv_gather.n_collectives if model.kv_gather is not None else 0
    if model.kv_gather is not None:
        model.kv_gather.timing = None
    native_reads = None
    if args.native_forward:   # events recorded by the C driver around its own self-attention launches of the TIMED steps
        native_reads = [model.native_profile_read()] + ([model._twin[0].native_profile_read()] if model._twin is not None else [])
        model.native_profile(False)
        if model._twin is not None:
            model._twin[0].native_profile(False)
    # host ISSUE time of one step, measured where nothing back-pressures it: the queue is empty (sync above) and one step's
    # launches fit in it, so this is what the host needs to enqueue a step; the GPU then runs it (outside the timed region)
    t1 = time.perf_counter()
    run_steps(args.warmup + args.steps, 1)
    t_issue_one = time.perf_counter() - t1
    sync()
    comm = None
    if world > 1:
        waits = kv_waits                                   # cfg+sp at N=2: no K|V exchange at all
        exposed_ms = sum(a.elapsed_time(b) for a, b in waits)
        tt = torch.tensor([exposed_ms, float(len(waits)), float(n_coll)], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        comm = {"rccl_ranks": world, "backend": dist.get_backend(), "kv_exchange": args.kv_exchange, "sp_chunks": args.sp_chunks,
                "kv_group_ranks": layout.sp_world, "rccl_max_channels": os.environ.get("NCCL_MAX_NCHANNELS") or "rccl default",
                "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES") or "runtime default (4)",
                "attention": ("one carried-state launch per row chunk, each gating on its e4m3 blobs' arrival flags inside the kernel (csrc/attn8.hip)"
                              if getattr(model, "fp8_wire", False) else "ONE arrival-gated launch per layer over the K|V pieces (csrc/attn7p.hip)")
                if getattr(model, "attn_arrival", False) else "one carried-state launch per row chunk",
                "kv_exchanges_per_step_per_rank": tt[2].item() / args.steps,
                # what one rank SENDS per layer exchange: its k | v rows in bf16, or (e4m3 on the wire) its e4m3 blobs
                "kv_bytes_sent_per_exchange_layer": 0 if layout.sp_world <= 1 else
                (sum(r8 for _, r8, _, _ in model._kv8_chunks) * cfg.dim if getattr(model, "fp8_wire", False) else 2 * 2 * plan.n_tok * cfg.dim),
                "kv_wire_format": "e4m3 blobs (quantised once per rank)" if getattr(model, "fp8_wire", False) else "bf16 rows",
                # max over ranks of the compute-stream stalls in front of the chunk launches; the arrival-driven attention has no such
                # stall to time (it waits INSIDE its one launch): multi_gpu.autotune's attn_under_exchange_ms - attn_from_memory_ms is its exposure
                "exposed_kv_wait_ms_per_step": None if getattr(model, "attn_arrival", False) else tt[0].item() / args.steps,
                "kv_chunk_waits_per_step": tt[1].item() / args.steps}
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(latent).all(), "latent went non-finite"

    if piece_events:   # N>1, arrival-driven: one launch per layer over all keys of the sequence
        attn_ms = sum(a.elapsed_time(b) for a, b, _ in piece_events) / len(piece_events)
        attn_flops = 4.0 * plan.n_tok * (sum(kk for _, _, kk in piece_events) / len(piece_events)) * cfg.dim
        attn_events = piece_events
    elif chunk_events:   # N>1: one self-attention = sp_chunks launches over S/sp_chunks keys each
        attn_ms = sum(a.elapsed_time(b) for a, b, _ in chunk_events) / len(chunk_events)
        attn_flops = 4.0 * plan.n_tok * (sum(kk for _, _, kk in chunk_events) / len(chunk_events)) * cfg.dim
        attn_events = chunk_events
    elif args.native_forward:   # events recorded by the C driver around its own self-attention launches
        reads = native_reads
        n_timed = sum(n for _, n in reads)
        attn_ms = sum(ms for ms, _ in reads) / max(n_timed, 1)
        attn_events = [None] * n_timed
        attn_flops = 4.0 * plan.n_tok * grid.S * cfg.dim
        if model.sp_on:           # the C driver times each key-chunk launch: average flops of one chunk launch
            attn_flops /= (len(model.sp_bounds) - 1)
    else:
        attn_ms = sum(a.elapsed_time(b) for a, b in attn_events) / max(len(attn_events), 1)
        attn_flops = 4.0 * plan.n_tok * grid.S * cfg.dim        # per launch on this rank (4 S^2 d)
    attn_tflops = attn_flops / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else 0.0
    f_step = 2.0 * dit_forward_flops(cfg, grid.S)
    attn_peak = PEAK_BF16_TFLOPS if args.attn_dtype == "bf16" else PEAK_FP8_TFLOPS   # the dominant kernel's own MFMA peak

    traffic, traffic_source = None, None
    try:   # HBM bytes per self-attention launch, measured offline with rocprofv3 PMC passes (see profiles/)
        tj = json.load(open(os.path.join(ROOT, "profiles", "attn_traffic.json")))
        if world == 1 and args.model in tj and (args.frames, args.height, args.width) == (93, 480, 832):
            traffic = tj[args.model]["hbm_bytes_per_launch"]
            traffic_source = tj[args.model].get("source")
    except Exception:
        traffic, traffic_source = None, None
    if rank == 0:
        out = {
            "metric": METRIC,
            "value": args.steps / elapsed,
            "unit": "denoise steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "bf16" if (args.gemm_dtype, args.attn_dtype) == ("bf16", "bf16") else
                     f"projections {args.gemm_dtype}, self-attention {args.attn_dtype} (e4m3 operands, f32 accumulate), rest bf16/f32",
            "data": "synthetic (seeded latents/text context/guidance-buffer latents, random-init weights of the named architecture)",
            "config": {
                "workload": f"Wan2.1-{args.model.upper()} {'i2v' if cfg.has_image_input else 't2v'} DiT, {args.frames} frames {args.height}x{args.width}, "
                            f"S={grid.S} tokens, 1 step = 2 DiT forwards (cond+uncond, cfg {CFG_SCALE}) + Euler; "
                            f"50-step flow-match schedule (shift 5)",
                "model": cfg.name, "gemm_dtype": args.gemm_dtype, "tokens": grid.S, "layers": cfg.num_layers, "dim": cfg.dim,
                "parallelism": "single-gpu" if world == 1 else (
                    f"sp{world} (token-sequence shards over all ranks, K/V all-gather in {args.sp_chunks} chunks overlapped with attention)"
                    if layout.mode == "sp" else
                    f"cfg2 x sp{layout.sp_world} (cond / uncond forwards on two groups of {layout.sp_world} ranks; token-sequence shards and "
                    f"K/V all-gather in {args.sp_chunks} chunks inside a group; one velocity swap per step between the groups)"),
                "wallclock_50_steps_s": 50.0 * elapsed / args.steps,
                "cfg_stem_shared": bool(args.share_stem),
                # the two forwards of a step issued as ONE batch of 2n rows through the token-local kernels (WanDiT.forward_pair;
                # bit-identical to two sequential forwards, ICV_CFG_BATCH=0 switches it off): same work, better tile quantisation
                # (single rank, and every rank of the `sp` layout)
                "cfg_forwards_batched": bool(layout.mode == "sp" and model._pair_ok()),     # cfg+sp: one forward per rank, nothing to batch
                "c_abi_calls_per_forward": abi_calls / (args.steps * (1 if layout.mode == "cfg+sp" else 2)),
                "host_enqueue_ms_per_step": 1e3 * t_issue_one,       # one extra step issued on a drained queue (untimed for the metric)
                "host_ms_in_timed_loop_per_step": 1e3 * t_in_loop / args.steps,   # includes blocking inside launches once the HIP queue is full: NOT issue cost
                "forward_driver": "icv_dit_forward (C)" if args.native_forward else "per-op C entry points driven from videogen/dit.py",
                "algorithmic_pflop_per_step": f_step / 1e15,
                "model_tflops_all_gpus": f_step * args.steps / elapsed / 1e12,
                "frac_of_bf16_mfma_peak": f_step * args.steps / elapsed / 1e12 / (PEAK_BF16_TFLOPS * world),
            },
            "roofline": {
                "kernel": ("att7p::attn7p_kernel (self-attention, K6: ONE arrival-gated launch per layer over the K|V pieces; the time includes "
                           "whatever the launch waited for rows inside)" if piece_events else
                           "att7p::attn7p_kernel, one piece (self-attention, K6: attn7's schedule in the pieces kernel, bit-identical, 1.5-2.7 % faster)") if args.attn_dtype == "bf16" else
                          "att8::attn8_kernel + its quantise pre-pass (e4m3 self-attention, K6)",
                "bound": "mfma", "achieved": attn_tflops, "peak": attn_peak, "unit": "TFLOP/s",
                "frac": attn_tflops / attn_peak, "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_ms": attn_ms, "launches_timed": len(attn_events),
                "algorithmic_flop_per_launch": attn_flops,
            },
        }
        if comm is not None:
            comm["autotune"] = autotune
            comm["parallelism"] = layout.mode
            try:      # what the first contact ran on: library version, device, the RCCL / HSA switches in the environment
                comm["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                comm["rccl_version"] = None
            comm["device_name"] = torch.cuda.get_device_name(device)
            comm["env"] = {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_", "ICV_")) and "INJECT" not in k}
            out["multi_gpu"] = comm
        if world == 1 and args.e2e:
            phase("e2e")
            del model, ctx_c, ctx_u, buf
            torch.cuda.empty_cache()
            import importlib.util
            spec = importlib.util.spec_from_file_location("_icv_e2e", os.path.join(ROOT, "tools", "e2e_wallclock.py"))
            e2e_mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(e2e_mod)
            out["e2e"] = e2e_mod.run_e2e(args.model, args.e2e, args.gemm_dtype, str(device), log=lambda m: print(f"[e2e] {m}", file=sys.stderr, flush=True))
            out["e2e"]["loop_only_50_steps_s"] = out["config"]["wallclock_50_steps_s"]
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or min(os.cpu_count() or 1, 32)  # >32 threads oversubscribes these GEMM sizes
            out["cpu_baseline"] = cpu_baseline(cfg, grid, threads)
        if not guard.write_result(out):      # under a supervisor ITS rank 0 prints the line (with the attempt history)
            emit(out, stdout_fd)
    if world > 1:
        from infinicube_amd.videogen.seqpar import _NativeComm
        _NativeComm.close_all()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
