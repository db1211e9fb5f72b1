"""N real ranks, one per GPU, over RCCL (BASELINE.json configs #4 / #5) — and the CPU twin of every case.

The `-m gpu` tests here start N processes with the `nccl` backend (= RCCL over xGMI) when the box has >= 2 GPUs and
compare their result with the 1-GPU result of the SAME worker script (tests/rank_worker.py); on a 1-GPU box they skip
with the reason "needs >= 2 GPUs" (the driver's round-end box has one GPU; an 8-GPU node runs them for real).  The
twins run the same launcher, worker and comparisons here on CPU over gloo with the test-only oracle operator set, so the
harness itself is known to work before it first meets a multi-GPU node.

Bars: N-GPU vs 1-GPU "should be ~ bit-close: only the softmax merge order changes".  With the fp32-exact
oracle operator set (CPU twins) that is rel-L2 <= 2e-3 / PSNR >= 55 dB of the final latent; with the HIP kernels the merge
order also moves bf16 roundings of P and of the stored activations, which a 3-step CFG-5 loop amplifies: measured 2.7e-3 /
64 dB on 4 processes sharing one GPU, bar rel-L2 <= 1e-2 / PSNR >= 50 dB (one 14B block: 2e-3 on the residual stream).
Every rank ends with the identical gathered result (asserted inside the worker); the bench line for N > 1 carries
multi_gpu.rccl_ranks == N and exposed_kv_wait_ms_per_step.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
WORKER = os.path.join(HERE, "rank_worker.py")


def _n_gpus() -> int:
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def needs_gpus(n):
    return pytest.mark.skipif(_n_gpus() < n, reason=f"needs >= {n} GPUs (this box has {_n_gpus()})")


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(n, out, args, timeout=420, extra_env=None):
    """Start n copies of the worker (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them), wait,
    fail with the failing rank's output; returns the dict rank 0 saved."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PYTHONPATH=os.pathsep.join([ROOT, HERE, os.environ.get("PYTHONPATH", "")]), OMP_NUM_THREADS="2")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("GPU_MAX_HW_QUEUES", "16")      # what every real launcher of ranks sets (the copy-engine transport refuses a process without it)
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, WORKER, "--out", out] + [str(x) for x in args], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL))
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            outs.append(o.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()          # exactly the PIDs started here
                p.wait()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} of {n} failed (rc {p.returncode}):\n{outs[r][-3000:] if r < len(outs) else ''}"
    return torch.load(out)


def _close(got, ref, what, rel_bound=2e-3, psnr_bound=55.0):
    from oracle import wan_ref as R
    a, b = got["result"], ref["result"]
    assert a.shape == b.shape
    rel = float((a - b).norm() / b.norm())
    p = R.psnr(a, b)
    print(f"{what}: {got['info']} vs single rank: rel-L2 {rel:.3g}, PSNR {p:.1f} dB")
    assert rel <= rel_bound and p >= psnr_bound, f"{what}: N-rank result differs from the single-rank one: rel-L2 {rel}, PSNR {p:.1f} dB"


# ---------------------------------------------------------------------------------------------------------------------
# CPU twins (gloo + the oracle operator set): same launcher, same worker, same comparisons
# ---------------------------------------------------------------------------------------------------------------------
CPU = ["--backend", "gloo", "--ops", "oracle", "--model", "tiny", "--frames", 9, "--height", 64, "--width", 96]


@pytest.fixture(scope="module")
def cpu_single(tmp_path_factory):
    d = tmp_path_factory.mktemp("ranks_cpu")
    return {sc: run_ranks(1, str(d / f"single_{sc}.pt"), CPU + ["--scenario", sc]) for sc in ("loop", "layer")}


@pytest.mark.parametrize("world,parallelism,kv_exchange", [(2, "sp", "p2p"), (4, "cfg+sp", "allgather")])
def test_twin_loop_ranks_equal_single(tmp_path, cpu_single, world, parallelism, kv_exchange):
    got = run_ranks(world, str(tmp_path / "multi.pt"), CPU + ["--scenario", "loop", "--parallelism", parallelism, "--kv-exchange", kv_exchange])
    assert got["info"]["world"] == world and got["info"]["backend"] == "gloo" and got["info"]["kv_collectives"] > 0
    _close(got, cpu_single["loop"], f"gloo x{world} {parallelism}/{kv_exchange}")


def test_twin_layer_ranks_equal_single(tmp_path, cpu_single):
    got = run_ranks(3, str(tmp_path / "multi.pt"), CPU + ["--scenario", "layer", "--kv-exchange", "allgather"])
    assert got["info"]["mode"] == "sp" and got["info"]["sp_world"] == 3
    _close(got, cpu_single["layer"], "gloo x3 one block")


def test_twin_fp8_mode_ranks_equal_single(tmp_path):
    """The e4m3 mode on the sharded path (every gathered K|V chunk quantised on its own): e4m3-level agreement."""
    args = CPU + ["--scenario", "loop", "--gemm-dtype", "fp8", "--attn-dtype", "fp8"]
    ref = run_ranks(1, str(tmp_path / "single.pt"), args)
    got = run_ranks(2, str(tmp_path / "multi.pt"), args + ["--parallelism", "sp"])
    _close(got, ref, "gloo x2 fp8 mode", rel_bound=6e-2, psnr_bound=30.0)


def test_twin_failing_rank_is_reported(tmp_path):
    with pytest.raises(AssertionError, match="rank . of 2 failed"):
        run_ranks(2, str(tmp_path / "x.pt"), CPU + ["--scenario", "loop", "--parallelism", "tp"])


# ---------------------------------------------------------------------------------------------------------------------
# real RCCL ranks (one per GPU)
# ---------------------------------------------------------------------------------------------------------------------
GPU_TINY = ["--backend", "nccl", "--ops", "hip", "--model", "small", "--frames", 17, "--height", 128, "--width", 160]


@pytest.fixture(scope="module")
def gpu_single(tmp_path_factory):
    d = tmp_path_factory.mktemp("ranks_gpu")
    return {"loop": run_ranks(1, str(d / "single_loop.pt"), GPU_TINY + ["--scenario", "loop"])}


@pytest.mark.gpu
def test_shared_gpu_gloo_p2p_transport_four_ranks_sp(tmp_path, gpu_single):
    """The grouped send / recv transport under the `sp` layout with 4 ranks (both CFG branches' exchanges of a layer in flight
    together, 3 peers each) between processes sharing the GPU - the combination bench.py's autotune picks on such a box."""
    args = ["--backend", "gloo", "--share-gpu"] + GPU_TINY[2:] + ["--scenario", "loop", "--parallelism", "sp", "--kv-exchange", "p2p"]
    got = run_ranks(4, str(tmp_path / "multi.pt"), args)
    _close(got, gpu_single["loop"], "gloo x4 on one GPU, sp / p2p", rel_bound=1e-2, psnr_bound=50.0)


@pytest.mark.gpu
@pytest.mark.parametrize("world,parallelism", [(2, "sp"), (4, "cfg+sp")])
def test_shared_gpu_gloo_ranks_equal_single_gpu(tmp_path, gpu_single, world, parallelism):
    """What a ONE-GPU box can run of the multi-process path: N real processes, each with the PRODUCT operator set (HIP
    kernels through the C ABI) on the one GPU they share, exchanging their K|V rows / velocity tokens / latents through a real
    process group (gloo: RCCL refuses two ranks on one device).  Everything of configs #4 / #5 except the RCCL transport
    itself: shard plans, RoPE offsets, chunked carried-state attention, the cfg+sp velocity swap, the final latent gather."""
    args = ["--backend", "gloo", "--share-gpu"] + GPU_TINY[2:] + ["--scenario", "loop", "--parallelism", parallelism, "--kv-exchange", "allgather"]
    got = run_ranks(world, str(tmp_path / "multi.pt"), args)
    assert got["info"]["world"] == world and got["info"]["backend"] == "gloo"
    _close(got, gpu_single["loop"], f"gloo x{world} on one GPU, {parallelism}", rel_bound=1e-2, psnr_bound=50.0)


@pytest.mark.gpu
@pytest.mark.parametrize("world,parallelism", [(2, "sp"), (4, "sp")])
def test_shared_gpu_copy_engine_transport_is_bit_identical_to_allgather(tmp_path, world, parallelism):
    """KVGather mode "ipc" (csrc/ipc.hip, the CU-free K|V transport) between REAL processes: N ranks sharing the one GPU open
    each other's symmetric heap through hipIpc, pull each other's K|V row chunks with hipMemcpyAsync on one stream per peer,
    gated by the flag words in the shared segment (hipStreamWriteValue32 / hipStreamWaitValue32) - the whole protocol of
    the N-GPU run except that the copy is a same-device blit instead of SDMA over xGMI.  The exchange only moves bytes, so the
    final latent must be BIT-IDENTICAL to the same N-rank run on the collective (`allgather`, gloo here) transport."""
    args = ["--backend", "gloo", "--share-gpu"] + GPU_TINY[2:] + ["--scenario", "loop", "--parallelism", parallelism]
    ref = run_ranks(world, str(tmp_path / "allgather.pt"), args + ["--kv-exchange", "allgather"])
    got = run_ranks(world, str(tmp_path / "ipc.pt"), args + ["--kv-exchange", "ipc"])
    assert got["info"]["world"] == world and got["info"]["kv_exchange"] == "ipc" and got["info"]["kv_collectives"] == ref["info"]["kv_collectives"] > 0
    assert torch.equal(got["result"], ref["result"]), "the copy-engine transport moved different bytes than the collective one"


@pytest.mark.gpu
@pytest.mark.parametrize("world,kv_exchange", [(2, "ipc+arrival"), (4, "ipc+arrival"), (4, "allgather+arrival"), (4, "p2p+arrival")])
def test_shared_gpu_arrival_driven_attention(tmp_path, gpu_single, world, kv_exchange):
    """SURVEY §8e "process K/V chunks in arrival order (own shard first) with online-softmax merging" between REAL processes: N ranks
    sharing the one GPU run the `sp` loop with ONE arrival-gated attention launch per layer and branch (csrc/attn7p.hip): this
    rank's own rows are read in place, every (row chunk, peer) piece is gated inside the kernel on the flag its transfer raises - the
    copy-engine transport's per-peer device word (icv_ipc_arrival), or one flag per chunk written by a side stream that waited for the
    collective (icv_flag_write).  Against the single-GPU loop at the N-rank bar (only the softmax merge order and the bf16
    roundings it moves change), and against the SAME ranks on the chunked launches (also not bit-identical: pieces re-associate the
    fp32 sums differently from chunks).  No rank may report a wait that gave up (WanDiT.check_exchange runs at the end of the loop)."""
    args = ["--backend", "gloo", "--share-gpu"] + GPU_TINY[2:] + ["--scenario", "loop", "--parallelism", "sp"]
    got = run_ranks(world, str(tmp_path / "arrival.pt"), args + ["--kv-exchange", kv_exchange], extra_env={"GPU_MAX_HW_QUEUES": "16"})
    ref = run_ranks(world, str(tmp_path / "chunked.pt"), args + ["--kv-exchange", kv_exchange.split("+")[0]], extra_env={"GPU_MAX_HW_QUEUES": "16"})
    # (the arrival-driven schedule cuts its row chunks on the 64-key tile grid: these small shards give it fewer chunks than the chunked arm)
    assert got["info"]["world"] == world and got["info"]["kv_exchange"] == kv_exchange and 0 < got["info"]["kv_collectives"] <= ref["info"]["kv_collectives"]
    _close(got, gpu_single["loop"], f"gloo x{world} on one GPU, sp / {kv_exchange}", rel_bound=1e-2, psnr_bound=50.0)
    _close(got, ref, f"arrival-driven vs chunked launches, x{world} {kv_exchange}", rel_bound=1e-2, psnr_bound=50.0)


DEAD_PEER_WORKER = r"""
import os, sys, time, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
from infinicube_amd.videogen.ops import HipOps
from infinicube_amd.videogen.seqpar import KVGather, ShardPlan
ops = HipOps("cuda:0")
kg = KVGather(ShardPlan.make(world * 256, world, rank), None, "ipc")
kg.reserve(1 << 20, "cuda:0")                       # set-up + self-test: everybody alive
rows = kg.local_rows(256, 512, torch.bfloat16, ops.alloc)
rows.fill_(float(rank + 1))
out = torch.zeros((world * 256, 512), dtype=torch.bfloat16, device="cuda:0")
torch.cuda.synchronize(); dist.barrier()
if rank == world - 1:
    os._exit(0)                                     # the peer DIES here: it never publishes its rows, never pulls anybody's
t0 = time.time()
kg.acquire()
h = kg.start(rows, out)
kg.wait(h)
torch.cuda.synchronize()                            # must RETURN: the pull's wait gives up after ICV_IPC_WAIT_TIMEOUT_MS
t_sync = time.time() - t0
try:
    kg.check()
    print("NO ERROR"); rc = 1
except RuntimeError as e:
    print(f"rank {rank}: sync returned after {t_sync:.2f} s; check(): {e}", flush=True)
    rc = 0 if (f"rank {world - 1}" in str(e) and "gave up waiting" in str(e) and t_sync < 10.0) else 2
t1 = time.time()
kg.acquire()                                        # the NEXT layer's acquire would wait for the dead peer's pull of our rows: bounded as well
torch.cuda.synchronize()
kg.close()                                          # teardown must not hang either
print(f"rank {rank}: second bounded wait + close took {time.time() - t1:.2f} s", flush=True)
if time.time() - t1 > 20.0:
    rc = 3
os._exit(rc)                                        # no collective teardown with a dead peer
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_copy_engine_transport_dead_peer_is_a_bounded_wait_and_a_named_error(world):
    """a peer that crashes mid-run used to leave the survivors' pull streams spinning for ever (the runtime's
    hipStreamWaitValue32 is an unbounded spin kernel) and their teardown blocked.  Now every device-side wait has a deadline: the
    survivors' device synchronise returns, icv_ipc_check names the dead rank, the next acquire is bounded too and close() returns."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ICV_IPC_WAIT_TIMEOUT_MS="800",
                   ICV_IPC_DRAIN_TIMEOUT_MS="1000", PYTHONPATH=os.pathsep.join([ROOT, HERE, os.environ.get("PYTHONPATH", "")]),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="16")
        procs.append(subprocess.Popen([sys.executable, "-c", DEAD_PEER_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL))
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=240)
            outs.append(o.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} (rc {p.returncode}):\n{outs[r][-3000:]}"
    print("\n".join(line for o in outs for line in o.splitlines() if line.startswith("rank ")))


LATE_PEER_WORKER = r"""
import math, os, sys, time, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
from infinicube_amd.videogen.ops import HipOps
from infinicube_amd.videogen.seqpar import KVGather, ShardPlan
ops, H, n, late_ms = HipOps("cuda:0"), 8, 2048, float(os.environ["LATE_MS"])
d = H * 128
kg = KVGather(ShardPlan.make(world * n, world, rank), None, "ipc")
kg.reserve(n * 2 * d * 2 + 4096, "cuda:0")
kg.enable_arrival(ops)
rows = kg.local_rows(n, 2 * d, torch.bfloat16, ops.alloc)
g = torch.Generator(device="cuda:0").manual_seed(100 + rank)
rows.copy_((torch.randn((n, 2 * d), device="cuda:0", generator=g) * 0.5).to(torch.bfloat16))
q = (torch.randn((1024, d), device="cuda:0", generator=g) * 0.5).to(torch.bfloat16)
full = torch.full((world * n, 2 * d), float("nan"), dtype=torch.bfloat16, device="cuda:0")    # nothing has landed yet
err = torch.zeros((1,), dtype=torch.int32, device="cuda:0")
nwg = H * 4
trace = torch.zeros((nwg, world), dtype=torch.int64, device="cuda:0")
out = {}
for rep in range(2):                      # rep 0 warms every first-use path (streams, modules); rep 1 is the one looked at
    full.fill_(float("nan")); trace.zero_()
    torch.cuda.synchronize(); dist.barrier()
    kg.acquire()
    if rank == world - 1:
        time.sleep(late_ms / 1e3)          # the LATE peer: its rows are published late_ms after everybody else's
    h = kg.start(rows, full)
    flags, entries = kg.arrival(h)
    pieces = [(rows[:, :d], rows[:, d:], -1, 0)] + [(full[j * n:(j + 1) * n, :d], full[j * n:(j + 1) * n, d:], idx, val)
                                                    for j, idx, val in sorted(entries, key=lambda e: (e[0] - rank) % world)]
    o = torch.zeros((1024, d), dtype=torch.bfloat16, device="cuda:0")
    ops.attention_pieces(q, pieces, o, H, 1.0 / math.sqrt(128), flags=flags, err=err, timeout_us=10_000_000, trace=trace)
    kg.consumed(h)
    torch.cuda.synchronize()
    out[rep] = o
assert int(err.item()) == 0, hex(int(err.item()) & 0xffffffff)
kg.check()
# the same launch once everything is there (no flags): must be bit-identical
o2 = torch.zeros_like(out[1])
ops.attention_pieces(q, [(k, v, -1, 0) for k, v, _, _ in pieces], o2, H, 1.0 / math.sqrt(128))
torch.cuda.synchronize()
assert torch.isfinite(out[1].float()).all() and torch.equal(out[1], o2), "rows were consumed before they had landed"
t = trace.cpu()
own_to_first_peer = float((t[:, 1] - t[:, 0]).double().median()) / 100.0          # us
last_piece = float((t[:, world - 1] - t[:, 0]).double().min()) / 100.0
print(f"rank {rank}: own piece consumed, first peer piece started after {own_to_first_peer:.0f} us; last piece started after {last_piece:.0f} us", flush=True)
if rank == 0:              # rank 0's LAST piece is the late peer's (pull order rank+1, rank+2, ...)
    assert last_piece >= 0.6 * late_ms * 1e3, f"the late peer's piece was started after {last_piece:.0f} us: before its rows existed?"
    if world > 2:
        assert own_to_first_peer < 0.5 * late_ms * 1e3, "progress on the pieces that WERE there must not wait for the late peer"
kg.close()
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_arrival_driven_attention_makes_progress_while_one_peer_publishes_late(world):
    """VERDICT r5 item 2's delayed-peer test, between REAL processes: N ranks sharing the GPU exchange K|V rows over the copy-engine
    transport and consume them with ONE arrival-gated attention launch; the last rank publishes its rows 5 ms late.  On rank 0 the launch
    must consume its own rows (and, with three ranks, the punctual peer's) within microseconds, start the late peer's piece only after
    ~5 ms, end with the result bit-identical to the same launch over rows that are all there, and report no wait that gave up."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LATE_MS="5",
                   PYTHONPATH=os.pathsep.join([ROOT, HERE, os.environ.get("PYTHONPATH", "")]), HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="16")
        procs.append(subprocess.Popen([sys.executable, "-c", LATE_PEER_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, stdin=subprocess.DEVNULL))
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=300)
            outs.append(o.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed (rc {p.returncode}):\n{outs[r][-3000:]}"
    print("\n".join(line for o in outs for line in o.splitlines() if line.startswith("rank ")))


IPC_ABORT_WORKER = r"""
import os, sys, time, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
from infinicube_amd.videogen.seqpar import _IpcHeap
t0 = time.time()
try:
    _IpcHeap(dist, None, list(range(world)), rank, world, 1 << 20, "cuda:0")
    print("NO ERROR")
    rc = 1
except RuntimeError as e:
    print(f"rank {rank}: error after {time.time() - t0:.1f} s: {e}")
    rc = 0 if "self-test" in str(e) and "injected failure" in str(e) else 2
assert not _IpcHeap._live
dist.destroy_process_group()
sys.exit(rc)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world,bad", [(2, 1), (3, 0)])
def test_copy_engine_transport_one_rank_failing_is_an_error_on_every_rank_not_a_hang(world, bad):
    """One rank fails in the middle of the set-up self-test (test hook) while its peers' pull streams already spin on its
    "rows ready" flag: icv_ipc_abort releases them (the flag words jump past every sequence number), the peers pull undefined bytes,
    the self-test reports it, and EVERY rank raises the same RuntimeError within seconds - what lets the start-up ladder drop the
    transport symmetrically instead of waiting for a watchdog."""
    import time
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PYTHONPATH=os.pathsep.join([ROOT, HERE, os.environ.get("PYTHONPATH", "")]), OMP_NUM_THREADS="2",
                   ICV_TEST_HOOKS="1", ICV_IPC_INJECT=f"selftest:{bad}")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("GPU_MAX_HW_QUEUES", "16")
        procs.append(subprocess.Popen([sys.executable, "-c", IPC_ABORT_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      stdin=subprocess.DEVNULL))
    t0, outs = time.time(), []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=120)
            outs.append(o.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()          # exactly the PIDs started here
                p.wait()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} (rc {p.returncode}):\n{outs[r][-2000:] if r < len(outs) else ''}"
    print(f"{world} ranks, rank {bad} failing inside the self-test: every rank raised after {time.time() - t0:.1f} s (incl. start-up)")


@pytest.mark.gpu
def test_shared_gpu_fp8_mode_cfg_sp_subgroups(tmp_path):
    """The e4m3 wire format under the `cfg+sp` layout: 4 ranks = two branch groups of 2; the abs-max reduction and the blob exchange
    run inside each branch's sub-group (copy-engine transport), the branches swap velocity tokens per step."""
    args = GPU_TINY[2:] + ["--scenario", "loop", "--gemm-dtype", "fp8", "--attn-dtype", "fp8"]
    ref = run_ranks(1, str(tmp_path / "single.pt"), ["--backend", "nccl"] + args)
    got = run_ranks(4, str(tmp_path / "multi.pt"), ["--backend", "gloo", "--share-gpu"] + args + ["--parallelism", "cfg+sp", "--kv-exchange", "ipc"])
    assert got["info"]["mode"] == "cfg+sp" and got["info"]["sp_world"] == 2
    _close(got, ref, "gloo x4 on one GPU, cfg+sp, fp8 mode, e4m3 on the wire via ipc", rel_bound=2e-2, psnr_bound=45.0)


@pytest.mark.gpu
@pytest.mark.parametrize("kv_exchange", ["allgather", "ipc", "ipc+arrival", "allgather+arrival"])
def test_shared_gpu_fp8_mode_e4m3_on_the_wire(tmp_path, kv_exchange):
    """Config #5's kernels on the sharded path between real processes (2 ranks sharing the GPU): every rank quantises its own K|V
    rows once with the group's per-head scales (one max-reduce per layer), the exchange moves e4m3 blobs - over the collective
    and over the copy-engine transport - and icv_attention_fp8_fwd_pieces consumes them in place.  The scales are those of the
    unsharded launch, so the 2-rank result differs from the 1-GPU fp8 result by the softmax merge order only (bar: the bf16
    path's), not by a second quantisation as in rounds 2-4 (bar then: rel-L2 6e-2 / 30 dB).
    "+arrival" (round 6): every chunk's launch starts without a host-side wait and gates on its blobs' arrival flags inside the kernel
    (icv_attention_fp8_fwd_pieces_gated), this rank's own blob read where it was quantised."""
    args = GPU_TINY[2:] + ["--scenario", "loop", "--gemm-dtype", "fp8", "--attn-dtype", "fp8"]
    ref = run_ranks(1, str(tmp_path / "single.pt"), ["--backend", "nccl"] + args)
    got = run_ranks(2, str(tmp_path / "multi.pt"), ["--backend", "gloo", "--share-gpu"] + args + ["--parallelism", "sp", "--kv-exchange", kv_exchange])
    assert got["info"]["kv_exchange"] == kv_exchange
    _close(got, ref, f"gloo x2 on one GPU, fp8 mode, e4m3 on the wire via {kv_exchange}", rel_bound=2e-2, psnr_bound=45.0)


@pytest.mark.gpu
@needs_gpus(2)
@pytest.mark.parametrize("world,parallelism,kv_exchange", [
    (2, "sp", "allgather"), (2, "sp", "native"), (2, "sp", "ipc"), (2, "cfg+sp", "allgather"),
    (4, "cfg+sp", "p2p"), (4, "sp", "native"), (4, "cfg+sp", "ipc"), (8, "auto", "allgather"), (8, "sp", "p2p"), (8, "sp", "ipc")])
def test_rccl_loop_ranks_equal_single_gpu(tmp_path, gpu_single, world, parallelism, kv_exchange):
    """A full CFG loop on N GPUs over RCCL — `sp` and `cfg+sp`, the three K|V transports — against the 1-GPU latent."""
    if _n_gpus() < world:
        pytest.skip(f"needs >= {world} GPUs (this box has {_n_gpus()})")
    got = run_ranks(world, str(tmp_path / "multi.pt"), GPU_TINY + ["--scenario", "loop", "--parallelism", parallelism, "--kv-exchange", kv_exchange])
    assert got["info"]["world"] == world and got["info"]["backend"] == "nccl"
    _close(got, gpu_single["loop"], f"RCCL x{world} {parallelism}/{kv_exchange}", rel_bound=1e-2, psnr_bound=50.0)


BIG_LAYER = ["--backend", "nccl", "--ops", "hip", "--model", "14b", "--frames", 93, "--height", 480, "--width", 832, "--scenario", "layer",
             "--sp-chunks", 4]


@pytest.fixture(scope="module")
def big_layer_single(tmp_path_factory):
    """The 1-GPU run of the 14B block, once for the three transports (the driver's suite has a time limit)."""
    return run_ranks(1, str(tmp_path_factory.mktemp("ranks_big") / "single.pt"), BIG_LAYER, timeout=600)


@pytest.mark.gpu
@needs_gpus(2)
@pytest.mark.parametrize("kv_exchange", ["allgather", "p2p", "native", "ipc"])
def test_rccl_14b_layer_full_S_ranks_equal_single_gpu(tmp_path, big_layer_single, kv_exchange):
    """Config #4's layer: ONE Wan2.1-14B block at S = 37 440 sharded over every GPU of the box, K|V rows over RCCL,
    against the same block on one GPU."""
    n = max(w for w in (2, 4, 8) if w <= _n_gpus())
    big, ref = BIG_LAYER, big_layer_single
    got = run_ranks(n, str(tmp_path / "multi.pt"), big + ["--kv-exchange", kv_exchange], timeout=600)
    assert got["info"]["sp_world"] == n and got["info"]["kv_collectives"] == 4
    _close(got, ref, f"RCCL x{n} 14B block S=37440 {kv_exchange}")


@pytest.mark.gpu
def test_shared_gpu_14b_layer_full_S_over_the_copy_engine_transport(tmp_path, big_layer_single):
    """Config #4's layer at its real size through the copy-engine transport: ONE Wan2.1-14B block at S = 37 440 as two ranks
    (18 720 tokens each, 4 ramped chunks: 383 MB of K|V rows pulled per rank per layer through hipIpc, the 1.15 GB symmetric heap
    carved for kv_loc) sharing the one GPU, against the same block on one rank."""
    args = ["--backend", "gloo", "--share-gpu"] + BIG_LAYER[2:] + ["--kv-exchange", "ipc"]
    got = run_ranks(2, str(tmp_path / "multi.pt"), args, timeout=600)
    assert got["info"]["sp_world"] == 2 and got["info"]["kv_collectives"] == 4 and got["info"]["kv_exchange"] == "ipc"
    _close(got, big_layer_single, "2 ranks on one GPU, 14B block S=37440, copy-engine transport")


@pytest.mark.gpu
@needs_gpus(2)
@pytest.mark.parametrize("kv_exchange", ["allgather", "ipc"])
def test_rccl_fp8_mode_ranks_equal_single_gpu(tmp_path, kv_exchange):
    """Config #5's kernels (e4m3 projections + e4m3 self-attention) on the sharded path with e4m3 K|V on the wire: every rank
    quantises its own rows with the group's scales (= the unsharded launch's), so the N-GPU result differs from the 1-GPU one by the
    softmax merge order only (bar: the bf16 path's; rounds 2-4 re-quantised every gathered chunk: 6e-2 / 30 dB)."""
    n = max(w for w in (2, 4, 8) if w <= _n_gpus())
    args = GPU_TINY + ["--scenario", "loop", "--gemm-dtype", "fp8", "--attn-dtype", "fp8"]
    ref = run_ranks(1, str(tmp_path / "single.pt"), args)
    got = run_ranks(n, str(tmp_path / "multi.pt"), args + ["--parallelism", "sp", "--kv-exchange", kv_exchange])
    _close(got, ref, f"RCCL x{n} fp8 mode, e4m3 on the wire via {kv_exchange}", rel_bound=2e-2, psnr_bound=45.0)


@pytest.mark.gpu
@pytest.mark.parametrize("n,launcher", [(1, "plain"), (2, "self"), (4, "torchrun")])
def test_bench_stdout_is_one_json_line_on_one_gpu(n, launcher):
    """The bench contract on a 1-GPU box: stdout is ONE JSON line and nothing else (native libraries print there too: gloo's
    "[Gloo] Rank ..." lines, RCCL's banner), for the plain N = 1 form and — N ranks sharing the one GPU over gloo, a code-path
    check, never a measurement — for the self-launching and the torch.distributed.run forms, with the layout `auto` builds."""
    tail = ["bench.py", "--gpus", str(n), "--model", "small", "--frames", "9", "--height", "128", "--width", "160", "--steps", "2", "--warmup", "1",
            "--no-cpu-baseline"]
    cmd = [sys.executable] + tail if launcher != "torchrun" else [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port",
        str(_free_port())] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    if n > 1:
        # gloo moving device tensors between processes that share one GPU is slow and erratic (seconds per all-gather): give the
        # watched phases room, this test is about the line, not about speed
        env.update(ICV_BENCH_SHARE_GPU="1", ICV_DIST_BACKEND="gloo", ICV_GUARD_BUDGETS="autotune=400,warmup=400,timed=400")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), f"stdout must be ONE JSON line and nothing else, got {len(lines)} lines: {r.stdout[:400]!r}"
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["unit"] == "denoise steps/s"
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    if n > 1:
        assert d["multi_gpu"]["rccl_ranks"] == n and d["multi_gpu"]["backend"] == "gloo" and "exposed_kv_wait_ms_per_step" in d["multi_gpu"]
        assert ("cfg2 x sp" in d["config"]["parallelism"]) and d["scaling"] == "strong"
        # the run went through the supervisors: which plan ran, nothing failed before it, and (sp groups of >= 2 ranks) the
        # start-up autotune's table of transports x chunk counts with the choice that was timed
        assert d["multi_gpu"]["plan"].startswith("cfg+sp / kv-exchange auto") and d["multi_gpu"]["failed_attempts"] == []
        if n >= 4:
            at = d["multi_gpu"]["autotune"]
            assert {(r["kv_exchange"], r["sp_chunks"]) for r in at["table"]} >= {("allgather", 4), ("p2p", 4), ("allgather", 2), ("allgather+arrival", 4), ("ipc+arrival", 4)}
            # first contact made self-explaining: per candidate the attention under the real exchange next to the same
            # launches from memory, the raw exchange's receive rate, and for the copy-engine transport whether a pull needs compute units
            ok = [r for r in at["table"] if r["ms"]]
            assert all(r.get("attn_under_exchange_ms") and r.get("attn_from_memory_ms") and r.get("recv_gb_per_s_per_rank") for r in ok), ok
            ipc = [r for r in ok if r["kv_exchange"].startswith("ipc")]
            assert ipc and all(r.get("ipc_peer_copy") for r in ipc), "the copy-engine candidates must say whether their pulls need CUs"
            # ranks SHARING one GPU pull with same-device blits - but the hardware scheduler time-slices the processes' queues, so the occupier
            # cannot always hold the device: "blit" or "inconclusive", never "copy engine" (the instrument's own controls run in ONE process:
            # tests/test_kernels_gpu.py::test_copy_path_probe_controls)
            assert all("copy engine" not in r["ipc_peer_copy"] for r in ipc), [r["ipc_peer_copy"] for r in ipc]
            assert d["multi_gpu"]["kv_exchange"] == at["chosen"]["kv_exchange"] and any(r["ms"] for r in at["table"])


@pytest.mark.gpu
@pytest.mark.parametrize("inject,plan_prefix,n_failed", [("0:1:groups:raise", "cfg+sp / kv-exchange allgather", 1),
                                                        ("0:3:autotune:hang,1:0:warmup:raise", "sp / kv-exchange allgather", 2)])
def test_bench_falls_back_and_says_so(inject, plan_prefix, n_failed):
    """First-contact robustness on the real bench.py (4 ranks sharing the one GPU over gloo): a rank that RAISES while the groups
    are built, and a rank that HANGS in the autotune followed by one that raises in the warm-up of the next plan - every rank moves
    to the next plan together, the run still ends with a measured line, and `multi_gpu` names the plan that ran and what failed."""
    cmd = [sys.executable, "bench.py", "--gpus", "4", "--model", "tiny", "--frames", "9", "--height", "128", "--width", "160", "--steps", "1",
           "--warmup", "1", "--no-cpu-baseline"]      # tiny: the last plan is plain `sp` over gloo with device tensors - seconds per all-gather
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(ICV_BENCH_SHARE_GPU="1", ICV_DIST_BACKEND="gloo", ICV_GUARD_INJECT=inject, ICV_TEST_HOOKS="1", ICV_BENCH_DIST_TIMEOUT_S="40",
               ICV_GUARD_BUDGETS="autotune=12,warmup=400,timed=400" if "hang" in inject else "autotune=400,warmup=400,timed=400")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:400]
    d = json.loads(lines[0])
    mg = d["multi_gpu"]
    assert d["value"] > 0 and mg["plan"].startswith(plan_prefix) and len(mg["failed_attempts"]) == n_failed, mg
    first = mg["failed_attempts"][0]
    assert first["plan"].startswith("cfg+sp / kv-exchange auto") and first["phase"].split(":")[0] in ("groups", "autotune") and first["reason"]   # "autotune:<candidate>": one phase per candidate
    if n_failed == 2:
        assert mg["parallelism"] == "sp" and mg["kv_group_ranks"] == 4 and mg["failed_attempts"][1]["phase"] == "warmup"


@pytest.mark.gpu
@needs_gpus(2)
@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_line_for_n_gpus(launcher):
    """`python bench.py --gpus N` (no launcher: it starts its own ranks) and the driver's torch.distributed.run form both
    end in ONE JSON line with the multi-GPU block."""
    n = max(w for w in (2, 4, 8) if w <= _n_gpus())
    tail = ["bench.py", "--gpus", str(n), "--model", "small", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), f"stdout must be ONE JSON line and nothing else, got {len(lines)} lines: {r.stdout[:400]!r}"
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["multi_gpu"]["rccl_ranks"] == n and "exposed_kv_wait_ms_per_step" in d["multi_gpu"]
    assert d["scaling"] == "strong" and d["value"] > 0


def _pool_frames(tmp_path, monkeypatch, n, backend, share, factory="gpu_factory", max_diff=2, max_frac=0.02, max_mean=None, check=None):
    """Frames of the single-GPU generator and of the same generator behind an n-rank worker pool (ICV_WORLD=n).
    ``check(generator, frames)`` runs while the pool is alive."""
    import contextlib
    import io
    import numpy as np
    from safetensors.torch import save_file
    import mgpu_factory as F
    import torch.distributed as dist
    from infinicube.videogen import WanVideoGenerator
    from infinicube_amd.videogen import synthetic as syn
    if dist.is_initialized():          # e.g. the one-rank group tests/test_dit_gpu.py leaves behind in this session:
        dist.destroy_process_group()   # with a foreign group alive ICV_WORLD stands down (multigpu.requested_world)
    path = str(tmp_path / "step-1.safetensors")
    save_file({"buffer_embedder." + k: v for k, v in syn.make_buffer_embedder_state_dict(F.CFG).items()}, path)
    sem, co = syn.make_dummy_buffers(F.GRID)
    if share:
        monkeypatch.setenv("ICV_TEST_SHARE_GPU", "1")

    def run():
        with contextlib.redirect_stdout(io.StringIO()):
            g = WanVideoGenerator(path, device="cuda:0", use_wan_1pt3b=True, pipeline_factory=getattr(F, factory))
            frames = g.generate(sem, co, seed=3)
        return g, np.stack([np.asarray(f) for f in frames])

    _, ref = run()
    monkeypatch.setenv("ICV_WORLD", str(n))
    monkeypatch.setenv("ICV_DIST_BACKEND", backend)
    monkeypatch.setenv("ICV_WORKER_FACTORY", f"mgpu_factory:{factory}")
    monkeypatch.setenv("ICV_WORLD_TIMEOUT_S", "240")
    monkeypatch.setenv("PYTHONPATH", os.pathsep.join([ROOT, HERE, os.environ.get("PYTHONPATH", "")]))
    g = None
    try:
        g, got = run()
        # the caller's process is the pool's client: no process group here, N fresh processes behind it
        assert not dist.is_initialized() and g._pool is not None and g._pool.world == n and g._pool.backend == backend
        assert all(r["GPU_MAX_HW_QUEUES"] == "16" and not r["hip_initialised_at_start"] for r in g._pool.plan_record()["ranks"])
        d = np.abs(ref.astype(np.int16) - got.astype(np.int16))
        assert d.max() <= max_diff and (d > 0).mean() < max_frac, f"{n}-rank frames differ from the single-GPU frames: max {d.max()}, {100 * (d > 0).mean():.2f} % pixels"
        assert max_mean is None or d.mean() <= max_mean, f"mean |diff| {d.mean():.3f} levels"
        if check is not None:
            check(g, got)
    finally:
        if g is not None and g._pool is not None:
            g._pool.close()
    assert not dist.is_initialized()


@pytest.mark.gpu
def test_worker_pool_two_processes_sharing_the_gpu(tmp_path, monkeypatch):
    """ICV_WORLD=2 behind the unchanged single-process caller with the PRODUCT operator set in both processes, on a 1-GPU
    box: the caller's process and one worker share cuda:0 and talk over gloo (request broadcast, velocity swap, latent
    gather) — the worker-pool path end to end on HIP kernels, minus the RCCL transport."""
    _pool_frames(tmp_path, monkeypatch, 2, "gloo", share=True)


@pytest.mark.gpu
def test_worker_pool_client_with_hip_already_initialised_four_ranks_copy_engine_transport(tmp_path, monkeypatch):
    """The reference's caller has long initialised the GPU when it builds the generator
    [R infinicube/inference/guidance_buffer_generation.py:459-460,626 vs :755-766].  Here the parent allocates and launches on
    cuda:0 FIRST, then builds the generator with ICV_WORLD=4 (four ranks sharing the GPU), `sp` layout and the copy-engine
    K|V transport forced: every rank - 0 included - must be a fresh process whose runtime initialised under
    GPU_MAX_HW_QUEUES=16 / HSA_ENABLE_IPC_MODE_LEGACY=0 (the pool's plan record says so), the parent must have joined no
    process group and keep its environment, and the frames must be BIT-IDENTICAL to the same four ranks on the collective
    transport (the exchange only moves bytes) and equal the single-GPU frames to rounding."""
    import numpy as np
    x = torch.randn(512, 512, device="cuda:0")
    y = (x @ x).sum().item()                                     # the caller's own GPU work, before the generator exists
    assert torch.cuda.is_initialized() and y == y
    env_before = {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "NCCL_MAX_NCHANNELS", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    monkeypatch.setenv("ICV_PARALLELISM", "sp")
    monkeypatch.setenv("ICV_WORLD_FALLBACK", "0")                # the forced plan or an error: no silent ladder in this test
    frames = {}

    def check_for(kv):
        def check(g, got):
            rec = g._pool.plan_record()
            assert rec["plan"] == ["sp", kv] and rec["failed_plans"] == []
            assert rec["client"]["hip_initialised"] is True and rec["client"]["joined_process_group"] is False
            assert len(rec["ranks"]) == 4 and all(r["GPU_MAX_HW_QUEUES"] == "16" and r["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
                                                  and r["hip_initialised_at_start"] is False and r["device"] == "cuda:0" for r in rec["ranks"])
            assert {k: os.environ.get(k) for k in env_before} == env_before
            frames[kv] = got
            print(f"pool plan record ({kv}): {json.dumps(rec)}")
        return check

    for kv in ("allgather", "ipc"):
        monkeypatch.setenv("ICV_KV_EXCHANGE", kv)
        # four-way sharded attention on 18-token shards reorders more fp32 sums than the two-rank case: max 1 level on 2.5 % of the bytes measured
        _pool_frames(tmp_path, monkeypatch, 4, "gloo", share=True, check=check_for(kv), max_diff=1, max_frac=0.06)
    assert np.array_equal(frames["ipc"], frames["allgather"]), "copy-engine transport behind the client differs from the collective transport"


@pytest.mark.gpu
def test_worker_pool_sharing_the_gpu_with_the_tiled_vae_dealt_to_the_ranks(tmp_path, monkeypatch):
    """The same with the PRODUCT's tiled Wan-VAE (bf16, NDHWC volumes, every convolution on libicvideo's icv_conv3d_ndhwc): both
    buffer encodes and the decode are shared out over the two processes on the GPU (the non-zero rank joins the decode's
    broadcasts and returns nothing); frames against the single-process generator.  The latents are identical (no K|V sharding
    at two ranks: one CFG branch per rank) and since round 5 no process picks its own convolution kernel (no MIOpen search:
    the same HIP kernel in every process, tests/test_vae_shard.py asserts the dealt tiles bit-identical on the GPU), so the
    bar is EQUALITY of the frames (rounds 2-4: max 24 levels, when MIOpen's per-process kernel choice moved the tiles)."""
    _pool_frames(tmp_path, monkeypatch, 2, "gloo", share=True, factory="gpu_real_vae_factory", max_diff=0, max_frac=1e-9, max_mean=0.0)


@pytest.mark.gpu
@needs_gpus(2)
def test_worker_pool_on_rccl(tmp_path, monkeypatch):
    """ICV_WORLD=N behind the unchanged single-process caller, on real GPUs: this process is rank 0 on cuda:0, the
    workers take cuda:1.., the process group is RCCL; frames against the single-GPU generator."""
    _pool_frames(tmp_path, monkeypatch, max(w for w in (2, 4, 8) if w <= _n_gpus()), "nccl", share=False)
