"""A stand-in for bench.py's rank process in the launch-guard tests: the same distributed skeleton (init with a timeout,
sub-groups as the plan says, one collective per group, a result from rank 0), no model.  Failures are injected by
launch_guard.PhaseReporter (ICV_GUARD_INJECT) or by ICV_TEST_FAIL_NEW_GROUP=<attempt>:<rank> (a raising new_group)."""
import datetime
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from infinicube_amd.videogen import launch_guard as guard
from infinicube_amd.videogen.seqpar import ParallelLayout


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    plan = json.loads(os.environ["ICV_BENCH_PLAN"])
    phase = guard.PhaseReporter(rank)
    phase("init")
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=20))
    phase("groups")
    bad = os.environ.get("ICV_TEST_FAIL_NEW_GROUP")
    if bad and plan["parallelism"] == "cfg+sp" and bad == f"{os.environ['ICV_GUARD_ATTEMPT']}:{rank}":
        real = dist.new_group

        def failing_new_group(*a, **k):
            raise RuntimeError("injected: new_group failed (NCCL error: unhandled system error)")
        dist.new_group = failing_new_group
        try:
            ParallelLayout.make(world, rank, plan["parallelism"], use_cfg=True)
        finally:
            dist.new_group = real
    layout = ParallelLayout.make(world, rank, plan["parallelism"], use_cfg=True)
    for g in (layout.sp_group, layout.pair_group):
        if g is not None:
            t = torch.ones(1)
            dist.all_reduce(t, group=g)
    phase("timed")
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    phase("report")
    if rank == 0:
        guard.write_result({"value": 1.0, "sum_of_ranks": float(t.item()), "multi_gpu": {"parallelism": layout.mode, "kv_exchange": plan["kv_exchange"]}})
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
