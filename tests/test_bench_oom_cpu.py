"""CPU: bench.py's PPO out-of-memory fallback (bench.run_ppo_block).  The retry must happen outside the except block
(the failed run's frames are released first), and under data parallelism every rank must end on the SAME rollouts-per-pass
group -- a rank that fitted retries with the smaller group as well (world_size 2 over gloo)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_process_retries_once_with_half_the_group_and_frees_the_failed_run():
    import bench
    calls, alive = [], []

    class Big:
        def __del__(self):
            alive.append("freed")

    def run(a, b, group=None, flag=None):
        calls.append((a, b, group, flag, list(alive)))
        hog = Big()                                      # a local of the failing frame (the models, in the real run)
        if len(calls) == 1:
            raise torch.OutOfMemoryError("stub")
        del hog
        return {"ok": group}

    logs = []
    res, group = bench.run_ppo_block(run, 16, 1, "cpu", logs.append, 64, 1024, flag="x")
    assert res == {"ok": 8} and group == 8
    assert [c[:4] for c in calls] == [(64, 1024, 16, "x"), (64, 1024, 8, "x")]
    assert calls[1][4] == ["freed"]                      # the failed run's locals were gone before the retry started
    assert len(logs) == 1 and "16" in logs[0] and "8" in logs[0]


def test_gives_up_below_one_rollout_per_pass():
    import bench

    def run(group=None):
        raise torch.OutOfMemoryError("stub")

    with pytest.raises(RuntimeError, match="one rollout"):
        bench.run_ppo_block(run, 2, 1, "cpu", lambda m: None)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    calls = []

    def run(group=None):
        calls.append(group)
        if rank == 1 and len(calls) == 1:                # only ONE rank runs out of memory
            raise torch.OutOfMemoryError("stub")
        return {"group": group}

    res, group = bench.run_ppo_block(run, 16, world, "cpu", lambda m: None)
    torch.save({"calls": calls, "group": group, "res": res}, out % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_ranks_agree_on_the_smaller_group(tmp_path):
    port = 29500 + os.getpid() % 2000
    out = str(tmp_path / "r%d.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    assert r0["calls"] == [16, 8] and r1["calls"] == [16, 8]
    assert r0["group"] == r1["group"] == 8 and r0["res"] == r1["res"] == {"group": 8}
