"""The N > 1 path on CPU (gloo, world_size 2): static image sharding and the start-up weight-buffer broadcast
(on the GPU box the same code runs over RCCL / xGMI with the packed weight arena as the buffer)."""
import os

import pytest
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pnpinversion_amd.distributed import broadcast_buffer, shard_items


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    items = ["img_%03d" % i for i in range(11)]
    mine = shard_items(items)                    # reads RANK / WORLD_SIZE like the sweep driver
    buf = torch.arange(3000, dtype=torch.int64).to(torch.uint8) if rank == 0 else torch.zeros(3000, dtype=torch.uint8)
    broadcast_buffer(buf, src=0, chunk_bytes=1024)   # several pieces
    # a max-over-ranks timing reduction like bench.py's
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, mine, int(buf.to(torch.int64).sum()), float(t)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_broadcast():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    items = ["img_%03d" % i for i in range(11)]
    assert sorted(res[0][1] + res[1][1]) == items and not set(res[0][1]) & set(res[1][1])
    assert res[0][1] == items[0::2] and res[1][1] == items[1::2]
    expect = int(torch.arange(3000, dtype=torch.int64).to(torch.uint8).to(torch.int64).sum())
    assert res[0][2] == expect and res[1][2] == expect
    assert res[0][3] == 2.0 and res[1][3] == 2.0


def test_single_rank_is_identity():
    assert shard_items(list(range(5)), rank=0, world=1) == list(range(5))


@pytest.mark.parametrize("n,steps,warmup", [(2, 3, 1), (8, 20, 5)])
def test_bench_self_spawns_its_ranks(n, steps, warmup):
    """`python bench.py --gpus N ...` (N = 2, and the driver's 8-GPU command line `--gpus 8 --steps 20 --warmup 5`) without a launcher starts its own two ranks under torch.distributed.run (the driver's N > 1
    command form is a plain python call): rendezvous on 127.0.0.1, barrier, max-over-ranks, ONE JSON line from rank 0.  --dry-run
    keeps it on the CPU (gloo) and leaves `value` null -- only the process plumbing is under test here."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup), "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == n and out["steps"] == steps and out["warmup"] == warmup and out["value"] is None
    assert out["max_rank_seconds"] >= 0.01 * n      # the last rank sleeps longest: the line carries the MAX over ranks
