"""World-size-2 rehearsal of the data-parallel sampling plumbing on CPU (gloo): batch sharding and the
final gather of latents in rank order, for equal and unequal shards."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusion_model_amd.parallel import gather_latents, init_distributed, shard_batch, shard_bounds


def test_shard_bounds_cover_the_batch_exactly():
    for total in (0, 1, 7, 16, 128):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(128, 8, 3) == (48, 64)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    full = torch.arange(total * 4 * 2 * 2, dtype=torch.float32).reshape(total, 4, 2, 2)
    mine = shard_batch(full)                       # this rank's images
    latents = mine * 2.0 + 1.0                     # stands in for the sampled latents of the shard
    gathered = gather_latents(latents)
    ok = torch.equal(gathered, full * 2.0 + 1.0)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        torch.save({"ok": bool(flag.item()), "shape": tuple(gathered.shape)}, out)
    dist.barrier()
    dist.destroy_process_group()


def _run(total, tmp_path):
    out = str(tmp_path / f"r{total}.pt")
    mp.spawn(_worker, args=(2, _free_port(), total, out), nprocs=2, join=True)
    res = torch.load(out, weights_only=True)
    assert res["ok"] and res["shape"] == (total, 4, 2, 2)


def test_gather_equal_shards_gloo(tmp_path):
    _run(8, tmp_path)


def test_gather_unequal_shards_gloo(tmp_path):
    _run(5, tmp_path)
