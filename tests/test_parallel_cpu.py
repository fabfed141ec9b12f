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
    gathered = gather_latents(latents, total=total)
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


def test_bench_starts_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` outside a launcher becomes the launcher: two fresh rank processes (gloo here, a
    stub workload on the CPU), one JSON line from rank 0 saying n_gpus = 2, exit code 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["DM_DIST_BACKEND"] = "gloo"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=240, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["gather_ok"] is True
    assert out["config"]["parallelism"] == "dp2"


def test_bench_rank_failure_is_reported():
    """A rank that exits non-zero makes the launcher exit non-zero, whatever rank 0 printed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DM_DIST_BACKEND="gloo", DM_BENCH_STUB_FAIL_RANK="1")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub", "--steps", "1",
                          "--warmup", "0"], capture_output=True, text=True, timeout=240, env=env)
    assert res.returncode == 3, (res.returncode, res.stderr[-1000:])
