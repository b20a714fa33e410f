"""N>1 path on CPU: two gloo ranks shard a particle batch, exchange the 8-float partial statistics and must
reproduce the single-process estimators of BaseOCLoss.compute_results (oracle formulas)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sde_oracle as orc
from sde_sampler_lrds_amd import parallel


def local_stats(rnd):
    """What sdeng_logz writes for one shard (include/sdeng.h), in torch (test helper)."""
    v = -rnd.double().view(-1)
    mx = v.max()
    e = torch.exp(v - mx)
    var = rnd.double().var() if v.numel() > 1 else torch.tensor(0.0)
    return torch.tensor([v.mean(), mx + e.sum().log() - torch.log(torch.tensor(float(v.numel()))), var,
                         e.sum() ** 2 / (e ** 2).sum() / v.numel(), mx, e.sum(), (e ** 2).sum(), v.sum()], dtype=torch.float32)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    rnd_all = 3.0 * torch.randn(total, 1, generator=g) + 5.0  # every rank can rebuild the global batch
    lo, hi = parallel.shard_bounds(total, world, rank)
    payload = torch.cat([local_stats(rnd_all[lo:hi]), torch.tensor([float(hi - lo)])])
    gathered = torch.empty(world * 9)
    dist.all_gather_into_tensor(gathered, payload)
    res = parallel.combine_stats(gathered.view(world, 9)[:, :8], gathered.view(world, 9)[:, 8])
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [1000, 1001])
def test_two_rank_sharded_estimators_match_single_process(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    rnd_all = 3.0 * torch.randn(total, 1, generator=g) + 5.0
    ref = orc.compute_results(rnd_all)
    assert abs(res["log_norm_const_is"] - ref["log_norm_const_is"]) < 1e-5
    assert abs(res["elbo"] - ref["elbo"]) < 1e-5
    assert abs(res["lv_loss"] - ref["lv_loss"]) < 1e-4
    assert abs(res["ess"] - ref["ess"]) < 1e-6
    assert res["n"] == total


def test_shard_bounds_cover_batch():
    for total in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
