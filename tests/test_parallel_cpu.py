"""N>1 path on CPU: two gloo ranks shard a particle batch, exchange the 8-float partial statistics through the product's own
gather / combine code (parallel.py) and must reproduce the single-process estimators and importance weights of
BaseOCLoss.compute_results (oracle formulas).  The per-shard reduction is parallel.stats_reference -- the torch restatement the HIP
kernel sdeng_logz is checked against index by index on the GPU (tests/test_gpu_units.py::test_logz_layout_contract)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sde_oracle as orc
from sde_sampler_lrds_amd import parallel


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
    # the product's own gather / combine code (parallel.global_weights -> global_results_async -> combine_stats); only the per-shard
    # reduction is swapped for its torch restatement (the HIP kernel is checked against that same function on the GPU)
    w, res = parallel.global_weights(rnd_all[lo:hi], dist, stats_fn=parallel.stats_reference)
    wsum = w.double().sum().view(1)
    dist.all_reduce(wsum)
    res["weights_sum_over_ranks"] = float(wsum)
    res["w_first"] = float(w[0]) if rank == 0 else None
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
    # Results.weights normalised over ALL ranks (losses/oc.py:150-161): the shards' weights sum to one, entries match the softmax
    assert abs(res["weights_sum_over_ranks"] - 1.0) < 1e-6
    assert abs(res["w_first"] - float(torch.softmax(-rnd_all.double(), 0)[0])) < 1e-9


def test_shard_bounds_cover_batch():
    for total in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_stats_layout_constants_match_header():
    """include/sdeng.h documents stats[0..7]; parallel.py names the indices once for every consumer."""
    import os
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sdeng.h")).read()
    doc = dict(re.findall(r"stats\[(\d)\] = ([^\n]*?)(?=\s{2,}|\n)", header))
    assert "elbo" in doc[str(parallel.ELBO)] and "logsumexp" in doc[str(parallel.LOGZ)] and "var(" in doc[str(parallel.VAR)]
    assert "max(-rnd)" in doc[str(parallel.MAX)] and "sum exp(-rnd - max)" in doc[str(parallel.SUM_EXP)]
    assert "sum exp(2" in doc[str(parallel.SUM_EXP2)] and "sum(-rnd)" in doc[str(parallel.SUM)]
    r = torch.tensor([[0.5], [1.5], [-0.25]])
    s = parallel.stats_reference(r)
    assert abs(float(s[parallel.SUM]) + float(r.sum())) < 1e-6 and abs(float(s[parallel.MAX]) - 0.25) < 1e-7
    one = parallel.combine_stats(s.view(1, -1), torch.tensor([3.0]))
    assert abs(one["log_norm_const_is"] - float(s[parallel.LOGZ])) < 1e-6 and abs(one["ess"] - float(s[parallel.ESS])) < 1e-6
