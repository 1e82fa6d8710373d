"""2-rank gloo rehearsal (CPU) of bench.py's two scaling modes: the sharding, the per-rank workload and the
halo exchange are the code the GPU bench runs; only the aggregation itself is the CPU oracle here (the HIP
kernel needs a GPU -- tests may use the oracle as the checker, the bench never does)."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import conftest  # noqa: F401
import bench
from hierarchicalgnn_amd import synth
from oracle import hgnn_oracle as O

SIZES = (3000, 20000)
L = 8


def _features(n_hits, n_rows):
    g = torch.Generator().manual_seed(11)
    return torch.randn(n_hits, L, generator=g), torch.randn(n_rows, L, generator=g)


def _worker(rank, world, path, scaling, mode, q):
    dist.init_process_group("gloo", init_method=f"file://{path}", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        wl = bench.Workload(scaling, None, world, rank, "cpu", mode, sizes=SIZES)
        nodes, edges = _features(wl.n_hits, 2 * wl.n_edges)
        s = wl.shard
        out = O.scatter_add(edges[s.edge_global], wl.graph[1], 0, wl.n_local)      # the step's K1, on my rows
        halo_rows = wl.halo.exchange(nodes[s.owned_global])                        # the step's exchange
        stats = wl.gather_stats()
        q.put((rank, wl.n_hits, wl.n_edges, wl.M, wl.n_local, s.owned_global.numpy(), out.numpy(),
               s.halo_global.numpy(), halo_rows.numpy(), stats.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scaling,mode", [("strong", "all_gather"), ("weak", "all_to_all")])
def test_bench_workload_two_ranks(scaling, mode):
    world = 2
    fd, path = tempfile.mkstemp(prefix="hgnn_bench_gloo_")
    os.close(fd)
    os.unlink(path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, path, scaling, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if os.path.exists(path):
        os.unlink(path)
    n_hits = SIZES[0] * (world if scaling == "weak" else 1)
    n_edges = SIZES[1] * (world if scaling == "weak" else 1)
    _, ei = synth.trackml_event(n_hits, n_edges, seed=1234)
    graph = synth.directed(ei)
    nodes, edges = _features(n_hits, 2 * n_edges)
    ref = O.scatter_add(edges, graph[1], 0, n_hits)
    seen = torch.zeros(n_hits, dtype=torch.long)
    for rank, nh, ne, M, n_local, owned, out, halo_global, halo_rows, stats in res:
        assert (nh, ne) == (n_hits, n_edges)                  # strong: the SAME event on every rank count
        owned = torch.from_numpy(owned)
        seen[owned] += 1
        assert torch.allclose(torch.from_numpy(out), ref[owned], rtol=1e-5, atol=1e-5)
        assert torch.equal(torch.from_numpy(halo_rows), nodes[torch.from_numpy(halo_global)])
        assert stats.shape == (world, 4) and int(stats[rank, 0]) == M and int(stats[rank, 1]) == n_local
        assert int(stats[:, 0].sum()) == 2 * n_edges          # every directed row aggregated exactly once
        assert int(stats[:, 2].sum()) == int(stats[:, 3].sum()) or mode == "all_gather"
        assert float(stats[:, 0].max()) / float(stats[:, 0].mean()) < 1.05   # balanced on rows
    assert int(seen.min()) == 1 and int(seen.max()) == 1


def test_bench_event_sizes_and_n1_equivalence():
    """N = 1: weak scaling and strong scaling on the headline event are the same workload (the BENCH line);
    strong scaling defaults to the fixed full-pileup event of BASELINE config 5 at every N"""
    assert bench.event_size("weak", None, 1) == (120_000, 1_000_000, "headline")
    assert bench.event_size("strong", "headline", 1) == (120_000, 1_000_000, "headline")
    assert bench.event_size("weak", None, 8) == (960_000, 8_000_000, "headline")
    for p in (1, 2, 4, 8):
        assert bench.event_size("strong", None, p) == (480_000, 4_000_000, "full_pileup")
    a = bench.Workload("weak", None, 1, 0, "cpu", sizes=SIZES)
    b = bench.Workload("strong", None, 1, 0, "cpu", sizes=SIZES)
    assert torch.equal(a.graph, b.graph) and a.n_local == b.n_local == SIZES[0] and a.M == 2 * SIZES[1]
    assert bench.algorithmic_bytes(2_000_000, 120_000, 256) == 2_178_880_000
    args = bench.parse([])
    assert (args.gpus, args.scaling, args.event) == (1, "weak", None)
