"""The multi-GPU code path on ONE MI355X with a real RCCL communicator (world_size 1): the
collectives, streams and HIP pack/unpack kernels of partition.py run exactly as they do per rank
at N > 1; the result must equal the plain single-GPU model.  (Multi-rank correctness is covered by
the gloo tests; the 8-GPU run is the driver's.)"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_world1():
    import tempfile
    import torch.distributed as dist
    fd, path = tempfile.mkstemp(prefix="hgnn_rccl_")   # FileStore rendezvous: no fixed TCP port to collide on
    os.close(fd)
    os.unlink(path)
    dist.init_process_group("nccl", init_method=f"file://{path}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()
    try:
        os.unlink(path)
    except OSError:
        pass


@pytest.mark.parametrize("mode", ["all_gather", "all_to_all"])
def test_distributed_ec_forward_on_rccl_world1(rccl_world1, mode):
    from hierarchicalgnn_amd import partition, synth
    from hierarchicalgnn_amd.models import EC_InteractionGNN
    torch.manual_seed(0)
    hp = dict(spatial_channels=3, latent=32, hidden=64, n_interaction_graph_iters=2, nb_node_layer=3,
              nb_edge_layer=2, output_layers=3, hidden_output_activation="GELU", hidden_activation="GELU",
              layernorm=True, share_weight=False)
    model = EC_InteractionGNN(hp).cuda().eval()
    x, ei = synth.trackml_event(3000, 20000, seed=5)
    shard = partition.partition_event(x, ei, 1, 0)
    halo = partition.HaloExchange(shard, "cuda", mode=mode)
    pairs = partition.edge_pair_exchange(x, ei, 1, 0, shard)
    with torch.no_grad():
        ref = model(x.cuda(), ei.cuda())
        scores, ids = partition.distributed_ec_forward_model(model, shard, halo, pairs, x[shard.owned_global].cuda())
    assert scores.shape == (ei.shape[1],)
    out = torch.empty_like(ref)
    out[ids.cuda()] = scores
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5)


def test_bench_step_shape_on_rccl_world1(rccl_world1):
    """what bench.py does per step at N > 1: local K1 + halo exchange on a side stream"""
    import hierarchicalgnn_amd as H
    from hierarchicalgnn_amd import partition, synth
    x, ei = synth.trackml_event(5000, 40000, seed=6)
    shard = partition.partition_event(x, ei, 1, 0)
    halo = partition.HaloExchange(shard, "cuda", mode="all_gather")
    graph = shard.local_graph.cuda()
    edges = torch.randn(graph.shape[1], 64, device="cuda")
    nodes = torch.randn(shard.n_owned, 64, device="cuda")
    side = torch.cuda.Stream()
    for _ in range(3):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            h = halo.exchange(nodes)
        out = H.scatter_add(edges, graph[1], dim=0, dim_size=shard.n_owned)
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert h.shape == (shard.n_halo, 64) and out.shape == (shard.n_owned, 64)


def test_distributed_bc_forward_on_rccl_world1(rccl_world1):
    """BASELINE config 5's model (BC-HGNN-GMM on shards) with a real RCCL communicator of world size 1: the
    all-gathered embeddings, the replicated hierarchy decision, the all-reduced pooling sums and weight mean run
    as they do per rank at N > 1; scores equal the plain single-GPU model's on the same bipartite edges"""
    from hierarchicalgnn_amd import partition
    from hierarchicalgnn_amd.models import BC_MessagePassing
    from conftest import load_golden
    z = load_golden("bc_hgnn_L32.npz")
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
    model = BC_MessagePassing(hp)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}, strict=True)
    model = model.cuda().eval()
    x = torch.from_numpy(z["x"])
    ei = torch.from_numpy(z["edge_index"])
    with torch.no_grad():
        bg_ref, s_ref, emb_ref = model(x.cuda(), ei.cuda())
    shard = partition.partition_event(x, ei, 1, 0)
    halo = partition.HaloExchange(shard, "cuda", mode="all_to_all")
    owned = partition.all_owned_lists(x, ei, 1)
    assert torch.equal(owned[0], shard.owned_global)
    directed = torch.cat([ei, ei.flip(0)], dim=1).cuda()
    with torch.no_grad():
        bg, s, emb = partition.distributed_bc_forward(partition.bc_pieces_from_model(model), shard, halo,
                                                      x[shard.owned_global].cuda(), owned, directed)
    key = lambda g: (g[0] * 100000 + g[1]).cpu()
    o_ref, o = torch.argsort(key(bg_ref)), torch.argsort(key(bg))
    assert torch.equal(key(bg_ref)[o_ref], key(bg)[o])                   # the same bipartite edges
    assert torch.allclose(s.cpu()[o], s_ref.cpu()[o_ref], rtol=1e-4, atol=1e-5)
    assert torch.allclose(emb.cpu(), emb_ref.cpu()[shard.owned_global], rtol=1e-4, atol=1e-5)


def test_partition_event_on_the_device_equals_the_cpu_partition():
    """partition_event runs on the GPU with device-wide sort / scan / unique / compaction (no per-peer host loop, one
    host read): bit-identical shards to the CPU run of the same code (exact integer balance arithmetic), at the
    full-pileup size of BASELINE config 5 (480k hits, 4M edges -> 8M directed rows), in milliseconds"""
    import time
    from hierarchicalgnn_amd import partition, synth
    x, ei = synth.trackml_event(480_000, 4_000_000, seed=1234)
    world, rank = 8, 3
    cpu = partition.partition_event(x, ei, world, rank)
    xd, eid = x.cuda(), ei.cuda()
    partition.partition_event(xd, eid, world, rank)                     # warm-up (allocator, code objects)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gpu = partition.partition_event(xd, eid, world, rank)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for f in ("owned_global", "halo_global", "local_graph", "edge_global", "send_index"):
        assert torch.equal(getattr(gpu, f).cpu(), getattr(cpu, f)), f
    assert (gpu.n_owned, gpu.n_halo, gpu.n_interior, gpu.send_splits, gpu.recv_splits) == \
        (cpu.n_owned, cpu.n_halo, cpu.n_interior, cpu.send_splits, cpu.recv_splits)
    assert gpu.local_graph.is_cuda and 0 < gpu.n_interior < gpu.local_graph.shape[1]
    print(f"partition_event 480k hits / 8M directed edges, rank {rank} of {world}: GPU {dt * 1e3:.1f} ms, "
          f"CPU {cpu.partition_s * 1e3:.0f} ms")
    assert dt < 0.05                                                    # (10 ms is the target; 50 ms the hard bar)


class _ReplayHalo:
    """stands in for the RCCL exchange of a 2-rank run on ONE GPU: returns the halo rows taken from the
    single-process result (what the peer would have sent), on whatever stream it is called on"""

    def __init__(self, shard, table):
        self.shard, self.table, self.device = shard, table, table.device
        self._side = None

    def side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        return self._side

    def exchange(self, nodes_owned):
        return self.table[self.shard.halo_global.to(self.device)].clone()

    def extend(self, nodes_owned):
        return torch.cat([nodes_owned, self.exchange(nodes_owned)], dim=0)


@pytest.mark.parametrize("grad", [False, True])
def test_overlapped_interior_boundary_edge_update_equals_the_plain_cell(grad):
    """distributed_cell_forward's split schedule on the HIP cell: interior edges (owned source) on the current stream
    while the exchange runs on a side stream, then the cut edges; without autograd both write into one table
    (edge_update(out=)), with autograd they are concatenated.  Each of two shards equals the single-GPU cell."""
    import hierarchicalgnn_amd as H
    from hierarchicalgnn_amd import partition, synth
    torch.manual_seed(1)
    L = 64
    hp = dict(latent=L, hidden=2 * L, nb_edge_layer=2, nb_node_layer=3, layernorm=True, hidden_activation="GELU")
    cell = H.InteractionGNNCell(hp).cuda()
    x, ei = synth.trackml_event(4000, 30000, seed=9)
    graph = synth.directed(ei).cuda()
    nodes = torch.randn(4000, L, device="cuda")
    edges = torch.randn(graph.shape[1], L, device="cuda")
    with torch.no_grad():
        ref_n, ref_e = cell(nodes, edges, graph)
    for rank in range(2):
        shard = partition.partition_event(x.cuda(), ei.cuda(), 2, rank)
        assert 0 < shard.n_interior < shard.local_graph.shape[1] and shard.n_halo > 0
        halo = _ReplayHalo(shard, ref_n)
        n_loc = nodes[shard.owned_global].clone().requires_grad_(grad)
        e_loc = edges[shard.edge_global].clone().requires_grad_(grad)
        ctx = torch.enable_grad() if grad else torch.no_grad()
        with ctx:
            out_n, out_e = partition.distributed_cell_forward(cell, halo, n_loc, e_loc, shard.local_graph)
            plain_n, plain_e = partition.distributed_cell_forward(cell, halo, n_loc, e_loc, shard.local_graph,
                                                                  overlap=False)
        torch.cuda.synchronize()
        assert torch.allclose(out_n, ref_n[shard.owned_global], rtol=1e-4, atol=1e-5)
        assert torch.allclose(out_e, ref_e[shard.edge_global], rtol=1e-4, atol=1e-5)
        assert torch.allclose(out_e, plain_e, rtol=1e-5, atol=1e-6)
        if grad:
            out_e.sum().backward()
            assert n_loc.grad is not None and bool(torch.isfinite(n_loc.grad).all())


def test_distributed_bc_forward_training_mode_synchronised_batch_norm_on_rccl_world1(rccl_world1):
    """config 5 TRAINS on shards: in train() mode the bipartite attention weights' BatchNorm uses synchronised batch
    statistics (all-reduced sum / sum of squares / count) -- on a world of one rank that must reproduce the plain
    model's training-mode forward, and the backward must reach every parameter the plain model's reaches"""
    import copy
    from hierarchicalgnn_amd import partition
    from hierarchicalgnn_amd.models import BC_MessagePassing
    from conftest import load_golden
    z = load_golden("bc_hgnn_L32.npz")
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
    model = BC_MessagePassing(hp)
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}, strict=True)
    model = model.cuda().train()
    twin = copy.deepcopy(model)
    x = torch.from_numpy(z["x"])
    ei = torch.from_numpy(z["edge_index"])
    bg_ref, s_ref, emb_ref = model(x.cuda(), ei.cuda())
    (s_ref.sum() + (emb_ref * emb_ref.roll(1, 0)).sum()).backward()
    shard = partition.partition_event(x.cuda(), ei.cuda(), 1, 0)
    halo = partition.HaloExchange(shard, "cuda", mode="all_to_all")
    owned = partition.all_owned_lists(x, ei, 1)
    directed = torch.cat([ei, ei.flip(0)], dim=1).cuda()
    bg, s, emb = partition.distributed_bc_forward(partition.bc_pieces_from_model(twin), shard, halo,
                                                  x[shard.owned_global.cpu()].cuda(), owned, directed)
    (s.sum() + (emb * emb.roll(1, 0)).sum()).backward()
    key = lambda g: (g[0] * 100000 + g[1]).cpu()
    o_ref, o = torch.argsort(key(bg_ref)), torch.argsort(key(bg))
    assert torch.equal(key(bg_ref)[o_ref], key(bg)[o])
    assert torch.allclose(s.detach().cpu()[o], s_ref.detach().cpu()[o_ref], rtol=1e-4, atol=1e-5)
    bn_a = model.hgnn_block.bipartite_graph_construction.weight_normalization
    bn_b = twin.hgnn_block.bipartite_graph_construction.weight_normalization
    assert torch.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-4, atol=1e-6)
    assert torch.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-4, atol=1e-6)
    ga = {k: p.grad for k, p in model.named_parameters()}
    gb = {k: p.grad for k, p in twin.named_parameters()}
    assert {k for k, g in ga.items() if g is not None} == {k for k, g in gb.items() if g is not None}


def test_config5_full_pileup_shard_of_eight_equals_the_unpartitioned_cell():
    """BASELINE config 5's event (480k hits, 4M edges -> 8M directed rows) cut into EIGHT phi-wedges on the device; the
    HIP cell runs rank 3's shard with the overlapped interior / boundary schedule (its halo replayed from the
    unpartitioned result, as the peers would send it) and reproduces the unpartitioned cell on its hits and edges"""
    import hierarchicalgnn_amd as H
    from hierarchicalgnn_amd import partition, synth
    torch.manual_seed(2)
    L = 128
    hp = dict(latent=L, hidden=2 * L, nb_edge_layer=2, nb_node_layer=3, layernorm=True, hidden_activation="GELU")
    cell = H.InteractionGNNCell(hp).cuda().eval()
    x, ei = synth.trackml_event(480_000, 4_000_000, seed=1234)
    x, ei = x.cuda(), ei.cuda()
    graph = torch.cat([ei, ei.flip(0)], dim=1)
    gen = torch.Generator("cuda").manual_seed(4)
    nodes = torch.randn(480_000, L, device="cuda", generator=gen)
    edges = torch.randn(graph.shape[1], L, device="cuda", generator=gen)
    with torch.no_grad():
        ref_n, ref_e = cell(nodes, edges, graph)
        shard = partition.partition_event(x, ei, 8, 3)
        assert 0.9 < shard.local_graph.shape[1] / (graph.shape[1] / 8) < 1.1          # balanced on rows
        assert 0 < shard.n_halo < 0.05 * shard.n_owned                                 # phi-wedges: a thin halo
        halo = _ReplayHalo(shard, ref_n)
        out_n, out_e = partition.distributed_cell_forward(cell, halo, nodes[shard.owned_global],
                                                          edges[shard.edge_global], shard.local_graph)
    torch.cuda.synchronize()
    assert float((out_n - ref_n[shard.owned_global]).abs().max()) <= 1e-4 * float(ref_n.abs().max())
    assert float((out_e - ref_e[shard.edge_global]).abs().max()) <= 1e-4 * float(ref_e.abs().max())


def test_config5_partitioned_bc_forward_at_full_pileup_size_on_rccl_world1(rccl_world1):
    """the partitioned BC-HGNN-GMM forward (config 5's model path: all-gathered embeddings, replicated hierarchy, all-reduced
    pooling sums, halo exchange + overlapped edge updates, RCCL collectives) on the FULL-PILEUP event with a real RCCL
    communicator of one rank: scores equal the plain model's on the same bipartite edges.  (The hierarchy decision is a
    fixed phi-z binning on both sides: random-init embeddings make the GMM cut degenerate.)"""
    from hierarchicalgnn_amd import partition, synth
    from hierarchicalgnn_amd.models import BC_MessagePassing
    hp = dict(spatial_channels=3, latent=64, hidden=128, emb_dim=8, n_interaction_graph_iters=2,
              n_hierarchical_graph_iters=2, nb_node_layer=3, nb_edge_layer=2, output_layers=3,
              hidden_output_activation="Tanh", hidden_activation="GELU", layernorm=True, share_weight=False,
              bipartitegraph_sparsity=5, supergraph_sparsity=10, min_cluster_size=3, cluster_granularity=5)
    torch.manual_seed(3)
    model = BC_MessagePassing(hp).cuda().eval()
    model.hgnn_block.super_graph_construction.knn_radius.fill_(2.0)
    model.hgnn_block.bipartite_graph_construction.knn_radius.fill_(2.0)
    x, ei = synth.trackml_event(480_000, 4_000_000, seed=1234)
    xd, eid = x.cuda(), ei.cuda()
    cl = ((xd[:, 1] + 1) * 50).long().clamp(0, 99) * 100 + ((xd[:, 2] + 1) * 50).long().clamp(0, 99)
    _, clusters = torch.unique(cl, return_inverse=True)
    n_cl = int(clusters.max()) + 1
    with torch.no_grad():
        directed, emb, nodes, edges, _ = model.embed(xd, eid)
        means, bg, bw, sg, sw, _ = model.hgnn_block.hierarchy_from_clusters(emb, clusters, n_cl)
        n_out, sn_out, _, _ = model.hgnn_block(nodes, edges, directed, means, bg, bw, sg, sw)
        s_ref = model.score(n_out, sn_out, bg)
        shard = partition.partition_event(xd, eid, 1, 0)
        halo = partition.HaloExchange(shard, "cuda", mode="all_to_all")
        owned = partition.all_owned_lists(xd, eid, 1)
        pieces = partition.bc_pieces_from_model(model)
        pieces["cluster"] = lambda e, g: (clusters, n_cl)
        bg2, s2, emb2 = partition.distributed_bc_forward(pieces, shard, halo, xd[shard.owned_global], owned,
                                                         torch.cat([eid, eid.flip(0)], dim=1))
    key = lambda g: g[0] * 100000 + g[1]
    o_ref, o = torch.argsort(key(bg)), torch.argsort(key(bg2))
    assert torch.equal(key(bg)[o_ref], key(bg2)[o])
    assert s2.shape[0] == 480_000 * 5 and float((s2[o] - s_ref[o_ref]).abs().max()) <= 1e-4
    assert torch.allclose(emb2, emb[shard.owned_global], rtol=1e-4, atol=1e-5)
