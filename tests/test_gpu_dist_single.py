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
