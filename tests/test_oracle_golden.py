"""Pins the CPU oracle (oracle/) against the golden fixtures produced by running the
reference's own modules (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import hgnn_oracle as O
from oracle import c_oracle as C

TOL = 2e-6  # oracle vs reference: same arithmetic, only op-fusion / summation-order noise


def _cases(npz):
    return sorted({k.split(".")[0] for k in npz.files})


def test_k1_scatter_add_torch_oracle():
    z = load_golden("k1_scatter_add.npz")
    for c in _cases(z):
        out = O.scatter_add(torch.from_numpy(z[c + ".src"]), torch.from_numpy(z[c + ".index"]),
                            dim=0, dim_size=int(z[c + ".dim_size"]))
        assert out.shape == z[c + ".out"].shape, c
        assert rel_err(out.numpy(), z[c + ".out"]) <= TOL, c


def test_k1_scatter_add_c_oracle_bit_exact():
    """sequential arrival-order sum: the C loop reproduces the fixture bit for bit"""
    z = load_golden("k1_scatter_add.npz")
    for c in _cases(z):
        out = C.scatter_add(z[c + ".src"], z[c + ".index"], int(z[c + ".dim_size"]))
        assert np.array_equal(out, z[c + ".out"]), c


def test_k5_pool():
    z = load_golden("k5_pool.npz")
    nodes = torch.from_numpy(z["nodes"])
    bg = torch.from_numpy(z["bipartite_graph"])
    bw = torch.from_numpy(z["bipartite_edge_weights"])
    out = O.supernode_pool(nodes, bg, bw, z["out"].shape[0])
    assert rel_err(out.numpy(), z["out"]) <= TOL
    rs = 1.0 / np.maximum(np.abs(z["nodes"]).sum(1), 1e-12)
    outc = C.gather_scale_scatter(z["nodes"], z["bipartite_graph"][0], z["bipartite_graph"][1], z["out"].shape[0],
                                  weight=z["bipartite_edge_weights"], row_scale=rs.astype(np.float32))
    assert rel_err(outc, z["out"]) <= 1e-6


def _sd(z, prefix="sd."):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def _hp(latent):
    return dict(latent=latent, hidden=2 * latent, nb_edge_layer=2, nb_node_layer=3, layernorm=True,
                hidden_activation="GELU")


@pytest.mark.parametrize("latent", [32, 128])
def test_ignn_cell_forward_and_grads(latent):
    z = load_golden(f"ignn_cell_L{latent}.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in _sd(z).items()}
    nodes = torch.from_numpy(z["nodes"]).requires_grad_(True)
    edges = torch.from_numpy(z["edges"]).requires_grad_(True)
    graph = torch.from_numpy(z["graph"])
    on, oe = O.ignn_cell(sd, "", _hp(latent), nodes, edges, graph)
    assert rel_err(on.detach().numpy(), z["out_nodes"]) <= TOL
    assert rel_err(oe.detach().numpy(), z["out_edges"]) <= TOL
    ((on * torch.from_numpy(z["r_nodes"])).sum() + (oe * torch.from_numpy(z["r_edges"])).sum()).backward()
    assert rel_err(nodes.grad.numpy(), z["grad_nodes"]) <= 1e-5
    assert rel_err(edges.grad.numpy(), z["grad_edges"]) <= 1e-5
    for k, v in sd.items():
        assert rel_err(v.grad.numpy(), z["grad." + k]) <= 1e-5, k


@pytest.mark.parametrize("latent", [32, 64])
def test_hgnn_cell_forward_and_grads(latent):
    z = load_golden(f"hgnn_cell_L{latent}.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in _sd(z).items()}
    t = {k: torch.from_numpy(z[k]) for k in ("nodes", "edges", "supernodes", "superedges",
                                              "bipartite_edge_weights", "super_edge_weights")}
    for v in t.values():
        v.requires_grad_(True)
    outs = O.hgnn_cell(sd, "", _hp(latent), t["nodes"], t["edges"], t["supernodes"], t["superedges"],
                       torch.from_numpy(z["graph"]), torch.from_numpy(z["bipartite_graph"]),
                       t["bipartite_edge_weights"], torch.from_numpy(z["super_graph"]), t["super_edge_weights"])
    names = ("nodes", "edges", "supernodes", "superedges")
    for nm, o in zip(names, outs):
        assert rel_err(o.detach().numpy(), z["out_" + nm]) <= TOL, nm
    sum((o * torch.from_numpy(z["r_" + nm])).sum() for nm, o in zip(names, outs)).backward()
    for nm in names + ("bipartite_edge_weights", "super_edge_weights"):
        assert rel_err(t[nm].grad.numpy(), z["grad_" + nm]) <= 1e-5, nm
    for k, v in sd.items():
        assert rel_err(v.grad.numpy(), z["grad." + k]) <= 1e-5, k


def test_ec_in_forward_config1():
    """BASELINE config 1: flat EC-IN, latent=32, CPU"""
    z = load_golden("ec_in_L32.npz")
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
    scores = O.ec_in_forward(_sd(z), hp, torch.from_numpy(z["x"]), torch.from_numpy(z["edge_index"]))
    assert scores.shape == z["scores"].shape
    assert np.abs(scores.numpy() - z["scores"]).max() <= 5e-6
    assert int(z["n_params"]) == 286977


def test_bc_hgnn_call_sites_and_cells():
    """every scatter_add call site of a BC-HGNN-GMM forward (K1..K5) and the cell loop"""
    z = load_golden("bc_hgnn_L32.npz")
    for i in range(int(z["n_scatter"])):
        out = O.scatter_add(torch.from_numpy(z[f"scatter{i}.src"]), torch.from_numpy(z[f"scatter{i}.index"]),
                            dim=0, dim_size=int(z[f"scatter{i}.dim_size"]))
        assert rel_err(out.numpy(), z[f"scatter{i}.out"]) <= TOL, i
    hp = {k[3:]: z[k].item() for k in z.files if k.startswith("hp.")}
    sd = _sd(z)
    names = ["nodes", "edges", "supernodes", "superedges", "graph", "bipartite_graph",
             "bipartite_edge_weights", "super_graph", "super_edge_weights"]
    for i in range(int(z["n_cells"])):
        args = [torch.from_numpy(z[f"cell{i}.in.{n}"]) for n in names]
        outs = O.hgnn_cell(sd, f"hgnn_block.hgnn_cells.{i}.", hp, *args)
        for nm, o in zip(names[:4], outs):
            assert rel_err(o.numpy(), z[f"cell{i}.out.{nm}"]) <= TOL, (i, nm)


# ---------------------------------------------------------------- configs BASELINE.json names (latent 128 / 256 / 512)
import json
import os

import conftest
from golden import seeded  # tests/golden/seeded.py: weights as a function of (name, seed)


def _ref_configs():
    with open(os.path.join(conftest.GOLDEN, "ref_configs.json")) as f:
        return json.load(f)


def _seeded_model(cls, raw_hparams, z):
    model = cls(raw_hparams)
    seeded.fill_parameters(model, int(z["seed"]))
    seeded.check_parameters(model, z["param_checksums"])
    return model


def test_reference_yaml_configs_drop_in():
    """the reference's shipped YAMLs carry ``hidden: ratio`` (EdgeClassifier/Configs/IN.yaml:35-36,
    resolved by training_utils.py:13-15): the mirrors accept the raw dictionaries and build the reference's
    parameter counts"""
    from hierarchicalgnn_amd.models import BC_MessagePassing, EC_InteractionGNN
    cfg = _ref_configs()
    assert cfg["EC-IN"]["raw"]["hidden"] == "ratio" and cfg["BC-HGNN-GMM"]["raw"]["hidden"] == "ratio"
    with torch.device("meta"):
        ec = EC_InteractionGNN(cfg["EC-IN"]["raw"])
        bc = BC_MessagePassing(cfg["BC-HGNN-GMM"]["raw"])
    assert sum(p.numel() for p in ec.parameters()) == cfg["EC-IN"]["n_params"] == 4441089
    assert sum(p.numel() for p in bc.parameters()) == cfg["BC-HGNN-GMM"]["n_params"] == 25299957
    assert ec.hparams["hidden"] == cfg["EC-IN"]["resolved_hidden"] == 256
    assert bc.hparams["hidden"] == cfg["BC-HGNN-GMM"]["resolved_hidden"] == 512
    assert bc.hparams["cluster_granularity"] == 5
    with pytest.raises(KeyError):
        EC_InteractionGNN({k: v for k, v in cfg["EC-IN"]["raw"].items() if k != "hidden_ratio"})


def test_ec_in_config2_oracle_forward_and_input_gradient():
    """BASELINE config 2 (IN.yaml: latent 128, 14 cells): oracle == the reference's own class"""
    from hierarchicalgnn_amd.models import EC_InteractionGNN
    from hierarchicalgnn_amd.utils import process_hparams
    z = load_golden("ec_in_L128.npz")
    raw = _ref_configs()["EC-IN"]["raw"]
    model = _seeded_model(EC_InteractionGNN, raw, z)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    scores = O.ec_in_forward(sd, process_hparams(raw), x, torch.from_numpy(z["edge_index"]))
    assert np.abs(scores.detach().numpy() - z["scores"]).max() <= 5e-6
    (scores * torch.from_numpy(z["r_scores"])).sum().backward()
    assert rel_err(x.grad.numpy(), z["grad_x"]) <= 2e-5


def test_hgnn_cell_config3_width_oracle():
    from hierarchicalgnn_amd import HierarchicalGNNCell
    z = load_golden("hgnn_cell_L256.npz")
    L, seed = int(z["latent"]), int(z["seed"])
    cell = HierarchicalGNNCell(_hp(L))
    seeded.fill_parameters(cell, seed)
    seeded.check_parameters(cell, z["param_checksums"])
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in cell.state_dict().items()}
    names = ("nodes", "edges", "supernodes", "superedges")
    t = {k: seeded.randn(seed, "in." + k, *z["out_" + k].shape).requires_grad_(True) for k in names}
    bw = torch.from_numpy(z["bipartite_edge_weights"]).requires_grad_(True)
    sw = torch.from_numpy(z["super_edge_weights"]).requires_grad_(True)
    outs = O.hgnn_cell(sd, "", _hp(L), t["nodes"], t["edges"], t["supernodes"], t["superedges"],
                       torch.from_numpy(z["graph"]), torch.from_numpy(z["bipartite_graph"]), bw,
                       torch.from_numpy(z["super_graph"]), sw)
    for nm, o in zip(names, outs):
        assert rel_err(o.detach().numpy(), z["out_" + nm]) <= TOL, nm
    sum((o * seeded.randn(seed, "r." + nm, *o.shape)).sum() for nm, o in zip(names, outs)).backward()
    for nm in names:
        assert rel_err(t[nm].grad.numpy(), z["grad_" + nm]) <= 1e-5, nm
    assert rel_err(bw.grad.numpy(), z["grad_bipartite_edge_weights"]) <= 1e-5
    assert rel_err(sw.grad.numpy(), z["grad_super_edge_weights"]) <= 1e-5
    for k in ("edge_network.0.weight", "supernode_network.3.weight"):
        assert rel_err(sd[k].grad.numpy(), z["grad." + k]) <= 1e-5, k
    sk = seeded.grad_sketch((k, v.grad) for k, v in sd.items())
    assert np.abs(sk - z["grad_sketch"]).max() <= 1e-4 * np.abs(z["grad_sketch"]).max()


@pytest.mark.parametrize("latent", [256, 512])
def test_bc_hgnn_config3_and_config4_reference_oracle(latent):
    """BASELINE config 3 (HGNN_GMM.yaml: latent 256, 6 + 6 cells) and the fp32 reference of config 4
    (latent 512): the oracle's message passing on the captured hierarchy == the reference's forward"""
    from hierarchicalgnn_amd.models import BC_MessagePassing
    from hierarchicalgnn_amd.utils import process_hparams
    z = load_golden(f"bc_hgnn_L{latent}.npz")
    raw = dict(_ref_configs()["BC-HGNN-GMM"]["raw"], latent=latent)
    hp = process_hparams(raw)
    model = _seeded_model(BC_MessagePassing, raw, z)
    assert sum(p.numel() for p in model.parameters()) == int(z["n_params"])
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    x, graph = torch.from_numpy(z["x"]), torch.from_numpy(z["edge_index"])
    directed = torch.cat([graph, graph.flip(0)], dim=1)
    with torch.no_grad():
        emb, nodes, edges = O.bc_ignn_block(sd, hp, x, directed)
        assert rel_err(emb.numpy(), z["embeddings"]) <= 1e-5
        assert rel_err(nodes.numpy(), z["cell0.in.nodes"]) <= 1e-5
        assert rel_err(edges[torch.from_numpy(z["cell0.in.edges_rows"])].numpy(), z["cell0.in.edges_sub"]) <= 1e-5
        t = lambda k: torch.from_numpy(z[k])
        means = t("cell0.in.supernodes")[:, :hp["emb_dim"]]
        n_out, sn_out, sn0, se0 = O.bc_hgnn_block(sd, hp, nodes, edges, directed, means,
                                                  t("cell0.in.bipartite_graph"), t("cell0.in.bipartite_edge_weights"),
                                                  t("cell0.in.super_graph"), t("cell0.in.super_edge_weights"))
        assert rel_err(sn0.numpy(), z["cell0.in.supernodes"]) <= 1e-5
        assert rel_err(se0.numpy(), z["cell0.in.superedges"]) <= 1e-5
        assert rel_err(n_out.numpy(), z["last.out.nodes"]) <= 2e-5
        assert rel_err(sn_out.numpy(), z["last.out.supernodes"]) <= 2e-5
        scores = O.bc_scores(sd, hp, n_out, sn_out, t("bipartite_graph"))
        assert np.abs(scores.numpy() - z["bipartite_scores"]).max() <= 2e-5


def test_bc_hgnn_config3_training_step_oracle():
    """the oracle's BC-HGNN-GMM (training-mode BatchNorm in the attention weights, autograd through every stage)
    on the captured hierarchy decision == the reference's own training-mode forward + backward
    (make_golden.gen_bc_hgnn_backward): pins what the GPU test of the same fixture is held to"""
    from hierarchicalgnn_amd.models import BC_MessagePassing
    from hierarchicalgnn_amd.utils import process_hparams
    z = load_golden("bc_hgnn_train_L256.npz")
    raw = _ref_configs()["BC-HGNN-GMM"]["raw"]
    hp = process_hparams(raw)
    model = _seeded_model(BC_MessagePassing, raw, z)
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v.detach())
          for k, v in model.state_dict().items()}
    t = lambda k: torch.from_numpy(z[k])
    x = t("x").requires_grad_(True)
    graph = t("edge_index")
    directed = torch.cat([graph, graph.flip(0)], dim=1)
    emb, nodes, edges = O.bc_ignn_block(sd, hp, x, directed)
    emb.retain_grad()
    cl, n_cl = t("clusters"), int(z["n_clusters"])
    keep = cl >= 0
    sums = torch.zeros(n_cl, emb.shape[1]).index_add(0, cl[keep], emb[keep])                  # HGNN_GMM.py:251
    cnt = torch.zeros(n_cl).index_add(0, cl[keep], torch.ones(int(keep.sum()))).clamp(min=1)
    means = torch.nn.functional.normalize(sums / cnt.unsqueeze(1))
    bn = lambda name: [sd[f"hgnn_block.{name}.weight_normalization.{k}"] for k in
                       ("weight", "bias", "running_mean", "running_var")]
    sg, bg = t("super_graph"), t("bipartite_graph")
    sw, _ = O.graph_edge_weights(means, means, sg, *bn("super_graph_construction"), "sigmoid", True, training=True)
    bw, _ = O.graph_edge_weights(emb, means, bg, *bn("bipartite_graph_construction"), "exp", True, training=True)
    bw.retain_grad()
    sw.retain_grad()
    assert rel_err(emb.detach().numpy(), z["embeddings"]) <= 1e-5
    assert rel_err(bw.detach().numpy(), z["bipartite_edge_weights"]) <= 1e-5
    assert rel_err(sw.detach().numpy(), z["super_edge_weights"]) <= 1e-5
    n_out, sn_out, _, _ = O.bc_hgnn_block(sd, hp, nodes, edges, directed, means, bg, bw, sg, sw)
    scores = O.bc_scores(sd, hp, n_out, sn_out, bg)
    assert np.abs(scores.detach().numpy() - z["bipartite_scores"]).max() <= 2e-5
    loss = (scores * t("r_scores")).sum() + float(z["c_emb"]) * (emb * emb.roll(1, 0)).sum()
    assert abs(float(loss) - float(z["loss"])) <= 1e-4
    loss.backward()
    assert rel_err(x.grad.numpy(), z["grad_x"]) <= 5e-5
    assert rel_err(emb.grad.numpy(), z["grad_embeddings"]) <= 5e-5
    assert rel_err(bw.grad.numpy(), z["grad_bipartite_edge_weights"]) <= 5e-5
    assert rel_err(sw.grad.numpy(), z["grad_super_edge_weights"]) <= 5e-5
    for k in [f[5:] for f in z.files if f.startswith("grad.")]:
        assert rel_err(sd[k].grad.numpy(), z["grad." + k]) <= 5e-5, k
    names = sorted(n for n, _ in model.named_parameters())
    sk = seeded.grad_sketch((n, sd[n].grad if sd[n].grad is not None else torch.zeros_like(sd[n])) for n in names)
    ref = z["grad_sketch"]
    assert np.abs(sk[:, 1] - ref[:, 1]).max() <= 1e-4 * np.abs(ref[:, 1]).max()
    rel = np.abs(sk[:, 1] - ref[:, 1]) / np.maximum(ref[:, 1], 1e-12)
    worst = [(names[i], float(rel[i]), float(ref[i, 1])) for i in np.argsort(-rel)[:5]]
    # per-parameter |.|-sums within 2e-4; the BatchNorm affine in front of a mean-normalised weight (gnn_utils.py:213)
    # has a true gradient of ~0 (cancellation residue), it is held to an absolute floor instead
    assert np.all(np.abs(sk[:, 1] - ref[:, 1]) <= 2e-4 * ref[:, 1] + 1e-4), worst
