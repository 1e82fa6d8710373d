"""Node-partitioned events across the GPUs of one node, with a halo exchange per cell.

The reference is single-GPU (Notebooks/script.py:35, README.md:65); this module is
the MI355X-side design SURVEY.md section 8e describes, with no reference counterpart:

  * hits are ordered by phi and cut into ``P`` contiguous wedges balanced on
    in-degree (work is proportional to the rows a rank aggregates, not to its hits);
  * a directed edge (u -> v) belongs to ``owner(v)``: the K1 aggregation
    (Modules/gnn_utils.py:50) is purely local and needs no communication;
  * only the edge update (gnn_utils.py:61) reads remote data -- ``nodes[u]`` of
    cut edges -- so each cell has exactly one exchange step: after the node
    update every rank sends the rows of its boundary hits to the ranks that
    own an edge leaving them (RCCL over xGMI; point-to-point grouped send/recv
    = ``all_to_all_single`` with ragged splits, or a padded ``all_gather`` of
    the boundary blocks -- ``mode``);
  * HGNN supernodes / superedges are small ([S,L] ~ 10 MB) and stay replicated;
    the node->supernode sums (K3/K5) are computed on the owned hits and combined
    with one ``all_reduce``.

Everything here is index bookkeeping plus ``torch.distributed`` calls, so it runs
unchanged under the ``gloo`` backend on CPU tensors (tests/test_partition_gloo.py);
on the GPU the pack/unpack gathers go through the HIP row-gather kernel.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch


@dataclass
class EventShard:
    rank: int
    world: int
    n_global: int
    n_owned: int
    n_halo: int
    owned_global: torch.Tensor      # [n_owned]  global hit id of local row i
    halo_global: torch.Tensor       # [n_halo]   global hit id of halo row n_owned + j (grouped by owner)
    local_graph: torch.Tensor       # [2, M_p]   row 0: local source id (owned or halo), row 1: owned destination
    edge_global: torch.Tensor       # [M_p]      column of the global DIRECTED graph each local edge is
    send_index: torch.Tensor        # [sum(send_splits)] owned local ids, grouped by destination rank
    send_splits: List[int] = field(default_factory=list)
    recv_splits: List[int] = field(default_factory=list)
    n_interior: int = 0             # local edges [0, n_interior) have an OWNED source (no halo row needed): the
                                    # edge update of these can run while the halo exchange is in flight
    partition_s: float = 0.0        # wall seconds partition_event took (host-synchronised)


def node_owner(x: torch.Tensor, directed_graph: torch.Tensor, world: int, return_order: bool = False):
    """phi-ordered contiguous wedges, balanced on in-degree.  Returns (owner[N], pos[N]) (+ order[N] = hit at
    phi position).  Runs on the device of its inputs with no host loop; the balance arithmetic is EXACT integer
    arithmetic (weight of a hit = 1000 in-degree + 1: hits without edges still cost a row), so every rank -- and
    the CPU and the GPU -- derive bit-identical cuts."""
    n = x.shape[0]
    dev = x.device
    order = torch.argsort(x[:, 1], stable=True)
    pos = torch.empty(n, dtype=torch.long, device=dev)
    pos[order] = torch.arange(n, device=dev)
    w = torch.bincount(directed_graph[1], minlength=n) * 1000 + 1
    cum = torch.cumsum(w[order], 0)
    if n:
        total = cum[-1]
        bounds = (total * torch.arange(1, world, device=dev, dtype=torch.long) + world - 1) // world  # ceil(total k / P)
        cuts = torch.searchsorted(cum, bounds)
        owner_sorted = torch.searchsorted(cuts, torch.arange(n, device=dev), right=True)
    else:
        owner_sorted = torch.zeros(0, dtype=torch.long, device=dev)
    owner = torch.empty(n, dtype=torch.long, device=dev)
    owner[order] = owner_sorted
    if return_order:
        return owner, pos, order
    return owner, pos


def _grouped_unique(group: torch.Tensor, nodes: torch.Tensor, pos: torch.Tensor, order: torch.Tensor, n: int, world: int):
    """distinct ``nodes`` per ``group`` (a rank id), groups ascending, phi order inside a group:
    (node ids [sum], counts per group [world]) -- ONE sort-unique of a composite key instead of a loop over peers"""
    key = torch.unique(group * n + pos[nodes])                      # sorted
    counts = torch.bincount(torch.div(key, n, rounding_mode="floor"), minlength=world)
    return order[key % n], counts


def partition_event(x: torch.Tensor, edge_index: torch.Tensor, world: int, rank: int,
                    already_directed: bool = False, device=None) -> EventShard:
    """Shard one event.  ``edge_index`` is the stored [2,E] graph; it is doubled exactly as the model does
    (EdgeClassifier/Models/IN.py:122) unless ``already_directed``.  Deterministic: every rank derives the same
    partition.  Runs on ``device`` (default: where the inputs live) with device-wide sort / scan / unique /
    compaction primitives and no per-peer host loop: on an MI355X the full-pileup event (480k hits, 8M directed
    edges) partitions in milliseconds instead of the 0.15-0.6 s of the former CPU loops; the gloo CPU tests run the
    same code on CPU tensors.

    Local edge order: edges whose SOURCE is owned first (``n_interior`` of them), then the cut edges whose source is
    a halo row; inside each class the order of the global directed graph is kept."""
    import time
    t0 = time.perf_counter()
    if device is not None:
        x, edge_index = x.to(device), edge_index.to(device)
    dev = x.device
    graph = edge_index if already_directed else torch.cat([edge_index, edge_index.flip(0)], dim=1)
    n = x.shape[0]
    owner, pos, order = node_owner(x, graph, world, return_order=True)
    src, dst = graph[0], graph[1]
    so, do = owner[src], owner[dst]
    mine = do == rank
    local_src = so == rank
    e_int = torch.nonzero(mine & local_src).squeeze(1)
    e_bnd = torch.nonzero(mine & ~local_src).squeeze(1)
    edge_global = torch.cat([e_int, e_bnd])
    n_interior = int(e_int.numel())
    e_src, e_dst = src[edge_global], dst[edge_global]

    owned_global = order[owner[order] == rank]                         # phi order inside the wedge
    n_owned = int(owned_global.numel())
    g2l = torch.full((n,), -1, dtype=torch.long, device=dev)
    g2l[owned_global] = torch.arange(n_owned, device=dev)

    # halo: remote sources of my cut edges, grouped by owner, phi-ordered inside a group
    b_src = src[e_bnd]
    halo_global, recv_counts = _grouped_unique(owner[b_src], b_src, pos, order, n, world)
    g2l[halo_global] = n_owned + torch.arange(halo_global.numel(), device=dev)

    # what I must send: my hits that are sources of edges owned by q (same order as q's halo group for me)
    out_cut = torch.nonzero(local_src & ~mine).squeeze(1)
    send_nodes, send_counts = _grouped_unique(do[out_cut], src[out_cut], pos, order, n, world)
    send_index = g2l[send_nodes]

    local_graph = torch.stack([g2l[e_src], g2l[e_dst]]).contiguous()
    splits = torch.stack([send_counts, recv_counts]).tolist()         # the ONE host read of the partition
    if local_graph.numel():
        assert int(local_graph.min()) >= 0
    return EventShard(rank=rank, world=world, n_global=n, n_owned=n_owned, n_halo=int(halo_global.numel()),
                      owned_global=owned_global, halo_global=halo_global, local_graph=local_graph,
                      edge_global=edge_global, send_index=send_index.contiguous(),
                      send_splits=[int(v) for v in splits[0]], recv_splits=[int(v) for v in splits[1]],
                      n_interior=n_interior, partition_s=time.perf_counter() - t0)


def _pack(rows: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    if rows.is_cuda:
        from .ops import gather_rows
        return gather_rows(rows, index)
    return rows.index_select(0, index)       # gloo / CPU rehearsal of the same data path


class _A2A(torch.autograd.Function):
    """ragged all-to-all of packed rows; backward is the reverse all-to-all"""

    @staticmethod
    def forward(ctx, send, send_splits, recv_splits, group):
        import torch.distributed as dist
        ctx.splits = (send_splits, recv_splits)
        ctx.group = group
        recv = send.new_empty((sum(recv_splits),) + tuple(send.shape[1:]))
        dist.all_to_all_single(recv, send.contiguous(), recv_splits, send_splits, group=group)
        return recv

    @staticmethod
    def backward(ctx, grad_recv):
        import torch.distributed as dist
        send_splits, recv_splits = ctx.splits
        grad_send = grad_recv.new_empty((sum(send_splits),) + tuple(grad_recv.shape[1:]))
        dist.all_to_all_single(grad_send, grad_recv.contiguous(), send_splits, recv_splits, group=ctx.group)
        return grad_send, None, None, None


class _AllGatherBlocks(torch.autograd.Function):
    """padded all_gather of every rank's boundary block; backward = reduce_scatter (sum)"""

    @staticmethod
    def forward(ctx, block, group):
        import torch.distributed as dist
        world = dist.get_world_size(group)
        ctx.group = group
        out = block.new_empty((world * block.shape[0],) + tuple(block.shape[1:]))
        dist.all_gather_into_tensor(out, block.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        import torch.distributed as dist
        world = dist.get_world_size(ctx.group)
        grad_block = grad_out.new_empty((grad_out.shape[0] // world,) + tuple(grad_out.shape[1:]))
        g = grad_out.contiguous()
        if g.is_cuda:
            dist.reduce_scatter_tensor(grad_block, g, group=ctx.group)
        else:  # gloo has no reduce_scatter: all_reduce then slice
            dist.all_reduce(g, group=ctx.group)
            r = dist.get_rank(ctx.group)
            grad_block = g[r * grad_block.shape[0]:(r + 1) * grad_block.shape[0]].clone()
        return grad_block, None


class HaloExchange:
    """One exchange step of a cell: ``extend(nodes_owned) -> [n_owned + n_halo, L]``.

    mode "all_to_all": each rank receives exactly the rows it reads (grouped send/recv,
                       one message per neighbour pair; phi-wedges make most pairs empty).
    mode "all_gather": every rank contributes its padded boundary block and gathers
                       all of them (one collective, more bytes); the halo is then a
                       row gather out of the gathered buffer.
    Differentiable: gradients of halo rows flow back to their owners.
    """

    def __init__(self, shard: EventShard, device=None, mode: str = "all_to_all", group=None):
        import torch.distributed as dist
        self.shard = shard
        self.mode = mode
        self.group = group
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.send_index = shard.send_index.to(self.device)
        self.send_splits = list(shard.send_splits)
        self.recv_splits = list(shard.recv_splits)
        if mode == "all_gather":
            # boundary block = my hits any peer needs (deduplicated), padded to the global max
            dev = self.device
            send_index = shard.send_index.to(dev)
            owned_global = shard.owned_global.to(dev)
            halo_global = shard.halo_global.to(dev)
            boundary = torch.unique(send_index)
            t = torch.tensor([boundary.numel()], dtype=torch.long, device=dev)
            sizes = [torch.zeros_like(t) for _ in range(shard.world)]
            dist.all_gather(sizes, t, group=group)
            self.block = max(1, int(torch.stack(sizes).max().item()))
            pad = torch.zeros(self.block - boundary.numel(), dtype=torch.long, device=dev)
            self.boundary_index = torch.cat([boundary, pad])
            # where, in the gathered [world*block] buffer, does each of my halo rows sit?
            # every rank needs the boundary lists of its peers: exchange them once.
            mine = torch.full((self.block,), -1, dtype=torch.long, device=dev)
            mine[:boundary.numel()] = owned_global[boundary]
            lists = [torch.empty(self.block, dtype=torch.long, device=dev) for _ in range(shard.world)]
            dist.all_gather(lists, mine, group=group)
            flat = torch.cat(lists)
            # position of every halo hit inside the gathered buffer (vectorised lookup)
            valid = torch.nonzero(flat >= 0).squeeze(1)
            keys, order = torch.sort(flat[valid])
            if halo_global.numel():
                pos = torch.searchsorted(keys, halo_global)
                assert bool((keys[pos.clamp(max=max(keys.numel() - 1, 0))] == halo_global).all())
                self.halo_from_gathered = valid[order[pos]]
            else:
                self.halo_from_gathered = torch.zeros(0, dtype=torch.long, device=dev)
        elif mode != "all_to_all":
            raise ValueError(mode)

    def side_stream(self):
        """the HIP stream the overlapped exchange runs on (one per HaloExchange, created on first use)"""
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(self.device)
        return self._side

    def exchange(self, nodes_owned: torch.Tensor) -> torch.Tensor:
        """rows of the remote sources of my edges, [n_halo, L]"""
        if self.mode == "all_to_all":
            send = _pack(nodes_owned, self.send_index)
            return _A2A.apply(send, self.send_splits, self.recv_splits, self.group)
        block = _pack(nodes_owned, self.boundary_index)
        gathered = _AllGatherBlocks.apply(block, self.group)
        return _pack(gathered, self.halo_from_gathered)

    def extend(self, nodes_owned: torch.Tensor) -> torch.Tensor:
        return torch.cat([nodes_owned, self.exchange(nodes_owned)], dim=0)

    def halo_bytes(self, latent: int) -> int:
        return 4 * latent * self.shard.n_halo


def _graph_cols(graph, a, b):
    """columns [a, b) of a [2, M] graph as a contiguous [2, b - a] tensor whose rows are VIEWS-stable per call site:
    plans / int32 copies are cached on the identity of the index tensors, so the slices are made once per shard"""
    return graph[:, a:b].contiguous()


def _split_edge_update(cell, halo: "HaloExchange", nodes_owned, edges_local, local_graph, n_interior):
    """The edge update of a shard (Modules/gnn_utils.py:57-64) with the halo exchange HIDDEN behind the interior
    edges: edges [0, n_interior) have an owned source, so their MLP needs no remote row and runs on
    the current stream while the exchange (pack kernel -> RCCL -> unpack) runs on a side stream; the cut edges
    [n_interior, M) -- a fraction of a percent with phi-wedges -- follow once the halo has landed.  Without autograd
    both calls write straight into one output table (no concatenation); under autograd the two results are
    concatenated (the backward of the exchange returns halo gradients to their owners as before)."""
    M = int(edges_local.shape[0])
    ni = int(n_interior)
    cache = halo.__dict__.setdefault("_split_graphs", {})
    key = (local_graph.data_ptr(), M, ni)
    if key not in cache:
        cache.clear()
        cache[key] = (_graph_cols(local_graph, 0, ni), _graph_cols(local_graph, ni, M), local_graph)
    g_int, g_bnd, _ = cache[key]
    e_int, e_bnd = edges_local[:ni], edges_local[ni:]
    # zero-copy assembly needs a cell whose edge_update writes into caller-supplied rows (the HIP cells do)
    no_grad = not (torch.is_grad_enabled() and (nodes_owned.requires_grad or edges_local.requires_grad)) \
        and getattr(cell, "edge_update_takes_out", False)
    out = torch.empty_like(edges_local) if no_grad else None

    def interior():
        if ni == 0:
            return e_int
        return cell.edge_update(nodes_owned, e_int, g_int, out=out[:ni]) if no_grad \
            else cell.edge_update(nodes_owned, e_int, g_int)

    if nodes_owned.is_cuda:
        cur = torch.cuda.current_stream(nodes_owned.device)
        side = halo.side_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            halo_rows = halo.exchange(nodes_owned)
        nodes_owned.record_stream(side)
        new_int = interior()
        cur.wait_stream(side)
        halo_rows.record_stream(cur)
    else:                                            # gloo / CPU rehearsal: same schedule, sequentially
        halo_rows = halo.exchange(nodes_owned)
        new_int = interior()
    if M == ni:
        return new_int if not no_grad else out
    nodes_ext = torch.cat([nodes_owned, halo_rows], dim=0)
    if no_grad:
        cell.edge_update(nodes_ext, e_bnd, g_bnd, out=out[ni:])
        return out
    new_bnd = cell.edge_update(nodes_ext, e_bnd, g_bnd)
    return torch.cat([new_int, new_bnd], dim=0)


def distributed_cell_forward(cell, halo: HaloExchange, nodes_owned, edges_local, local_graph, overlap: bool = True):
    """InteractionGNNCell.forward (Modules/gnn_utils.py:66-71) on one shard:
    local aggregation + node MLP, ONE halo exchange, then the edge update.  ``overlap``: run the exchange on a side
    stream under the interior edges' MLP (needs the shard's interior-first edge order, ``shard.n_interior``)."""
    nodes_owned = cell.node_update(nodes_owned, edges_local, local_graph)
    if overlap and halo.shard.n_interior > 0 and int(edges_local.shape[0]) == int(halo.shard.local_graph.shape[1]):
        edges_local = _split_edge_update(cell, halo, nodes_owned, edges_local, local_graph, halo.shard.n_interior)
        return nodes_owned, edges_local
    nodes_ext = halo.extend(nodes_owned)
    edges_local = cell.edge_update(nodes_ext, edges_local, local_graph)
    return nodes_owned, edges_local


class _AllReduceSum(torch.autograd.Function):
    """y = sum over ranks of x (replicated result).  Every rank then runs the same replicated
    computation on y but contributes its own local loss terms, so dL/dx_q = sum_r dL_r/dy_r:
    the backward is again an all_reduce(SUM)."""

    @staticmethod
    def forward(ctx, t, group):
        import torch.distributed as dist
        ctx.group = group
        out = t.clone()
        dist.all_reduce(out, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        g = g.contiguous().clone()
        dist.all_reduce(g, group=ctx.group)
        return g, None


def allreduce_supernode_sums(partial: torch.Tensor, group=None) -> torch.Tensor:
    """K3/K5 across shards: each rank sums its owned hits into the replicated [S,L]
    supernode table; one all_reduce combines them."""
    return _AllReduceSum.apply(partial, group)


def allreduce_gradients(params, group=None, bucket_bytes: int = 256 << 20) -> None:
    """Sum the gradients of the REPLICATED parameters over the ranks of a node-partitioned event.

    Every rank runs the same weights over its own shard (and the small replicated supernode /
    superedge updates over identical inputs with its share of their loss terms), so after
    ``backward`` each ``p.grad`` is a partial sum; the optimiser needs the total.  Gradients are
    packed into few large fp32 buckets (ring all-reduce over xGMI is per-link bound: large messages,
    few collectives) and summed in place.  A parameter without a gradient on this rank contributes
    zeros, so every rank issues identical collectives."""
    import torch.distributed as dist
    params = [p for p in params if p.requires_grad]
    if not params or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    i = 0
    while i < len(params):
        j, size = i, 0
        while j < len(params) and (j == i or (size + params[j].numel()) * 4 <= bucket_bytes):
            size += params[j].numel()
            j += 1
        chunk = params[i:j]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float()
                          for p in chunk])
        dist.all_reduce(flat, group=group)
        off = 0
        for p in chunk:
            g = flat[off:off + p.numel()].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += p.numel()
        i = j


def shard_bipartite(shard: EventShard, bipartite_graph: torch.Tensor, bipartite_edge_weights: torch.Tensor):
    """keep the bipartite edges whose hit this rank owns; hit ids become local, supernode ids
    stay global (supernodes are replicated).  Returns (graph[2,B_p], weights[B_p,1], selection)."""
    g2l = torch.full((shard.n_global,), -1, dtype=torch.long)
    g2l[shard.owned_global] = torch.arange(shard.n_owned)
    loc = g2l[bipartite_graph[0]]
    sel = torch.nonzero(loc >= 0).squeeze(1)
    return torch.stack([loc[sel], bipartite_graph[1][sel]]).contiguous(), bipartite_edge_weights[sel], sel


def distributed_hgnn_cell_forward(cell, halo: HaloExchange, nodes_owned, edges_local, supernodes, superedges,
                                  local_graph, bipartite_local, bipartite_w_local, super_graph, super_w, group=None,
                                  overlap: bool = True):
    """HierarchicalGNNCell.forward (Modules/gnn_utils.py:155-169) on one shard.  Supernodes and
    superedges are replicated (every rank computes their small updates redundantly); the only
    collectives are the all_reduce of the node->supernode sums and the halo exchange before the
    edge update.  The reduce is handed to ``supernode_update`` as an argument (not installed and removed
    around the call): under the cell's reentrant checkpoint the backward re-runs the update, and the
    recompute must issue the same all_reduce.  Gradients of the replicated parameters are partial sums
    on every rank: ``allreduce_gradients`` combines them after backward."""
    supernodes = cell.supernode_update(nodes_owned, supernodes, superedges, bipartite_local,
                                       bipartite_w_local, super_graph, super_w,
                                       node_message_reduce=lambda t: allreduce_supernode_sums(t, group))
    nodes_owned = cell.node_update(nodes_owned, edges_local, supernodes, local_graph, bipartite_local,
                                   bipartite_w_local)
    if overlap and halo.shard.n_interior > 0 and int(edges_local.shape[0]) == int(halo.shard.local_graph.shape[1]):
        # the exchange also hides behind the (replicated, small) superedge update
        superedges = cell.superedge_update(supernodes, superedges, super_graph, super_w)
        edges_local = _split_edge_update(cell, halo, nodes_owned, edges_local, local_graph, halo.shard.n_interior)
        return nodes_owned, edges_local, supernodes, superedges
    superedges = cell.superedge_update(supernodes, superedges, super_graph, super_w)
    nodes_ext = halo.extend(nodes_owned)
    edges_local = cell.edge_update(nodes_ext, edges_local, local_graph)
    return nodes_owned, edges_local, supernodes, superedges


# --------------------------------------------------------------------------- whole EC-IN model on shards
@dataclass
class EdgePairExchange:
    """The EC head pairs the latent of stored edge k (u -> v, owned by owner(v)) with the latent of its
    reverse E+k (v -> u, owned by owner(u)) (EdgeClassifier/Models/IN.py:126).  For cut edges the two
    halves live on different ranks: one ragged all-to-all per forward ships every reverse half to the
    owner of its forward half."""
    fwd_local: torch.Tensor      # [F_p] local indices of my forward edges (global directed id < E)
    fwd_global: torch.Tensor     # [F_p] their stored-edge ids k
    partner_local: torch.Tensor  # [F_p] local index of the reverse half, or -1 if it arrives by exchange
    partner_recv: torch.Tensor   # [F_p] row in the receive buffer, or -1 if local
    send_index: torch.Tensor     # local indices of the reverse halves I must send, grouped by destination
    send_splits: List[int] = field(default_factory=list)
    recv_splits: List[int] = field(default_factory=list)


def edge_pair_exchange(x, edge_index, world: int, rank: int, shard: EventShard) -> EdgePairExchange:
    E = int(edge_index.shape[1])
    graph = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    owner, _ = node_owner(x, graph, world)
    edge_owner = owner[graph[1]]                      # owner of every directed edge
    g2l = torch.full((2 * E,), -1, dtype=torch.long)
    g2l[shard.edge_global] = torch.arange(shard.edge_global.numel())
    k = torch.arange(E)
    mine_fwd = edge_owner[:E] == rank
    fwd_global = k[mine_fwd]
    fwd_local = g2l[fwd_global]
    partner_owner = edge_owner[E:][mine_fwd]
    partner_local = torch.where(partner_owner == rank, g2l[fwd_global + E], torch.full_like(fwd_global, -1))
    partner_recv = torch.full_like(fwd_global, -1)
    recv_splits, off = [], 0
    for a in range(world):
        sel = partner_owner == a
        n = int(sel.sum()) if a != rank else 0
        if a != rank:
            partner_recv[sel] = off + torch.arange(n)  # sender a sends in increasing k order
        recv_splits.append(n)
        off += n
    send_splits, parts = [], []
    for q in range(world):
        if q == rank:
            send_splits.append(0)
            continue
        sel = (edge_owner[E:] == rank) & (edge_owner[:E] == q)   # my reverse halves whose forward half is on q
        parts.append(g2l[k[sel] + E])
        send_splits.append(int(sel.sum()))
    send_index = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.long)
    return EdgePairExchange(fwd_local, fwd_global, partner_local, partner_recv, send_index.contiguous(),
                            send_splits, recv_splits)


def distributed_ec_forward(node_encode, edge_encode, cells, head, shard: EventShard, halo: HaloExchange,
                           pairs: EdgePairExchange, x_owned: torch.Tensor, group=None):
    """EC_InteractionGNN.forward (EdgeClassifier/Models/IN.py:80-128) on one shard of a node-partitioned
    event.  ``node_encode(x)``, ``edge_encode(x_ext, graph)``, ``cells`` (objects with node_update /
    edge_update) and ``head(pair_rows)`` are the model's own pieces.  Collectives: one halo exchange of
    the 3 input coordinates, one per cell (latent rows), one edge-pair exchange before the head.
    Returns (scores of my forward edges, their stored-edge ids)."""
    dev = x_owned.device
    graph = shard.local_graph.to(dev)
    x_ext = halo.extend(x_owned)
    nodes = node_encode(x_owned)
    edges = edge_encode(x_ext, graph)
    for cell in cells:
        nodes, edges = distributed_cell_forward(cell, halo, nodes, edges, graph)
    recv = _A2A.apply(_pack(edges, pairs.send_index.to(dev)), pairs.send_splits, pairs.recv_splits, group)
    both = torch.cat([edges, recv], dim=0)
    partner = torch.where(pairs.partner_local >= 0, pairs.partner_local,
                          edges.shape[0] + pairs.partner_recv).to(dev)
    rows = torch.cat([_pack(edges, pairs.fwd_local.to(dev)), _pack(both, partner)], dim=1)
    return torch.sigmoid(head(rows).squeeze(-1)), pairs.fwd_global


def distributed_ec_forward_model(model, shard: EventShard, halo: HaloExchange, pairs: EdgePairExchange,
                                 x_owned: torch.Tensor, group=None):
    """``distributed_ec_forward`` with the pieces of a ``hierarchicalgnn_amd.models.EC_InteractionGNN``"""
    from .mlp import concat_mlp
    blk = model.ignn_block
    return distributed_ec_forward(blk._encode_nodes, blk._encode_edges, list(blk.ignn_cells),
                                  lambda rows: concat_mlp(model.edge_classifier, [(rows, None)]),
                                  shard, halo, pairs, x_owned, group)


# --------------------------------------------------------------------------- whole BC-HGNN-GMM model on shards
def all_owned_lists(x: torch.Tensor, edge_index: torch.Tensor, world: int) -> List[torch.Tensor]:
    """global hit ids of every rank's owned block, in local-row order (what partition_event gives a rank for
    itself, for all ranks: deterministic, every rank derives the same lists)"""
    graph = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    owner, pos = node_owner(x, graph, world)
    out = []
    for r in range(world):
        own = torch.nonzero(owner == r).squeeze(1)
        out.append(own[torch.argsort(pos[own])])
    return out


class _AllGatherRows(torch.autograd.Function):
    """all ranks' row blocks (different lengths) -> one [N_global, F] table in global row order on every rank.
    Padded all_gather; backward: every rank receives the sum over ranks of the gradient rows of ITS block
    (the table is consumed by replicated computations whose loss terms are split over the ranks)."""

    @staticmethod
    def forward(ctx, rows, owned_lists, rank, group):
        import torch.distributed as dist
        world = len(owned_lists)
        block = max(int(o.numel()) for o in owned_lists)
        pad = rows.new_zeros((block,) + tuple(rows.shape[1:]))
        pad[:rows.shape[0]] = rows
        gathered = rows.new_empty((world * block,) + tuple(rows.shape[1:]))
        dist.all_gather_into_tensor(gathered, pad.contiguous(), group=group)
        n_global = sum(int(o.numel()) for o in owned_lists)
        table = rows.new_zeros((n_global,) + tuple(rows.shape[1:]))
        for r, own in enumerate(owned_lists):
            table[own.to(rows.device)] = gathered[r * block:r * block + own.numel()]
        ctx.owned, ctx.rank, ctx.group = owned_lists[rank], rank, group
        return table

    @staticmethod
    def backward(ctx, grad_table):
        import torch.distributed as dist
        g = grad_table.contiguous().clone()
        dist.all_reduce(g, group=ctx.group)
        return g[ctx.owned.to(g.device)], None, None, None


def distributed_bc_forward(pieces, shard: EventShard, halo: HaloExchange, x_owned: torch.Tensor,
                           owned_lists: List[torch.Tensor], directed_global: torch.Tensor, group=None):
    """BC_HierarchicalGNN_GMM.forward (BipartiteClassification/Models/HGNN_GMM.py:323-346) on one shard of a
    node-partitioned event (BASELINE config 5).  ``pieces`` holds the model's own callables:

        node_encode(x), edge_encode(x_ext, graph), ignn_cells, emb_head(nodes) -> L2-normalised embeddings,
        cluster(emb_global, directed_global) -> (clusters[N], n_clusters)        (no gradients),
        centroids(emb_global, clusters, n_clusters) -> means[S, emb_dim],
        super_graph(means) -> (graph[2,Q], weights[Q,1]),
        bipartite(emb_owned, means) -> (graph[2,B_p] LOCAL hit / global supernode ids, logits-free raw weights[B_p,1]),
        pool(nodes_owned, bg, bw, S) -> partial sums, supernode_encode(pooled), superedge_encode(supernodes, sg),
        hgnn_cells, head(rows)

    What is replicated and what is sharded: hits, edges and bipartite edges are sharded by the owner of the hit
    (destination); the 8-wide embeddings are all-gathered once (N x 8 floats: 15 MB at full pileup) and the
    hierarchy decision -- GMM cut + connected components over the global directed graph, 1.4 ms per 2M edges on
    one MI355X -- is run REDUNDANTLY by every rank on identical inputs (deterministic kernels => identical
    clusters everywhere, no label-exchange protocol); centroids, the super graph, supernodes and superedges are
    replicated; the node->supernode sums (K5, K3) are all-reduced; the bipartite weights' mean-normalisation
    (gnn_utils.py:213) uses the all-reduced global mean; in training mode the BatchNorm in front of those weights
    (gnn_utils.py:209) uses batch statistics synchronised over the ranks (``bc_pieces_from_model``).
    Returns (bipartite graph [global hit id, supernode id], scores, embeddings of the owned hits)."""
    dev = x_owned.device
    graph = shard.local_graph.to(dev)
    x_ext = halo.extend(x_owned)
    nodes = pieces["node_encode"](x_owned)
    edges = pieces["edge_encode"](x_ext, graph)
    for cell in pieces["ignn_cells"]:
        nodes, edges = distributed_cell_forward(cell, halo, nodes, edges, graph)
    emb_owned = pieces["emb_head"](nodes)
    emb_global = _AllGatherRows.apply(emb_owned, owned_lists, shard.rank, group)
    with torch.no_grad():
        clusters, n_clusters = pieces["cluster"](emb_global, directed_global)
    means = pieces["centroids"](emb_global, clusters, n_clusters)
    sg, sw = pieces["super_graph"](means)
    bg, bw_raw = pieces["bipartite"](emb_owned, means)
    # gnn_utils.py:213 norm=True: divide by the mean over ALL bipartite edges of the event
    tot = allreduce_supernode_sums(torch.stack([bw_raw.sum(), bw_raw.new_tensor(float(bw_raw.numel()))]), group)
    bw = bw_raw / (tot[0] / tot[1].clamp(min=1))
    pooled = allreduce_supernode_sums(pieces["pool"](nodes, bg, bw, means.shape[0]), group)
    supernodes = torch.cat([means.to(pooled.dtype), pieces["supernode_encode"](pooled)], dim=-1)
    superedges = pieces["superedge_encode"](supernodes, sg)
    for cell in pieces["hgnn_cells"]:
        nodes, edges, supernodes, superedges = distributed_hgnn_cell_forward(
            cell, halo, nodes, edges, supernodes, superedges, graph, bg, bw, sg, sw, group)
    rows = torch.cat([_pack(nodes, bg[0]), _pack(supernodes, bg[1])], dim=1)
    scores = torch.sigmoid(pieces["head"](rows).squeeze(-1))
    bg_global = torch.stack([shard.owned_global.to(dev)[bg[0]], bg[1]])
    return bg_global, scores, emb_owned


def bc_pieces_from_model(model, group=None):
    """``pieces`` of a ``hierarchicalgnn_amd.models.BC_MessagePassing`` for ``distributed_bc_forward`` (GPU).
    Training mode: the bipartite attention weights' BatchNorm1d(1) (gnn_utils.py:179,209) sees only this rank's
    bipartite edges, so its batch statistics are SYNCHRONISED over ``group`` (graph_construction.batch_norm_1); the
    super graph is replicated (identical inputs on every rank), its BatchNorm needs no collective."""
    from .mlp import concat_mlp
    from .ops import gather_scale_scatter, l1_row_scale, scatter_add
    import torch.nn as nn
    blk, hb = model.ignn_block, model.hgnn_block
    hp = model.hparams
    sync_stats = lambda t: allreduce_supernode_sums(t, group)

    def centroids(emb, clusters, n):
        lab = torch.where(clusters >= 0, clusters, torch.full_like(clusters, n)).contiguous()
        sums = scatter_add(emb, lab, dim=0, dim_size=n + 1, validate=False)[:n]
        cnt = scatter_add(torch.ones(emb.shape[0], 1, device=emb.device), lab, dim=0, dim_size=n + 1,
                          validate=False)[:n].clamp_(min=1)
        return nn.functional.normalize(sums / cnt)

    def bipartite(emb_owned, means):
        gc = hb.bipartite_graph_construction
        g = gc.build_graph(emb_owned, means, sym=False, k=hp["bipartitegraph_sparsity"])
        return g, gc.edge_weights(emb_owned, means, g, norm=False, stat_reduce=sync_stats if gc.training else None)

    return dict(
        node_encode=blk._encode_nodes, edge_encode=blk._encode_edges, ignn_cells=list(blk.ignn_cells),
        emb_head=lambda n: nn.functional.normalize(concat_mlp(blk.output_layer, [(n.float(), None)])),
        cluster=lambda emb, g: hb.clustering(emb, g, return_count=True),
        centroids=centroids,
        super_graph=lambda means: hb.super_graph_construction(means, means, sym=True, norm=True,
                                                              k=hp["supergraph_sparsity"]),
        bipartite=bipartite,
        pool=lambda nodes, bg, bw, S: gather_scale_scatter(nodes, bg[0], bg[1], S, bw, row_scale=l1_row_scale(nodes)),
        supernode_encode=hb._encode_supernodes, superedge_encode=hb._encode_superedges,
        hgnn_cells=list(hb.hgnn_cells),
        head=lambda rows: concat_mlp(model.bipartite_output_layer, [(rows, None)]))
