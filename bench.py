#!/usr/bin/env python3
"""Headline benchmark: edges aggregated / sec on TrackML-1GeV-shaped graphs, latent=256.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--event headline|full_pileup]

One "step" = one pass of the aggregation hot path over one event: the K1
``scatter_add(edges[M,256], graph[1], dim_size=N)`` of the message-passing cell
(reference Modules/gnn_utils.py:50), inputs resident in HBM.  At N = 1 and default flags the event is
the synthetic TrackML-shaped event of SURVEY.md section 8d (N=120,000 hits, E=1,000,000 edges ->
M=2,000,000 directed rows, fp32): the BENCH line.

With --gpus P > 1 (launched by torch.distributed.run, one rank per GPU) the event is node-partitioned
into P phi-wedges (hierarchicalgnn_amd.partition); every rank aggregates the directed edges whose
destination it owns (no data-path collective: K1 is local) and, in the same step, ships the rows of its
boundary hits to the neighbours whose edge update reads them (the one exchange a cell needs between its
node update and its edge update, gnn_utils.py:66-71; RCCL over xGMI, on a side stream).

  --scaling weak   (default) the event grows with P: P x 120k hits / P x 1M edges, per-GPU work fixed;
  --scaling strong ONE fixed event for every P: by default the full-pileup event of BASELINE config 5
                   (480k hits, 4M edges -> 8M rows), or the headline event with --event headline (its
                   N = 1 run is the BENCH line).

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip table)
LATENT = 256
EVENTS = {"headline": (120_000, 1_000_000), "full_pileup": (480_000, 4_000_000)}
N_HITS, N_EDGES = EVENTS["headline"]


def algorithmic_bytes(M: int, N: int, L: int) -> int:
    """src read + one int32 index per row + output write"""
    return 4 * L * M + 4 * M + 4 * L * N


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--latent", type=int, default=LATENT)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--event", default=None, choices=sorted(EVENTS),
                    help="default: headline for weak scaling, full_pileup for strong scaling")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--halo-mode", default="all_gather", choices=["all_gather", "all_to_all"])
    ap.add_argument("--extra", action="store_true", help="also time K2-K6 and a full cell (stderr)")
    return ap.parse_args(argv)


def event_size(scaling: str, event: str | None, world: int):
    """(n_hits, n_edges, event name) of the GLOBAL event"""
    name = event or ("headline" if scaling == "weak" else "full_pileup")
    n, e = EVENTS[name]
    if scaling == "weak":
        return n * world, e * world, name
    return n, e, name


class Workload:
    """what one rank aggregates: its directed rows, the destinations it owns, and (P > 1) the halo
    exchange of its shard.  Built from CPU tensors; `device` only decides where the index tensors live,
    so the sharding logic is exercised by the gloo CPU tests with the same code."""

    def __init__(self, scaling: str, event: str | None, world: int, rank: int, device, halo_mode="all_gather",
                 sizes=None, group=None):
        from hierarchicalgnn_amd import synth
        t0 = time.perf_counter()
        self.scaling, self.world, self.rank = scaling, world, rank
        if sizes is None:
            self.n_hits, self.n_edges, self.event = event_size(scaling, event, world)
        else:                                             # tests: a small event with the same code path
            self.n_hits, self.n_edges = sizes if scaling == "strong" else (sizes[0] * world, sizes[1] * world)
            self.event = f"custom{sizes}"
        x, ei = synth.trackml_event(self.n_hits, self.n_edges, seed=1234)
        self.synth_s = time.perf_counter() - t0
        t1 = time.perf_counter()
        self.shard = self.halo = None
        if world == 1:
            graph = synth.directed(ei)
            self.dst_cpu = graph[1].contiguous()
            self.n_local = self.n_hits
        else:
            from hierarchicalgnn_amd import partition
            self.shard = partition.partition_event(x, ei, world, rank)
            graph = self.shard.local_graph
            self.dst_cpu = None
            self.n_local = self.shard.n_owned
            self.halo = partition.HaloExchange(self.shard, device, mode=halo_mode, group=group)
        self.graph = graph.to(device)
        self.M = int(graph.shape[1])
        self.partition_s = time.perf_counter() - t1

    def gather_stats(self, group=None):
        """per-rank rows / owned hits / halo rows, gathered on every rank"""
        mine = torch.tensor([self.M, self.n_local, self.shard.n_halo if self.shard else 0,
                             sum(self.shard.send_splits) if self.shard else 0], dtype=torch.int64)
        if self.world == 1:
            return mine.view(1, 4)
        import torch.distributed as dist
        dev = self.graph.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        out = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(self.world)]
        dist.all_gather(out, mine.to(dev), group=group)
        return torch.stack([o.cpu() for o in out])


def cpu_baseline(graph_dst_cpu: torch.Tensor, n_hits: int, latent: int):
    """the reference CPU aggregation arithmetic (oracle restatement) on the host cores"""
    from oracle import hgnn_oracle
    g = torch.Generator().manual_seed(1235)
    M = graph_dst_cpu.numel()
    src = torch.randn(M, latent, generator=g)
    hgnn_oracle.scatter_add_cpu_timed(src, graph_dst_cpu, n_hits, reps=1)  # warm-up
    best, _ = hgnn_oracle.scatter_add_cpu_timed(src, graph_dst_cpu, n_hits, reps=5)
    return {
        "value": M / best,
        "unit": "edges/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"full workload (M={M}, N={n_hits}, L={latent} fp32), best of 5 calls of "
                  "zeros(N,L).scatter_add_(0, index, src); "
                  f"{best * 1e3:.1f} ms/call; host nproc={os.cpu_count()}; ATen's CPU scatter_add_ does not "
                  "scale with threads for this shape (8 threads measured faster than 128), so this is a "
                  "baseline, not a tuned CPU implementation",
    }


def load_traffic(workload_key: str):
    """per-launch HBM bytes from the committed rocprofv3 PMC passes (profiles/traffic.json), or None.
    STATIC: measured once under rocprofv3 (separate --pmc passes) and committed; not re-measured by this run."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    try:
        t = json.load(open(path))
        return t.get(workload_key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import hierarchicalgnn_amd as H
    from hierarchicalgnn_amd import _lib
    _lib.load()

    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("HGNN_BENCH_BACKEND", "nccl")  # "gloo": 2-rank rehearsal on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    L = args.latent
    t_setup = time.perf_counter()
    wl = Workload(args.scaling, args.event, world, rank, device, args.halo_mode)
    graph, M, n_local, halo = wl.graph, wl.M, wl.n_local, wl.halo
    gen = torch.Generator(device=device).manual_seed(1235 + rank)
    edges = torch.randn(M, L, device=device, generator=gen)
    nodes = torch.randn(n_local, L, device=device, generator=gen) if halo is not None else None

    plan = H.get_plan(graph[1], n_local)          # first build also loads the code object
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup
    builds = []
    for _ in range(3):                             # warm per-event cost: sort + CSR + work list, host wall, synced
        idx_copy = graph[1].clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        H.GraphPlan(idx_copy, n_local)
        torch.cuda.synchronize()
        builds.append((time.perf_counter() - t0) * 1e3)
    plan_ms = sorted(builds)[1]
    side = torch.cuda.Stream(device) if halo is not None else None

    @torch.no_grad()       # inference-style step: no autograd bookkeeping on the host between launches
    def step():
        if halo is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                halo.exchange(nodes)
        out = H.scatter_add(edges, graph[1], dim=0, dim_size=n_local, plan=plan)
        if halo is not None:
            torch.cuda.current_stream().wait_stream(side)
        return out

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record()
        step()
        ends[i].record()
    barrier()
    elapsed = time.perf_counter() - t0
    ev_ms = sorted(s.elapsed_time(e) for s, e in zip(starts, ends))
    kern_ms = sum(ev_ms) / len(ev_ms)

    stats = torch.tensor([elapsed, float(M)], dtype=torch.float64, device=device)
    if dist is not None:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        total_rows = float(sm[1])
    else:
        total_rows = float(M)
    per_rank = wl.gather_stats()

    # secondary figure: the same aggregation when the model keeps its edges in destination-sorted
    # order (models.InteractionGNNBlock does, once per forward): rows of a list are contiguous
    sorted_ms = None
    if world == 1:
        order = torch.argsort(graph[1], stable=True)
        g_sorted = graph[:, order].contiguous()
        e_sorted = edges[order].contiguous()
        plan_s = H.get_plan(g_sorted[1], n_local)
        assert plan_s.sorted
        for _ in range(args.warmup):
            H.scatter_add(e_sorted, g_sorted[1], dim=0, dim_size=n_local, plan=plan_s)
        ts = []
        for _ in range(args.steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            H.scatter_add(e_sorted, g_sorted[1], dim=0, dim_size=n_local, plan=plan_s)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        sorted_ms = sum(ts) / len(ts)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rows * args.steps / elapsed
        alg = algorithmic_bytes(M, n_local, L)
        achieved = alg / (kern_ms * 1e-3) / 1e9
        rows = per_rank[:, 0].double()
        name = f"trackml_synth[{wl.event}] N={wl.n_hits} E={wl.n_edges} M={int(total_rows)} latent={L} K1 scatter_add"
        res = {
            "metric": "edges aggregated/sec on TrackML-1GeV graphs, latent=256",
            "value": value,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": name,
                "event": wl.event,
                "rows_per_gpu": M,
                "dst_rows_per_gpu": n_local,
                "partition": "single event" if world == 1 else
                f"phi-wedge node partition x{world} of ONE {'fixed' if args.scaling == 'strong' else f'{world}x larger'} "
                f"event; per step: local K1 + {args.halo_mode} halo exchange of the boundary rows x {L} f32 on a "
                "side stream",
                "rows_per_rank": per_rank[:, 0].tolist(),
                "owned_hits_per_rank": per_rank[:, 1].tolist(),
                "halo_rows_per_rank": per_rank[:, 2].tolist(),
                "halo_bytes_received_per_rank": (per_rank[:, 2] * 4 * L).tolist(),
                "halo_bytes_sent_per_rank": (per_rank[:, 3] * 4 * L).tolist(),
                "row_imbalance_max_over_mean": float(rows.max() / rows.mean()),
                "setup_s": {"synth": wl.synth_s, "partition_and_halo_tables": wl.partition_s,
                            "total_before_first_step": setup_s},
                "plan_build_ms": plan_ms,
                "plan_chunk": plan.chunk,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_seg_reduce<64,1,...> (K1 segmented reduce, rank 0)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": alg,
                "avg_launch_ms": kern_ms,
                "median_launch_ms": ev_ms[len(ev_ms) // 2],
                "traffic": load_traffic(f"k1_M{M}_N{n_local}_L{L}"),
                "traffic_source": "static: profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                  "committed with the repo; NOT re-measured by this run)",
            },
            "sorted_layout": None if sorted_ms is None else {
                "note": "same K1 call on destination-sorted edges (the layout models.InteractionGNNBlock "
                        "runs its cells in); NOT the headline value, which keeps the reference's arbitrary edge order",
                "avg_launch_ms": sorted_ms,
                "achieved": (alg - 4 * M) / (sorted_ms * 1e-3) / 1e9,
                "algorithmic_bytes_per_launch": alg - 4 * M,
                "frac": (alg - 4 * M) / (sorted_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "edges_per_s": M / (sorted_ms * 1e-3),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(wl.dst_cpu, wl.n_hits, L)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
