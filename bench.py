#!/usr/bin/env python3
"""Headline benchmark: edges aggregated / sec on TrackML-1GeV-shaped graphs, latent=256.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the aggregation hot path over one event: the K1
``scatter_add(edges[M,256], graph[1], dim_size=N)`` of the message-passing cell
(reference Modules/gnn_utils.py:50) on the synthetic TrackML-shaped event of
SURVEY.md section 8d (N=120,000 hits, E=1,000,000 edges -> M=2,000,000 directed rows,
fp32), inputs resident in HBM.  With --gpus P > 1 (launched by torch.distributed.run,
one rank per GPU) the event is P times larger and node-partitioned into P
phi-wedges; every rank aggregates the edges whose destination it owns and, in
the same step, exchanges the boundary-node rows its neighbours' edge update
needs (RCCL over xGMI, on a side stream) -- weak scaling.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` and
`cpu_baseline` objects.
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
N_HITS = 120_000
N_EDGES = 1_000_000


def algorithmic_bytes(M: int, N: int, L: int) -> int:
    """src read + one int32 index per row + output write"""
    return 4 * L * M + 4 * M + 4 * L * N


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--latent", type=int, default=LATENT)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--halo-mode", default="all_gather", choices=["all_gather", "all_to_all"])
    ap.add_argument("--extra", action="store_true", help="also time K2-K6 and a full cell (stderr)")
    return ap.parse_args()


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
                  f"{best * 1e3:.1f} ms/call; host nproc={os.cpu_count()}",
    }


def load_traffic(workload_key: str):
    """per-launch HBM bytes from the committed rocprofv3 PMC passes (profiles/), or None"""
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
    from hierarchicalgnn_amd import synth, _lib
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
    halo = None
    if world == 1:
        x, ei = synth.trackml_event(N_HITS, N_EDGES, seed=1234)
        graph = synth.directed(ei)
        dst_cpu = graph[1].contiguous()
        graph = graph.to(device)
        n_local = N_HITS
    else:
        from hierarchicalgnn_amd import partition
        x, ei = synth.trackml_event(N_HITS * world, N_EDGES * world, seed=1234)
        shard = partition.partition_event(x, ei, world, rank)
        graph = shard.local_graph.to(device)
        n_local = shard.n_owned
        halo = partition.HaloExchange(shard, device, mode=args.halo_mode)
        dst_cpu = None
    M = int(graph.shape[1])
    gen = torch.Generator(device=device).manual_seed(1235 + rank)
    edges = torch.randn(M, L, device=device, generator=gen)
    nodes = torch.randn(n_local, L, device=device, generator=gen) if halo is not None else None

    plan = H.get_plan(graph[1], n_local)          # first build also loads the code object
    torch.cuda.synchronize()
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
        wl = f"trackml_synth N={N_HITS * world} E={N_EDGES * world} M={int(total_rows)} latent={L} K1 scatter_add"
        res = {
            "metric": "edges aggregated/sec on TrackML-1GeV graphs, latent=256",
            "value": value,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl,
                "rows_per_gpu": M,
                "dst_rows_per_gpu": n_local,
                "partition": "single event" if world == 1 else
                f"phi-wedge node partition x{world}; per step: local K1 + {args.halo_mode} halo exchange of "
                f"{halo.shard.n_halo} boundary rows x {L} f32 on a side stream (rank 0)",
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
            res["cpu_baseline"] = cpu_baseline(dst_cpu, N_HITS, L)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
