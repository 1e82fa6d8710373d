#!/usr/bin/env python3
"""Headline benchmark: edges aggregated / sec on TrackML-1GeV-shaped graphs, latent=256.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling both|weak|strong] [--event headline|full_pileup]

One "step" = one pass of the aggregation hot path over one event: the K1
``scatter_add(edges[M,256], graph[1], dim_size=N)`` of the message-passing cell
(reference Modules/gnn_utils.py:50), inputs resident in HBM.  At N = 1 and default flags the event is
the synthetic TrackML-shaped event of SURVEY.md section 8d (N=120,000 hits, E=1,000,000 edges ->
M=2,000,000 directed rows, fp32): the BENCH line.

Launch.  ``python bench.py --gpus P`` with P > 1 and no WORLD_SIZE in the environment starts its P ranks
ITSELF: the parent never touches the GPU (no HIP call, no ``torch.cuda`` query), spawns one child process
per rank (fresh interpreters, rendezvous on 127.0.0.1), relays rank 0's JSON line and exits with the worst
child status.  Under ``python -m torch.distributed.run --nproc-per-node P bench.py --gpus P`` (WORLD_SIZE
set) every process is a rank and nothing is spawned.  No process that has initialised the GPU is ever
exec'ed or re-launched.

With P > 1 the event is node-partitioned into P phi-wedges (hierarchicalgnn_amd.partition); a rank
aggregates the directed edges whose destination it owns (no data-path collective: K1 is local) and then
ships the rows of its boundary hits to the neighbours whose edge update reads them -- the ONE exchange a
cell needs, placed where the cell has it: AFTER the node-side result of the step (the rows shipped are rows
of this step's K1 output, standing in for the updated nodes of gnn_utils.py:53) and BEFORE anything
edge-side, on the same stream, so the next step's K1 starts after the exchange has landed (in the cell the
next aggregation depends on the edge update, which depends on the halo: gnn_utils.py:66-71).  K1 cannot hide
the exchange and the bench does not pretend it can; the model path overlaps it with the interior-edge MLP
(partition.distributed_cell_forward), which this K1-only step does not contain.

One line carries BOTH scaling figures (``--scaling both``, the default):
  value / scaling "weak"   the event grows with P (P x 120k hits / P x 1M edges, per-GPU work fixed); N = 1 is
                           the BENCH line;
  "strong_scaling": {...}   ONE fixed event for every P: the full-pileup event of BASELINE config 5 (480k hits,
                           4M edges -> 8M rows), with per-rank rows / halo bytes / imbalance.
``--scaling weak|strong`` runs one of them only (``strong`` then fills ``value``).

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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
    ap.add_argument("--scaling", default="both", choices=["both", "weak", "strong"])
    ap.add_argument("--event", default=None, choices=sorted(EVENTS),
                    help="default: headline for weak scaling, full_pileup for strong scaling")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--halo-mode", default="auto", choices=["auto", "all_gather", "all_to_all"],
                    help="auto: both are timed during warm-up and the faster one (max over ranks) is used")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="launcher rehearsal without a GPU: ranks rendezvous over gloo, shard the event, run the halo "
                         "exchange on CPU tensors and print the sharding statistics; measures nothing")
    ap.add_argument("--sizes", type=int, nargs=2, default=None, metavar=("HITS", "EDGES"),
                    help="per-GPU (weak) / total (strong) event size override (rehearsals and tests)")
    return ap.parse_args(argv)


def event_size(scaling: str, event: str | None, world: int):
    """(n_hits, n_edges, event name) of the GLOBAL event"""
    name = event or ("headline" if scaling == "weak" else "full_pileup")
    n, e = EVENTS[name]
    if scaling == "weak":
        return n * world, e * world, name
    return n, e, name


# ------------------------------------------------------------------------------------------- launcher
def _free_port() -> int:
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(argv, world: int, timeout_s: float = 3000.0, python: str | None = None) -> int:
    """Start `world` fresh rank processes of this script (one per GPU) and wait for them.  The calling process must
    not have initialised the GPU and does not do so here: it only spawns children, relays rank 0's stdout and returns
    the worst exit status.  Children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT."""
    import tempfile
    port = _free_port()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HGNN_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC only on this pool (RCCL needs it)
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // world)))
        cmd = [python or sys.executable, os.path.abspath(__file__)] + list(argv)
        procs.append(subprocess.Popen(cmd, env=env, stdout=out0 if r == 0 else sys.stderr, stderr=None))
    deadline = time.monotonic() + timeout_s
    rc = 0
    failed_at = None
    try:
        while any(p.poll() is None for p in procs):
            now = time.monotonic()
            if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
                failed_at = now              # a rank died: its peers would wait in a collective for ever
            if now > deadline or (failed_at is not None and now - failed_at > 15.0):
                rc = 124 if now > deadline else rc
                break
            time.sleep(0.1)
    finally:
        for p in procs:                      # our own children, by handle: never by pattern
            if p.poll() is None:
                p.kill()
                p.wait()
    out0.seek(0)
    for line in out0.read().splitlines(keepends=True):
        # stdout carries the ONE JSON line of rank 0; whatever a library printed there (gloo's "[Gloo] Rank 0 is connected
        # ..." banner of the CPU rehearsal) goes to stderr
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
    sys.stdout.flush()
    out0.close()
    for p in procs:
        if p.returncode != 0:
            rc = rc or (p.returncode if p.returncode and p.returncode > 0 else 1)
    return rc


# ------------------------------------------------------------------------------------------- workload
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
            self.event = f"custom{tuple(sizes)}"
        x, ei = synth.trackml_event(self.n_hits, self.n_edges, seed=1234)
        self.synth_s = time.perf_counter() - t0
        t1 = time.perf_counter()
        self.shard = None
        self.halos = {}
        if world == 1:
            graph = synth.directed(ei)
            self.dst_cpu = graph[1].contiguous()
            self.n_local = self.n_hits
        else:
            from hierarchicalgnn_amd import partition
            self.shard = partition.partition_event(x, ei, world, rank, device=device)   # on the GPU: device-wide sort / unique / compaction
            graph = self.shard.local_graph
            self.dst_cpu = None
            self.n_local = self.shard.n_owned
            modes = ["all_gather", "all_to_all"] if halo_mode == "auto" else [halo_mode]
            for m in modes:
                self.halos[m] = partition.HaloExchange(self.shard, device, mode=m, group=group)
        self.halo_mode = None if not self.halos else next(iter(self.halos))
        self.graph = graph.to(device)
        self.M = int(graph.shape[1])
        self.partition_s = time.perf_counter() - t1

    @property
    def halo(self):
        return self.halos.get(self.halo_mode)

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


# ------------------------------------------------------------------------------------------- one scaling mode
def _all_max(dist, value: float, device) -> float:
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def measure(args, scaling: str, world: int, rank: int, device, dist, coll_device, with_sorted: bool):
    """run one scaling mode; returns the result dict (meaningful on rank 0)"""
    import hierarchicalgnn_amd as H
    L = args.latent
    t_setup = time.perf_counter()
    wl = Workload(scaling, args.event, world, rank, device, args.halo_mode, sizes=args.sizes)
    graph, M, n_local = wl.graph, wl.M, wl.n_local
    gen = torch.Generator(device=device).manual_seed(1235 + rank)
    edges = torch.randn(M, L, device=device, generator=gen)
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

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the exchange alone (every candidate mode), event-timed on the launching stream; max over ranks picks the mode
    halo_ms = {}
    if wl.halos:
        probe = torch.randn(n_local, L, device=device, generator=gen)
        for m, h in wl.halos.items():
            with torch.no_grad():
                for _ in range(3):
                    h.exchange(probe)
                barrier()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(10):
                    h.exchange(probe)
                b.record()
                barrier()
            halo_ms[m] = _all_max(dist, a.elapsed_time(b) / 10, coll_device)
        wl.halo_mode = min(halo_ms, key=halo_ms.get)
        del probe
    halo = wl.halo

    @torch.no_grad()       # inference-style step: no autograd bookkeeping on the host between launches
    def step(ev_a=None, ev_b=None):
        if ev_a is not None:
            ev_a.record()
        out = H.scatter_add(edges, graph[1], dim=0, dim_size=n_local, plan=plan)     # K1 (gnn_utils.py:50)
        if ev_b is not None:
            ev_b.record()
        if halo is not None:
            # rows of THIS step's node-side result go to the neighbours; same stream: the next step's K1 is ordered
            # behind the landed halo, as the next cell's aggregation is behind this cell's edge update
            halo.exchange(out)
        return out

    for _ in range(args.warmup):
        step()
    barrier()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(starts[i], ends[i])
    barrier()
    elapsed = time.perf_counter() - t0
    ev_ms = sorted(s.elapsed_time(e) for s, e in zip(starts, ends))
    kern_ms = sum(ev_ms) / len(ev_ms)

    stats = torch.tensor([elapsed, float(M)], dtype=torch.float64, device=coll_device)
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
    if with_sorted:
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
        del e_sorted, g_sorted

    alg = algorithmic_bytes(M, n_local, L)
    achieved = alg / (kern_ms * 1e-3) / 1e9
    rows = per_rank[:, 0].double()
    res = {
        "scaling": scaling,
        "value": total_rows * args.steps / elapsed,
        "ms_per_step": elapsed / args.steps * 1e3,
        "workload": f"trackml_synth[{wl.event}] N={wl.n_hits} E={wl.n_edges} M={int(total_rows)} latent={L} "
                    "K1 scatter_add",
        "event": wl.event,
        "rows_per_gpu": M,
        "dst_rows_per_gpu": n_local,
        "partition": "single event" if world == 1 else
        f"phi-wedge node partition x{world} of ONE {'fixed' if scaling == 'strong' else f'{world}x larger'} event; "
        f"per step: local K1, then the {wl.halo_mode} halo exchange of the boundary rows (x {L} f32) of its output "
        "on the same stream (exposed, not hidden under K1)",
        "rows_per_rank": per_rank[:, 0].tolist(),
        "owned_hits_per_rank": per_rank[:, 1].tolist(),
        "halo_rows_per_rank": per_rank[:, 2].tolist(),
        "halo_bytes_received_per_rank": (per_rank[:, 2] * 4 * L).tolist(),
        "halo_bytes_sent_per_rank": (per_rank[:, 3] * 4 * L).tolist(),
        "row_imbalance_max_over_mean": float(rows.max() / rows.mean()),
        "halo_exchange_ms": halo_ms or None,
        "halo_mode": wl.halo_mode,
        "k1_launch_ms": kern_ms,
        "setup_s": {"synth": wl.synth_s, "partition_and_halo_tables": wl.partition_s,
                    "partition_event_only": wl.shard.partition_s if wl.shard is not None else 0.0,
                    "total_before_first_step": setup_s},
        "plan_build_ms": plan_ms,
        "plan_chunk": plan.chunk,
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
        "_dst_cpu": wl.dst_cpu, "_n_hits": wl.n_hits,
    }
    del edges, plan, wl
    torch.cuda.empty_cache()
    return res


def selftest_launch(args, world: int, rank: int):
    """launcher rehearsal (no GPU, nothing measured): rendezvous, shard, exchange on CPU tensors, gather the stats"""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    out = {}
    for scaling in (["weak", "strong"] if args.scaling == "both" else [args.scaling]):
        wl = Workload(scaling, args.event, world, rank, "cpu", "all_gather" if args.halo_mode == "auto" else args.halo_mode,
                      sizes=args.sizes or (3000, 20000))
        if wl.halo is not None:
            g = torch.Generator().manual_seed(7)
            table = torch.randn(wl.n_hits, 4, generator=g)            # same table on every rank
            got = wl.halo.exchange(table[wl.shard.owned_global])
            assert torch.equal(got, table[wl.shard.halo_global]), "halo rows differ from their owners' rows"
        st = wl.gather_stats()
        out[scaling] = {"rows_per_rank": st[:, 0].tolist(), "halo_rows_per_rank": st[:, 2].tolist(),
                        "n_hits": wl.n_hits, "n_edges": wl.n_edges}
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "value": None, "modes": out}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.selftest_launch:
        return selftest_launch(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    n_dev = max(torch.cuda.device_count(), 1)
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from hierarchicalgnn_amd import _lib
    _lib.load()

    dist = None
    backend = None
    coll_device = device
    if world > 1:
        import torch.distributed as dist
        # "gloo": rehearsal of P ranks on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device);
        # the timings of such a run mean nothing, the sharding and the launch path do
        backend = os.environ.get("HGNN_BENCH_BACKEND") or ("nccl" if n_dev >= world else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
            coll_device = torch.device("cpu")
            if args.halo_mode == "auto":
                args.halo_mode = "all_gather"       # gloo has no device all_to_all

    modes = ["weak", "strong"] if args.scaling == "both" else [args.scaling]
    results = {}
    for m in modes:
        # at N = 1 weak == the BENCH line; the sorted-layout figure rides on the first mode only
        results[m] = measure(args, m, world, rank, device, dist, coll_device, with_sorted=(world == 1 and m == modes[0]))

    if rank == 0:
        main_mode = modes[0]
        r = results[main_mode]
        res = {
            "metric": "edges aggregated/sec on TrackML-1GeV graphs, latent=256",
            "value": r["value"],
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": r["ms_per_step"],
            "higher_is_better": True,
            "scaling": main_mode,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {k: r[k] for k in (
                "workload", "event", "rows_per_gpu", "dst_rows_per_gpu", "partition", "rows_per_rank",
                "owned_hits_per_rank", "halo_rows_per_rank", "halo_bytes_received_per_rank",
                "halo_bytes_sent_per_rank", "row_imbalance_max_over_mean", "halo_exchange_ms", "halo_mode",
                "k1_launch_ms", "setup_s", "plan_build_ms", "plan_chunk")},
            "roofline": r["roofline"],
            "sorted_layout": r["sorted_layout"],
        }
        if backend is not None:
            res["config"]["backend"] = backend if backend == "nccl" else \
                f"{backend} (REHEARSAL: fewer GPUs than ranks or forced; timings are not RCCL-over-xGMI numbers)"
        if "strong" in results and main_mode != "strong":
            s = results["strong"]
            res["strong_scaling"] = {
                "note": "ONE fixed event for every N (BASELINE config 5 full-pileup shape); same step as `value`: "
                        "local K1 + exposed halo exchange; divide by this key's N=1 value for the strong-scaling curve",
                "value": s["value"], "unit": "edges/s", "ms_per_step": s["ms_per_step"],
                **{k: s[k] for k in ("workload", "event", "rows_per_rank", "owned_hits_per_rank", "halo_rows_per_rank",
                                     "halo_bytes_received_per_rank", "halo_bytes_sent_per_rank",
                                     "row_imbalance_max_over_mean", "halo_exchange_ms", "halo_mode", "k1_launch_ms",
                                     "setup_s")},
                "roofline": s["roofline"],
            }
        if world == 1 and not args.no_cpu_baseline and r["_dst_cpu"] is not None:
            res["cpu_baseline"] = cpu_baseline(r["_dst_cpu"], r["_n_hits"], args.latent)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's call: python3 bench.py --gpus N.  This process becomes the launcher and stays off the GPU.
        raise SystemExit(launch(sys.argv[1:], args.gpus))
    run_rank(args)


if __name__ == "__main__":
    main()
