#!/usr/bin/env python3
"""bench.py - candidate trees scored per second by the HIP Fitch path on MI355X.

  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the headline metric is quoted on):
synthetic 500-taxon x 50 000-site DNA alignment, SPR neighbourhood; one "step" = one launch of
the scoring kernel over one resident batch of B candidate topologies (edits against the resident
current tree).  Alignment, tree and candidate programs are already in HBM when the timed region
starts.  For N > 1 every rank is an independent restart (own seed, own start tree, own
candidates) on its own GPU; the only collective is an RCCL min-reduce of the best length.

Rank 0 prints ONE JSON line (see README/DESIGN.md for the fields).  `cpu_baseline` times LVB's
own CPU path (the reference compiled into oracle/_ref, or our C restatement if that did not
travel) on the host cores, on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
L2_PATH_PROBE_GBS = 31000.0  # tools/l2_probe.hip on MI355X: the walk's access pattern, loads only
MOVES = {"nni": 0, "spr": 1, "tbr": 2}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--taxa", type=int, default=500)
    ap.add_argument("--sites", type=int, default=50000)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--move", choices=list(MOVES), default="spr")
    ap.add_argument("--nbatches", type=int, default=4, help="distinct resident batches cycled through")
    ap.add_argument("--walk", type=int, default=75, help="accepted random moves applied to the start tree "
                    "before measuring (BASELINE.md: 300 proposals, every 4th accepted)")
    ap.add_argument("--dist", choices=["tree", "uniform"], default="tree",
                    help="synthetic alignment: tree-like (generator T of SURVEY.md 8d) or i.i.d. uniform (U)")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--anneal-seconds", type=float, default=4.0,
                    help="after the timed region, run the batched SA host end to end for this long and report "
                         "best-length-vs-wallclock (0 = skip)")
    ap.add_argument("--anneal-batch", type=int, default=4096, help="ceiling of the SA step size (it adapts)")
    ap.add_argument("--e2e-steps", type=int, default=40,
                    help="after the timed region: steps of lvbgpu_propose_score (neighbours drawn, programmed and "
                         "scored on the GPU, lengths back on the host) to report the end-to-end rate (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def synth_rows(n: int, m: int, seed: int, dist: str = "tree") -> list[bytes]:
    """Generator T of SURVEY.md 8(d): taxon 0 uniform, taxon i copies taxon (i-1)//2 with 10 % substitutions;
    generator U: every cell i.i.d. uniform over ACGT."""
    from tests.synth import treelike_rows, uniform_rows
    return treelike_rows(n, m, seed) if dist == "tree" else uniform_rows(n, m, seed)


def cpu_all_cores(taxa: int, sites_seed, kind: int, seconds: float):
    """One independent reference chain per host core, at the same time (separate processes: the
    reference is non-reentrant).  -> aggregate trees/s and the core count used."""
    import subprocess
    sites, seed = sites_seed
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 32))
    cmd = [sys.executable, "-m", "oracle.cpu_bench", "--taxa", str(taxa), "--sites", str(sites), "--seed", str(seed),
           "--kind", str(kind), "--seconds", str(seconds)]
    procs = [subprocess.Popen(cmd + ["--chain", str(c)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                              text=True) for c in range(cores)]
    rate = 0.0
    done = 0
    for p in procs:
        try:
            out, _ = p.communicate(timeout=seconds * 6 + 120)
            d = json.loads(out.strip().splitlines()[-1])
            rate += d["reps"] / (d["t_getplen"] + d["t_mutate"])
            done += 1
        except Exception:
            p.kill()
    return {"value": rate, "unit": "trees/s", "cores": done,
            "sample": f"{done} concurrent single-threaded reference chains, {seconds:.0f} s each"}


def cpu_baseline(rows, kind: int, budget_s: float, spot, args_sites_seed=None):
    """LVB's CPU path on this host: reference mutate_* + getplen (incl. its per-proposal treecopy)."""
    from oracle import binding as ob
    cores = 1
    if ob.load_ref() is not None:
        rr = ob.RefRun(rows=rows, seed=12345, nproc=1)
        try:
            rr.getplen(0)
            # calibrate on a few proposals, then size the sample to the budget
            tg, tm, _, _ = rr.time_proposals(kind, 20, 4)
            per = max((tg + tm) / 20, 1e-6)
            reps = int(max(50, min(20000, budget_s / per)))
            tg, tm, cs, ds = rr.time_proposals(kind, reps, 4)
            tfull, _ = rr.time_full(3)
            # spot parity: the reference's full evaluation of the bench's own start tree
            check = None
            if spot is not None:
                left, right, root, length = spot
                ot = ob.OracleTree(rr.n, rr.nwords, rr.enc())
                from tests.helpers import parents_of
                l64, r64 = np.asarray(left, np.int64), np.asarray(right, np.int64)
                ot.set_topology(parents_of(l64, r64), l64, r64, root)
                check = bool(ot.getplen() == length)
            all_cores = cpu_all_cores(len(rows), args_sites_seed, kind, min(budget_s, 8.0))
            return {
                "value": reps / (tg + tm), "unit": "trees/s", "cores": cores, "kind": "reference",
                "all_cores": all_cores,
                "sample": f"{reps} {['NNI', 'SPR', 'TBR'][kind]} proposals (mutate incl. treecopy + incremental "
                          f"getplen, serial branch), every 4th accepted, same alignment; getplen alone "
                          f"{reps / tg:.0f}/s ({1e3 * tg / reps:.3f} ms), mutate {1e3 * tm / reps:.3f} ms, "
                          f"full getplen {1e3 * tfull / 3:.2f} ms, mean dirty {ds / reps:.1f}",
                "getplen_only_trees_per_s": reps / tg,
                "start_tree_length_matches_cpu": check,
            }
        finally:
            rr.close()
    # fallback: our C restatement (the reference did not travel)
    from lvb_amd import host
    enc = ob.encode_rows(rows)
    n, nwords = enc.shape
    tree = host.HostTree(n, seed=12345)
    _, left, right = tree.arrays()
    from tests.helpers import apply_edits, parents_of
    cur = ob.OracleTree(n, nwords, enc)
    l64, r64 = left.astype(np.int64), right.astype(np.int64)
    cur.set_topology(parents_of(l64, r64), l64, r64, tree.root)
    cur.getplen()
    prop = ob.OracleTree(n, nwords)
    t_total, reps = 0.0, 0
    while t_total < budget_s and reps < 20000:
        edits = tree.propose(kind)
        prog = tree.program(mode=0, edits=edits)
        nl, nr = apply_edits(left, right, edits)
        par = parents_of(nl, nr)
        t0 = time.perf_counter()
        prop.copy_from(cur)  # the reference's per-proposal treecopy
        prop.set_topology(par, nl, nr, tree.root)
        prop.mark_dirty([d for d in prog["dsts"] if d >= 0])
        prop.getplen()
        t_total += time.perf_counter() - t0
        reps += 1
    return {"value": reps / t_total, "unit": "trees/s", "cores": 1, "kind": "port",
            "sample": f"{reps} proposals through oracle/fitch_oracle.c (treecopy + incremental getplen)"}


def main():
    args = parse_args()
    from lvb_amd.launch import Ranks
    ranks = Ranks()                       # torch.distributed (nccl = RCCL) only when WORLD_SIZE > 1
    rank, world, local_rank = ranks.rank, ranks.world, ranks.local_rank
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from lvb_amd import api, host

    kind = MOVES[args.move]
    t_setup = time.perf_counter()
    rows, min_len = host.prepare_alignment(synth_rows(args.taxa, args.sites, args.seed, args.dist))
    ctx = api.FitchContext(text_rows=rows, device=ranks.device)    # encode on the device
    tree = host.HostTree(args.taxa, seed=ranks.restart_seed(args.seed))  # each rank: its own restart
    length = tree.upload(ctx)
    for _ in range(args.walk):                                      # short random walk, as BASELINE.md
        e = tree.propose(kind)
        length = ctx.commit(e)
        tree.apply(e)
    spot = (tree.arrays()[1].copy(), tree.arrays()[2].copy(), tree.root, length)

    batches = []
    for b in range(args.nbatches):
        offs, edits = tree.propose_batch(kind, args.batch)
        bh = api.C.c_void_p()
        ctx._chk(ctx.lib.lvbgpu_batch_build(ctx.h, args.batch, offs, edits.ctypes.data, None, api.C.byref(bh)))
        batches.append(api.Batch(ctx, bh, args.batch))
    stats = [b.stats() for b in batches]
    setup_s = time.perf_counter() - t_setup

    own_comm = False
    if world > 1:
        # RCCL communicator of the scoring library itself (not torch's).  Its init is collective, so the
        # ranks first agree (through torch) that every one of them can take part: a rank that cannot must not
        # leave the others waiting inside ncclCommInitRank.
        if ranks.sum_over_ranks(int(api.comm_available())) == world:
            uid = None
            if rank == 0:
                try:
                    uid = api.comm_unique_id()
                except api.LvbGpuError as exc:
                    print(f"[rank 0] lvbgpu_comm_unique_id failed ({exc})", file=sys.stderr)
            uid = ranks.share_bytes(uid)           # None from rank 0: every rank skips the library's communicator
            if uid is not None:
                try:
                    ctx.comm_init(world, rank, uid)
                    own_comm = True
                except api.LvbGpuError as exc:   # keep the scaling run alive: same reduction through torch's RCCL
                    print(f"[rank {rank}] lvbgpu_comm_init failed ({exc}); min-reduce falls back to torch.distributed",
                          file=sys.stderr)
        own_comm = bool(ranks.sum_over_ranks(int(own_comm)) == world)

    def barrier():
        ctx.synchronize()
        ranks.barrier()

    for b in batches:          # setup, not warm-up: every resident batch is scored once, so that any --warmup /
        b.launch()             # --steps (even 0 / 1) reads back lengths that exist
    for i in range(args.warmup):
        batches[i % len(batches)].launch()
    for b in batches:          # the first read-back after a run of launches costs the runtime several milliseconds
        b.lengths()            # once (pinned-buffer mapping, signal pool): not a per-step cost, so it happens here
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for i in range(args.steps):
        batches[i % len(batches)].launch()
    kernel_ms = ctx.timer_stop()  # HIP events on the stream the kernels run on
    t_gpu_done = time.perf_counter()
    best_local = min(int(b.lengths().min()) for b in batches)
    best_global = best_local
    if world > 1:
        if own_comm:
            best_global, _ = ctx.allreduce_min(best_local)
        else:
            best_global = -ranks.max_over_ranks(-float(best_local))
            best_global = int(best_global)
    t_reduced = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    print(f"[rank {rank}] timed region: launches+sync {1e3 * (t_gpu_done - t0):.2f} ms (events {kernel_ms:.2f} ms), "
          f"lengths+min-reduce {1e3 * (t_reduced - t_gpu_done):.2f} ms, barrier {1e3 * (t0 + elapsed - t_reduced):.2f} ms",
          file=sys.stderr)

    elapsed = ranks.max_over_ranks(elapsed)   # the slowest rank defines the step time

    # per-launch accounting for the roofline: average over the batches actually launched
    launched = [stats[i % len(batches)] for i in range(args.steps)]
    alg_bytes = float(np.mean([s["algorithmic_bytes"] for s in launched]))
    mean_dirty = float(np.mean([s["dirty_nodes"] / s["candidates"] for s in launched]))
    launch_ms = kernel_ms / args.steps
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9

    traffic = None
    tfile = ROOT / "profiles" / "traffic.json"
    if tfile.exists():
        t = json.loads(tfile.read_text())
        if (t.get("taxa"), t.get("sites"), t.get("batch"), t.get("move")) == (args.taxa, args.sites, args.batch, args.move):
            traffic = t["hbm_bytes_per_launch"]   # rocprofv3 PMC passes (profiles/collect.sh), gfx950-corrected

    total_trees = args.batch * args.steps * world
    out = {
        "metric": "candidate trees scored/sec (Fitch getplen), 500 taxa x 50k sites",
        "value": total_trees / elapsed,
        "unit": "trees/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.taxa} taxa x {args.sites} sites synthetic DNA "
                        f"({'tree-like, 10% substitutions' if args.dist == 'tree' else 'i.i.d. uniform'}), "
                        f"{args.move.upper()} neighbourhood, incremental getplen semantics, "
                        f"B={args.batch} candidates per step, {args.nbatches} resident batches cycled",
            "taxa": args.taxa, "sites_after_constant_cut": len(rows[0]), "nwords": ctx.nwords,
            "batch": args.batch, "move": args.move, "mean_dirty_nodes": round(mean_dirty, 2),
            "parallelism": f"{world} independent restart(s), one per GPU; RCCL min-reduce of best length"
                           + ("" if world == 1 else (" (lvbgpu_allreduce_min)" if own_comm else " (torch fallback)")),
            "best_length": best_global, "min_len_tree": min_len, "setup_seconds": round(setup_s, 2),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": "lvbgpu::fitch_walk<false, false>", "launch_ms": launch_ms,
            "algorithmic_bytes_per_launch": alg_bytes,
            "note": "algorithmic bytes = (D+3) clean rows x nwords x 8 per candidate; rows are re-read from "
                    "the XCD L2 / Infinity Cache, so achieved may exceed the HBM figure",
            # the ceiling that does bound this kernel: what a pure-load probe with the same access pattern reads
            # through the L2 -> CU path on this chip (tools/l2_probe.hip)
            "cache_path": {"ceiling": L2_PATH_PROBE_GBS, "unit": "GB/s", "frac": achieved / L2_PATH_PROBE_GBS,
                           "source": "tools/l2_probe.hip measured on MI355X (not re-measured in this run)"},
        },
    }
    if args.e2e_steps > 0:
        # nothing resident but the tree: every step draws B fresh neighbours on the device
        ctx.propose_score(args.batch, kind, 1)
        ctx.synchronize()
        te = time.perf_counter()
        for i in range(args.e2e_steps):
            ctx.propose_score(args.batch, kind, 1000 + i)
        te = time.perf_counter() - te
        out["end_to_end"] = {
            "value": args.batch * args.e2e_steps / te, "unit": "trees/s", "steps": args.e2e_steps,
            "what": "lvbgpu_propose_score per step: draw B neighbours + build their programs + score on the GPU, "
                    "lengths and move descriptors copied back to the host (not part of `value`)",
        }
    if args.anneal_seconds > 0:
        # second half of the metric: best length vs wall clock, whole host loop included
        # (proposal generation, program build, H2D, kernels, D2H, accept/commit) - not part of `value`
        p = host.anneal_defaults()
        p.seed = args.seed * 7919 + rank + 1
        p.algorithm = {"nni": 10, "spr": 11, "tbr": 12}[args.move]
        p.batch = args.anneal_batch
        p.t0 = 0.0   # estimated as StartingTemperature() does (65 % of uphill moves accepted)
        p.min_len_tree = min_len
        p.max_seconds = args.anneal_seconds
        p.log_cap = 4096
        res, log = host.anneal(ctx, tree, p)
        keep = log[:: max(1, len(log) // 12)] + log[-1:]
        out["anneal"] = {
            "seconds": round(res["seconds"], 3), "start_length": res["start_length"],
            "best_length": res["best_length"], "scored": res["scored"], "consumed": res["consumed"],
            "accepted": res["accepted"], "device_steps": res["device_steps"],
            "scored_per_s": round(res["scored"] / res["seconds"]), "consumed_per_s": round(res["consumed"] / res["seconds"]),
            "device_fraction": round(res["seconds_device"] / res["seconds"], 3), "batch": args.anneal_batch,
            "t_final": res["t_final"], "temperatures": res["temperatures"], "frozen": res["frozen"],
            "best_length_vs_wallclock": [[round(t, 3), b] for t, b in keep],
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.dist == "tree":
        out["cpu_baseline"] = cpu_baseline(rows, kind, args.cpu_seconds, spot, (args.sites, args.seed))
    for b in batches:
        b.free()
    tree.close()
    ctx.close()
    ranks.close()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
