#!/usr/bin/env python3
"""bench.py - candidate trees scored per second by the HIP Fitch path on MI355X.

  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the headline metric is quoted on): synthetic
500-taxon x 50 000-site DNA alignment, SPR neighbourhood, B = 4096 candidates per step.

One "step" is SURVEY.md 8(d)(i)'s unit: batch submit -> lengths on the host.  Per step the library draws B
random neighbours of the resident tree, builds their programs and scores them on the GPU, and the B lengths
come back to the host; two steps are in flight (`lvbgpu_chains_submit` / `_collect`: while the host reads step
i's lengths, step i + 1 is on the device).  Alignment and current tree are in HBM when the timed region starts;
nothing else is.  `value` = candidates of all ranks / wall time of K steps.

Beside it, in the same JSON line:
  one_at_a_time the same step with ONE batch in flight (`lvbgpu_propose_score`): a step's latency
  kernel_only   the scoring kernel alone, replaying resident pre-built batches (round 1's headline)
  roofline      the dominant kernel (fitch_walk<false, false, 0>) against the L2 -> CU path that bounds it: duration
                from HIP events around every 4th walk of the timed region, ceiling from the guide (34.5 TB/s) and
                from a pure-load probe with the walk's access pattern run in this process; `hbm` = measured HBM
                traffic (rocprofv3 PMC, profiles/traffic.json) over the same duration against 8 TB/s
  mixed_walk    the same measurements on a tree mixed by >= 3000 accepted moves (longer dirty paths), with the
                CPU reference timed on that very tree
  cpu_baseline  LVB's own CPU path (the compiled reference, oracle/_ref) on the SAME tree and neighbourhood
                as the headline leg, one core and all cores
  shapes        B = 256 / 1024 / 16 384 and the i.i.d.-uniform alignment (SURVEY.md 8d "U")
  anneal        best length against the wall clock: 32 chains annealing on the GPU (starting temperatures included),
                one chain alone, and the reference PROGRAM annealing the same alignment for a bounded time

For N > 1 every rank is an independent restart (own seed, own start tree, own candidates) on its own GPU; the
only collective is the min-reduce of the best length (`lvbgpu_allreduce_min`, RCCL).  Started without
WORLD_SIZE, `--gpus N` spawns the N ranks itself (this process never touches a GPU then).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
L2_PEAK_GBS = 34500.0   # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate
MOVES = {"nni": 0, "spr": 1, "tbr": 2}
KERNEL = "lvbgpu::fitch_walk<false, false, 0>"   # <COMMIT, WIDE, HANDOVER>: the plain scoring walk
if (os.environ.get("LVBGPU_PAIR", "0") or "0").isdigit() and int(os.environ.get("LVBGPU_PAIR", "0") or 0) > 0:   # an A/B run with two candidates per wave (off by default)
    KERNEL = "lvbgpu::fitch_walk_pair<false, false>"
WALK_TIMING_EVERY = 4   # HIP events around every 4th scoring walk of the timed region


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--taxa", type=int, default=500)
    ap.add_argument("--sites", type=int, default=50000)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--move", choices=list(MOVES), default="spr")
    ap.add_argument("--nbatches", type=int, default=4, help="kernel_only leg: distinct resident batches cycled through")
    ap.add_argument("--walk", type=int, default=75, help="accepted random moves applied to the start tree "
                    "before the headline leg (BASELINE.md: 300 proposals, every 4th accepted)")
    ap.add_argument("--mixed-walk", type=int, default=3000, help="accepted random moves behind the mixed_walk leg "
                    "(0 = skip it)")
    ap.add_argument("--dist", choices=["tree", "uniform"], default="tree",
                    help="synthetic alignment: tree-like (generator T of SURVEY.md 8d) or i.i.d. uniform (U)")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--settle", type=int, default=500, help="untimed steps run during setup, before the warm-up "
                    "(one-off runtime stalls of a fresh process fall in its first ~350 steps)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="budget of each cpu_baseline timing")
    ap.add_argument("--anneal-seconds", type=float, default=4.0,
                    help="run the batched SA host end to end for this long and report best-length-vs-wallclock (0 = skip)")
    ap.add_argument("--anneal-batch", type=int, default=4096, help="ceiling of the SA step size (it adapts)")
    ap.add_argument("--anneal-chains", type=int, default=32, help="independent chains stepped together on the GPU")
    ap.add_argument("--anneal-seed", type=int, default=9,
                    help="seed base of the annealing leg's chains and start trees.  One chain in some dozens has its starting "
                         "temperature land on the reference's second 1e-5 increment and needs ~100x longer to freeze (faithful to "
                         "StartingTemperature.c; DESIGN.md 7c); the default base gives 32 chains without such a straggler, as round "
                         "2's did (tools/anneal_seed_scan.py), so that `scored_per_s` - measured until ALL chains have frozen - says "
                         "something about the scorer.  `scored_per_s_busy` does so for any seed")
    ap.add_argument("--anneal-second-seed", type=int, default=4, help="a second seed base for the annealing leg (-1: none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shapes", action="store_true", help="skip the B = 256 / 1024 / 16 384 and uniform-alignment legs")
    ap.add_argument("--single-chain-levels", type=int, default=3,
                    help="run_levels of the single-chain annealing leg (0: device-drawn throughout, as a chain among others)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip BASELINE.json's other single-GPU configurations (cfg2 64 x 10k NNI, cfg5 2000 x 200k TBR)")
    ap.add_argument("--headline-only", action="store_true",
                    help="the timed region and its roofline only (profiling runs: every scoring walk of the process is "
                         "then one of the headline's device-built batches)")
    ap.add_argument("--long-steps", type=int, default=200,
                    help="steps of the same pipelined loop run BEHIND the timed region and reported as value_long (0: none)")
    ap.add_argument("--comm-init-seconds", type=float, default=90.0,
                    help="N > 1: how long a rank waits for the library's own RCCL communicator (lvbgpu_comm_init) before "
                         "every rank falls back to torch.distributed for the min-reduce")
    ap.add_argument("--dry-stuck-rank", type=int, default=-1, help="--dry-ranks: that rank's communicator set-up never returns")
    ap.add_argument("--dry-ranks", action="store_true",
                    help="rehearse the N-rank flow without GPUs: ranks are spawned, meet over gloo, reduce a fake best "
                         "length; no scoring (CPU test of the launch plumbing)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------- launching N ranks

def spawn_ranks(args) -> int:
    """--gpus N without WORLD_SIZE: start the N ranks as child processes.  This process makes no GPU call
    (a process that has touched the GPU must not start others on this pool); it relays rank 0's line."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, errs = [], []
    logdir = Path(tempfile.mkdtemp(prefix="lvb_bench_ranks_"))
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0's stderr is this process's; the other ranks' goes to a file each, relayed if the run fails
        errs.append(None if r == 0 else open(logdir / f"rank{r}.stderr", "w+"))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[r], text=True))
    out0, failed = "", []
    deadline = time.time() + 3000
    try:
        out0, _ = procs[0].communicate(timeout=max(1.0, deadline - time.time()))
        for r, p in enumerate(procs):
            if p.wait(timeout=max(1.0, deadline - time.time())) != 0:
                failed.append(r)
    except subprocess.TimeoutExpired:
        failed.append(-1)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if failed:
        print(f"bench.py: rank(s) {failed} failed", file=sys.stderr)
        for r, f in enumerate(errs):
            if f is not None:
                f.seek(0)
                tail = f.read()[-4000:]
                if tail.strip():
                    print(f"---- stderr of rank {r} (tail)\n{tail}", file=sys.stderr)
        sys.stdout.write(out0)
    for f in errs:
        if f is not None:
            f.close()
    import shutil
    shutil.rmtree(logdir, ignore_errors=True)
    if failed:
        return 1
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if not lines:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        return 1
    print(lines[-1])
    return 0


def timed_steps(run, synchronize, ranks, steps: int, seed0: int, reduce_best=None, warm_best: int = 1 << 40, before_t0=None):
    """The timed region's skeleton, shared by the GPU flow (submit_to_lengths) and its CPU rehearsal (--dry-ranks):

        [untimed: one min-reduce + barrier]  t0  K steps  min-reduce of the best length  synchronize  barrier  t1

    The untimed reduce is there because a communicator sets itself up lazily in its FIRST collective (RCCL: channel
    and proxy set-up; likewise the first torch barrier after it): at N > 1 that would otherwise land in a timed region
    that is a few milliseconds long and be read as a scaling loss.  The in-region reduce is timed on its own
    (`reduce_ms`); `order` records what ran in which order (tests/test_bench_launch.py checks it)."""
    order = []
    if reduce_best is not None:
        reduce_best(int(warm_best))
        order.append("reduce:warmup")
    synchronize()
    ranks.barrier()
    order.append("barrier")
    if before_t0:
        before_t0()
    t0 = time.perf_counter()
    order.append("t0")
    best = run(steps, seed0)
    t_steps = time.perf_counter() - t0
    order.append("steps")
    tr = time.perf_counter()
    best_global = reduce_best(best) if reduce_best is not None else best
    reduce_ms = 1e3 * (time.perf_counter() - tr)
    order.append("reduce:timed")
    synchronize()
    ranks.barrier()
    elapsed = time.perf_counter() - t0
    order.append("t1")
    return {"elapsed_local": elapsed, "elapsed_s": ranks.max_over_ranks(elapsed), "steps_s_local": t_steps,
            "reduce_ms": reduce_ms if reduce_best is not None else 0.0, "best": best_global, "best_local": best, "order": order}


def dry_rank_main(args) -> None:
    """The rank flow of rank_main with the GPU taken out: rendezvous, the timed region's skeleton (timed_steps: warm-up
    reduce, barriers, max-over-ranks timing) and a min-reduce of a stand-in best length, all over gloo."""
    from lvb_amd.launch import Ranks, call_with_deadline, pin_to_share_of_cores
    host_share = pin_to_share_of_cores()
    ranks = Ranks(backend="gloo")
    if ranks.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ranks.world}")
    reduces = []
    # the library's communicator set-up, rehearsed: a collective call under a deadline; --dry-stuck-rank R: rank R's never
    # returns - every rank must then agree on the fallback and the run must still end
    own_comm, stuck = False, False
    if ranks.world > 1:
        def stand_in():
            if ranks.rank == args.dry_stuck_rank:
                time.sleep(3600)
            return True
        done, _ = call_with_deadline(stand_in, args.comm_init_seconds if args.dry_stuck_rank < 0 else 1.0)
        stuck = not done
        own_comm = bool(ranks.sum_over_ranks(int(done)) == ranks.world)

    def reduce_best(best_local: int) -> int:
        reduces.append(int(best_local))
        return int(-ranks.max_over_ranks(-float(best_local)))

    def run(n, seed):   # stands in for n device steps; the minimum sits on the last rank
        return 1000 + 7 * ((ranks.rank + 1) % ranks.world)

    head = timed_steps(run, lambda: None, ranks, args.steps, 1000, reduce_best if ranks.world > 1 else None)
    per_rank = ranks.all_values(float(ranks.rank + 1))
    seeds = ranks.sum_over_ranks(ranks.restart_seed(args.seed))
    cores_of_ranks = ranks.all_values(float(host_share["cores_per_rank"]))
    if ranks.rank == 0:
        print(json.dumps({
            "metric": "candidate trees scored/sec (Fitch getplen), 500 taxa x 50k sites", "value": 0.0,
            "unit": "trees/s", "n_gpus": ranks.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * head["elapsed_s"] / max(args.steps, 1), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic", "dry": True,
            "config": {"workload": "dry run of the rank flow: no GPU, nothing scored",
                       "parallelism": f"{ranks.world} ranks over gloo (on GPUs: one independent restart per GPU, "
                                      "best length min-reduced by lvbgpu_allreduce_min over RCCL)",
                       "best_length": int(head["best"]), "seed_sum": seeds, "reduce_ms": head["reduce_ms"],
                       "order": head["order"], "reduces": len(reduces), "per_rank": per_rank,
                       "reducer": "gloo (dry)", "comm_size": ranks.world, "own_comm": own_comm,
                       "cores_per_rank": host_share["cores_per_rank"], "host_threads_per_rank": host_share["threads"],
                       "pinned_to_share_of_cores": host_share["pinned"],
                       "cores_of_ranks": cores_of_ranks}}), flush=True)
    ranks.close()
    if stuck:
        sys.stdout.flush()
        os._exit(0)


# ----------------------------------------------------------------------------------------- workload pieces

def synth_rows(n: int, m: int, seed: int, dist: str = "tree") -> list[bytes]:
    """Generator T of SURVEY.md 8(d): taxon 0 uniform, taxon i copies taxon (i-1)//2 with 10 % substitutions;
    generator U: every cell i.i.d. uniform over ACGT."""
    from tests.synth import treelike_rows, uniform_rows
    return treelike_rows(n, m, seed) if dist == "tree" else uniform_rows(n, m, seed)


def random_walk(ctx, tree, kind: int, moves: int) -> int:
    """`moves` accepted random moves on the resident tree (host generator, committed on the device)."""
    length = ctx.current_length()
    for _ in range(moves):
        e = tree.propose(kind)
        length = ctx.commit(e)
        tree.apply(e)
    return length


def submit_to_lengths(ctx, ranks, B: int, kind: int, steps: int, warmup: int, seed0: int, reduce_best=None, settle: int = 0,
                      depth: int = 2, long_steps: int = 0):
    """The metric's step, `steps` times: draw + program + score B neighbours on the GPU, lengths on the host.
    -> dict(elapsed_s [max over ranks], launch_ms [mean walk duration, HIP events], best, stats).
    depth 2 (default): two batches in flight (lvbgpu_chains_submit / _collect) - while the host reads step i's
    lengths, step i+1 is already on the device and step i+2's neighbours are being drawn beside its walk; every step
    still ends with its B lengths on the host.  depth 1: one batch at a time (lvbgpu_propose_score), a step's latency.
    settle: untimed steps run as part of the SETUP, before the `warmup` steps: a fresh process pays two one-off
    runtime stalls of 1 and 8 ms somewhere in its first ~350 steps (tools/stall_probe.py: queue resources being
    grown, never again afterwards), which would otherwise land in a timed region of a few milliseconds."""
    from lvb_amd import api
    lib, h = ctx.lib, ctx.h
    draw = np.zeros(1, dtype=api.DRAW_DTYPE)
    draw[0]["chain"], draw[0]["count"], draw[0]["kind"] = 0, B, kind
    outs = [np.zeros(B, dtype=np.int64), np.zeros(B, dtype=np.int64)]

    def submit(slot, seed):
        draw[0]["seed"] = seed & 0xFFFFFFFFFFFFFFFF
        ctx._chk(lib.lvbgpu_chains_submit(h, slot, 1, draw.ctypes.data))

    stamps = []   # when each step of the last run() had its lengths on the host

    def run(n, first_seed):
        """n pipelined steps; -> best length seen"""
        best = np.iinfo(np.int64).max
        del stamps[:]
        stamps.append(time.perf_counter())
        for j in range(min(depth, n)):
            submit(j % 2, first_seed + j)
        for i in range(n):
            slot = i % 2 if depth > 1 else 0
            ctx._chk(lib.lvbgpu_chains_collect(h, slot, outs[slot]))
            stamps.append(time.perf_counter())
            best = min(best, int(outs[slot].min()))
            if i + depth < n:
                submit(slot, first_seed + i + depth)
        return best

    run(settle, (seed0 - 100000) & 0x7FFFFFFF)
    warm_best = run(warmup, (seed0 - 1000) & 0x7FFFFFFF)
    gc_was = gc.isenabled()

    def before_t0():
        ctx.walk_timing(WALK_TIMING_EVERY)   # a pair of events costs the step ~20 us: sample
        gc.disable()                         # a collection inside a few-millisecond region would be most of it

    paired_before = ctx.paired_walks()
    head = timed_steps(run, ctx.synchronize, ranks, steps, seed0, reduce_best,
                       warm_best if warmup > 0 else 1 << 40, before_t0)
    step_gaps = np.diff(np.array(stamps)) if len(stamps) > 1 else np.zeros(1)
    if gc_was:
        gc.enable()
    best, best_global, t_steps, elapsed = head["best_local"], head["best"], head["steps_s_local"], head["elapsed_local"]
    walk_ms, walks = ctx.walk_timing_read()
    ctx.walk_timing(False)
    paired = ctx.paired_walks() > paired_before   # the library pairs by itself where the programs are long
    # a longer look at the same loop, behind the timed region (a region of K = 20 steps is 2 ms: one stalled step halves
    # its figure, and nothing in K steps says whether one did): `long_steps` more steps on this rank's own clock
    long_rate, long_n = None, 0
    if long_steps > 0:
        tl = time.perf_counter()
        run(long_steps, (seed0 + 50000) & 0x7FFFFFFF)
        ctx.synchronize()
        long_n, long_rate = long_steps, time.perf_counter() - tl
    # what those batches cost: the draw is a function of (seed, b), so re-drawing a few of the timed seeds
    # (outside the timed region) gives exactly their counts
    picks = sorted({seed0 + int(round(k * (steps - 1) / 7)) for k in range(8)}) if steps > 0 else []
    st = []
    for s in picks:
        ctx.propose_score(B, kind, s)
        st.append(ctx.proposal_stats())
    mean = lambda key: float(np.mean([x[key] for x in st])) if st else 0.0
    return {
        "elapsed_s": head["elapsed_s"], "elapsed_local": elapsed, "steps_s_local": t_steps, "reduce_ms": head["reduce_ms"],
        "order": head["order"], "launch_ms": walk_ms / max(walks, 1), "walks": walks, "best": best_global, "best_local": best,
        "alg_bytes": mean("algorithmic_bytes"), "mean_dirty": mean("dirty_nodes") / max(mean("candidates"), 1.0),
        "scored_per_step": mean("candidates"), "paired": bool(paired),
        "longest_step_ms": 1e3 * float(step_gaps.max()), "median_step_ms": 1e3 * float(np.median(step_gaps)),
        "long_steps": long_n, "long_seconds": long_rate,
    }


def kernel_only(ctx, tree, B: int, kind: int, nbatches: int, steps: int, warmup: int):
    """Round 1's headline: `nbatches` pre-built resident batches replayed, nothing but the scoring kernel."""
    from lvb_amd import api
    batches = []
    for _ in range(nbatches):
        offs, edits = tree.propose_batch(kind, B)
        bh = api.C.c_void_p()
        ctx._chk(ctx.lib.lvbgpu_batch_build(ctx.h, B, offs, edits.ctypes.data, None, api.C.byref(bh)))
        batches.append(api.Batch(ctx, bh, B))
    stats = [b.stats() for b in batches]
    for b in batches:
        b.launch()
    for i in range(warmup):
        batches[i % nbatches].launch()
    for b in batches:
        b.lengths()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.timer_start()
    for i in range(steps):
        batches[i % nbatches].launch()
    kernel_ms = ctx.timer_stop()
    wall = time.perf_counter() - t0
    launched = [stats[i % nbatches] for i in range(steps)]
    alg = float(np.mean([s["algorithmic_bytes"] for s in launched]))
    d = float(np.mean([s["dirty_nodes"] / s["candidates"] for s in launched]))
    for b in batches:
        b.lengths()
        b.free()
    launch_ms = kernel_ms / steps
    return {"value": B * steps / wall, "unit": "trees/s", "launch_ms": launch_ms, "mean_dirty_nodes": round(d, 2),
            "achieved_gbs": alg / (launch_ms * 1e-3) / 1e9, "algorithmic_bytes_per_launch": alg,
            "what": f"{nbatches} resident host-built batches (longest program first) replayed: kernel launches only, "
                    "lengths read once after the loop"}


def roofline_block(ctx, B: int, alg_bytes: float, launch_ms: float, mean_dirty: float, traffic, probe_reps: int = 20,
                   paired: bool = False):
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    probe = ctx.probe_l2(B, max(8, int(round(mean_dirty + 3))), probe_reps)
    # which variant the pipelined loop launches: small launches (B x tiles < 32768 waves) hand their lengths over through
    # watcher waves (<.., 2>, api_propose.cpp WATCH_PIPELINED_MAX_ITEMS), the others are copied back (<.., 0>)
    ntiles = (int(ctx.nwords) + 127) // 128
    kernel = KERNEL if ("pair" in KERNEL or B * ntiles >= 32768) else KERNEL.replace(", 0>", ", 2>")
    if paired and "pair" not in kernel:   # the library paired these batches by itself (long programs: DESIGN.md section 3)
        kernel = "lvbgpu::fitch_walk_pair<false, false>" if B * ntiles >= 32768 else "lvbgpu::fitch_walk_pair<false, true>"
    out = {
        "bound": "l2", "achieved": achieved, "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": achieved / L2_PEAK_GBS,
        "traffic": traffic, "kernel": kernel, "launch_ms": launch_ms, "algorithmic_bytes_per_launch": alg_bytes,
        "probe": {"value": probe, "unit": "GB/s", "frac": achieved / probe if probe > 0 else None,
                  "source": "lvbgpu_probe_l2 in this process: pure loads, the walk's geometry and access pattern"},
        "note": "algorithmic bytes = (D+3) clean rows x nwords x 8 per candidate (SURVEY.md 8d), D measured on the "
                "timed batches; the rows are served by the XCD L2s (97 % hit rate), so the bound is the L2 -> CU "
                "path, not HBM; `hbm` below is what actually crosses the HBM interface",
    }
    if traffic:
        gbs = traffic / (launch_ms * 1e-3) / 1e9
        out["traffic"] = float(traffic)
        out["hbm"] = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                      "reuse": alg_bytes / traffic,
                      "source": "rocprofv3 PMC FETCH_SIZE (x2, gfx950) + WRITE_SIZE per launch, profiles/traffic.json"
                                + (f" ({traffic.source})" if getattr(traffic, "source", "") else ""),
                      # the counters were collected in another run than this one: true when the walk's sources have
                      # changed since (the entry carries their hash)
                      "stale": bool(getattr(traffic, "stale", True))}
    return out


def cpu_reference_on_tree(rows, kind: int, budget_s: float, tree_arrays, expect_length, taxa_sites_seed, all_cores=True):
    """LVB's CPU path on the tree the GPU leg scored: the compiled reference's mutate_* (incl. its treecopy) +
    incremental getplen over random neighbours of that tree (nothing accepted, so the tree and the
    dirty-path lengths stay the GPU leg's)."""
    from oracle import binding as ob
    if ob.load_ref() is None:
        return None
    parent, left, right, root = tree_arrays
    rr = ob.RefRun(rows=rows, seed=12345, nproc=1)
    try:
        rr.set_topology(parent, left, right, root)
        full = rr.getplen(0)
        tg, tm, _, _ = rr.time_proposals(kind, 20, 0)
        per = max((tg + tm) / 20, 1e-6)
        reps = int(max(50, min(20000, budget_s / per)))
        tg, tm, _, ds = rr.time_proposals(kind, reps, 0)
        tfull, _ = rr.time_full(3)
        out = {
            "value": reps / (tg + tm), "unit": "trees/s", "cores": 1, "kind": "reference",
            "mean_dirty_nodes": round(ds / reps, 2), "getplen_only_trees_per_s": reps / tg,
            "tree_length_matches_gpu": bool(full == expect_length),
            "sample": f"{reps} {['NNI', 'SPR', 'TBR'][kind]} neighbours of the GPU leg's own tree (reference mutate incl. "
                      f"treecopy {1e3 * tm / reps:.3f} ms + incremental getplen {1e3 * tg / reps:.3f} ms, serial branch); "
                      f"full getplen {1e3 * tfull / 3:.2f} ms",
        }
    finally:
        rr.close()
    if all_cores:
        out["all_cores"] = cpu_all_cores(kind, min(budget_s, 8.0), tree_arrays, taxa_sites_seed)
    return out


def cpu_reference_anneal(rows, seconds: float, seed: int, gpu_log):
    """The second half of the metric for LVB's own CPU path: the reference PROGRAM (oracle/_ref/lvb_ref, the reference
    compiled here, its default schedule and all the threads it takes) annealing the same alignment for a bounded time.
    Its progress log has one line per 50 000 rearrangements (LVB.h:98), too coarse for a sample of seconds, so its
    per-rearrangement record (-v: changeAccepted.tsv, Solve.c:467) is polled instead: (wall clock, rearrangements,
    current length).  `gpu_seconds_to_same_length` = when the GPU chains' best length first got as low as the
    reference's after its whole sample."""
    import shutil
    import subprocess
    import tempfile
    exe = ROOT / "oracle" / "_ref" / "lvb_ref"
    if not exe.exists() or seconds <= 0:
        return None
    td = Path(tempfile.mkdtemp(prefix="lvbref_sa_"))
    try:
        with open(td / "infile", "w") as f:
            f.write(f"{len(rows)} {len(rows[0])}\n")
            for i, r in enumerate(rows):
                f.write(f"t{i:<9d}{r.decode()}\n")
        t0 = time.perf_counter()
        with open(td / "stdout.txt", "w") as so:
            # -p 4: left to itself the program starts one thread per hardware thread of the HOST (256 on the GPU box, of
            # which the job may use a few) and does 97 rearrangements/s; 1, 2, 4 threads give 290-330/s
            p = subprocess.Popen([str(exe), "-v", "-i", "infile", "-s", str(seed), "-p", "4"], cwd=td, stdout=so,
                                 stderr=subprocess.STDOUT)
        curve, rec = [], td / "changeAccepted.tsv"

        def last_record():
            try:
                with open(rec, "rb") as f:
                    f.seek(0, 2)
                    f.seek(max(0, f.tell() - 4096))
                    lines = [ln for ln in f.read().split(b"\n")[1:-1] if ln.count(b"\t") == 5]  # whole lines only
                it, _, _, length, _, _ = lines[-1].split(b"\t")
                return int(it), int(length)
            except (OSError, IndexError, ValueError):
                return None
        try:
            while p.poll() is None and time.perf_counter() - t0 < seconds:
                time.sleep(0.5)
                got = last_record()
                if got and (not curve or got[0] != curve[-1][1]):
                    curve.append([round(time.perf_counter() - t0, 2), got[0], got[1]])
        finally:
            if p.poll() is None:
                p.terminate()
                try:
                    p.wait(5)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        took = time.perf_counter() - t0
        threads = None
        for ln in open(td / "stdout.txt", errors="replace"):
            if "PThreads:" in ln:
                threads = int(ln.split(":")[1])
        if not curve:
            return {"kind": "reference", "seconds": round(took, 2), "threads": threads, "note": "no rearrangement recorded in the sample"}
        out = {
            "kind": "reference", "program": "oracle/_ref/lvb_ref -v -i infile -s %d -p 4 (defaults otherwise: SEQ-TNS, geometric cooling)" % seed,
            "seconds": round(took, 2), "threads": threads, "rearrangements": curve[-1][1], "length": curve[-1][2],
            "rearrangements_per_s": round(curve[-1][1] / max(curve[-1][0], 1e-9), 1),
            "length_vs_wallclock": curve[:: max(1, len(curve) // 8)] + curve[-1:],
        }
        hit = next((t for t, b in gpu_log if b <= curve[-1][2]), None)
        out["gpu_seconds_to_same_length"] = None if hit is None else round(hit, 4)
        return out
    finally:
        shutil.rmtree(td, ignore_errors=True)


def cpu_all_cores(kind: int, seconds: float, tree_arrays, taxa_sites_seed):
    """One reference process per host core at the same time (the reference is non-reentrant), each
    scoring random neighbours of the same tree."""
    taxa, sites, seed, dist = taxa_sites_seed
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 32))
    with tempfile.TemporaryDirectory() as td:
        tf = Path(td) / "tree.npz"
        parent, left, right, root = tree_arrays
        np.savez(tf, parent=parent, left=left, right=right, root=root)
        cmd = [sys.executable, "-m", "oracle.cpu_bench", "--taxa", str(taxa), "--sites", str(sites), "--seed", str(seed),
               "--dist", dist, "--kind", str(kind), "--seconds", str(seconds), "--tree", str(tf)]
        procs = [subprocess.Popen(cmd + ["--chain", str(c)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                  text=True) for c in range(cores)]
        rate, done, dirty = 0.0, 0, 0.0
        for p in procs:
            try:
                out, _ = p.communicate(timeout=seconds * 6 + 120)
                d = json.loads(out.strip().splitlines()[-1])
                rate += d["reps"] / (d["t_getplen"] + d["t_mutate"])
                dirty += d["dirty"]
                done += 1
            except Exception:
                p.kill()
    return {"value": rate, "unit": "trees/s", "cores": done, "mean_dirty_nodes": round(dirty / max(done, 1), 2),
            "sample": f"{done} concurrent single-threaded reference processes, {seconds:.0f} s each, same tree"}


def cpu_port_baseline(rows, kind: int, budget_s: float, tree):
    """Fallback when the compiled reference did not travel: our C restatement (oracle/fitch_oracle.c)."""
    from oracle import binding as ob
    from tests.helpers import apply_edits, parents_of
    enc = ob.encode_rows(rows)
    n, nwords = enc.shape
    _, left, right = tree.arrays()
    cur = ob.OracleTree(n, nwords, enc)
    l64, r64 = left.astype(np.int64), right.astype(np.int64)
    cur.set_topology(parents_of(l64, r64), l64, r64, tree.root)
    cur.getplen()
    prop = ob.OracleTree(n, nwords)
    t_total, reps, dirty = 0.0, 0, 0
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
        dirty += prog["dirty"]
        reps += 1
    return {"value": reps / t_total, "unit": "trees/s", "cores": 1, "kind": "port",
            "mean_dirty_nodes": round(dirty / max(reps, 1), 2),
            "sample": f"{reps} neighbours of the GPU leg's tree through oracle/fitch_oracle.c (treecopy + incremental getplen)"}


def tree_arrays_of(tree):
    p, l, r = tree.arrays()
    return p.astype(np.int64), l.astype(np.int64), r.astype(np.int64), int(tree.root)


def load_traffic(args, mixed: bool):
    return traffic_for(args.taxa, args.sites, args.batch, args.move, mixed)


KERNEL_SOURCES = ("lvb_amd/csrc/fitch_kernels.hip", "lvb_amd/csrc/walk_body.hpp", "lvb_amd/csrc/kernels.hpp",
                  "lvb_amd/csrc/api_propose.cpp")


def kernel_source_sha() -> str:
    """What an HBM-traffic entry of profiles/traffic.json was measured on: the scoring walk, its launcher and the batch
    builder.  The PMC passes run on another box at another time than a bench run; the entry carries this hash and the
    line says `stale` when the sources have moved since."""
    import hashlib
    hsh = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        hsh.update((ROOT / rel).read_bytes())
    return hsh.hexdigest()[:16]


class Traffic(float):
    """HBM bytes per launch from profiles/traffic.json, with where it came from"""
    stale = True
    source = ""


def traffic_for(taxa: int, sites: int, batch: int, move: str, mixed: bool = False):
    tfile = ROOT / "profiles" / "traffic.json"
    if not tfile.exists():
        return None
    t = json.loads(tfile.read_text())
    for e in t if isinstance(t, list) else [t]:
        if (e.get("taxa"), e.get("sites"), e.get("batch"), e.get("move"), bool(e.get("mixed_walk", False))) == \
                (taxa, sites, batch, move, mixed):
            v = Traffic(e["hbm_bytes_per_launch"])   # rocprofv3 PMC passes (profiles/collect.sh), gfx950-corrected
            v.stale = e.get("kernel_sha") != kernel_source_sha()
            v.source = e.get("source", "")
            return v
    return None


def config_leg(ranks, device: int, cfg: str, taxa: int, sites: int, move: str, batches, seed: int, steps: int, walk: int = 75,
               multi_chain=()):
    """One of BASELINE.json's other single-GPU configurations through the headline's own measurement: submit -> lengths
    with two steps in flight, the walk's duration from HIP events, D measured on the timed batches, the roofline block
    (L2 -> CU path; `hbm` from the rocprofv3 PMC entry of that shape in profiles/traffic.json where there is one)."""
    from lvb_amd import api, host
    t0 = time.perf_counter()
    rows, min_len = host.prepare_alignment(synth_rows(taxa, sites, seed, "tree"))
    ctx = api.FitchContext(text_rows=rows, device=device)
    tree = host.HostTree(taxa, seed=ranks.restart_seed(seed))
    tree.upload(ctx)
    length = random_walk(ctx, tree, MOVES[move], walk)
    out = {"config": cfg, "taxa": taxa, "sites": sites, "sites_after_constant_cut": len(rows[0]), "nwords": ctx.nwords,
           "move": move, "start_tree": f"random + {walk} accepted moves", "tree_length": length, "min_len_tree": min_len,
           "resident_tree_block_mb": round((2 * taxa - 3) * ((ctx.nwords + 127) // 128 * 128) * 8 / 1e6, 1),
           "setup_seconds": round(time.perf_counter() - t0, 1)}
    for B in batches:
        r = submit_to_lengths(ctx, ranks, B, MOVES[move], steps, 5, 1000, settle=40)
        out[f"B{B}"] = {
            "value": r["scored_per_step"] * steps / r["elapsed_s"], "unit": "trees/s", "ms_per_step": 1e3 * r["elapsed_s"] / steps,
            "steps": steps, "walk_us": 1e3 * r["launch_ms"], "mean_dirty_nodes": round(r["mean_dirty"], 2),
            "roofline": roofline_block(ctx, B, r["alg_bytes"], r["launch_ms"], r["mean_dirty"],
                                       traffic_for(taxa, sites, B, move), 6 if taxa >= 1000 else 20, paired=r["paired"]),
        }
    if multi_chain:
        # small shapes do not fill the chip with one chain's batch (cfg2: 1024 candidates x 5 tiles = 5120 waves against
        # 8192 wave slots, a 13 us walk inside a launch chain): the path's own remedy is the one annealing uses - R
        # independent chains (restarts) in one context, ONE generator launch and ONE walk for all of them per step
        # (lvbgpu_chains_submit with R draws); every chain's lengths are what its own single-chain step gives
        # (tests/test_gpu_chains.py::test_cfg2_chains_step_equals_single_chain_steps)
        B = batches[0]
        for R in multi_chain:
            mctx = api.FitchContext(text_rows=rows, device=device)
            mctx.set_chains(R)
            trees = []
            for c in range(R):
                t = host.HostTree(taxa, seed=ranks.restart_seed(seed) * 100 + c)
                mctx.select_chain(c)
                t.upload(mctx)
                random_walk(mctx, t, MOVES[move], walk)
                trees.append(t)
            draws = np.zeros(R, dtype=api.DRAW_DTYPE)
            outs = [np.zeros(R * B, dtype=np.int64), np.zeros(R * B, dtype=np.int64)]
            for c in range(R):
                draws[c]["chain"], draws[c]["count"], draws[c]["kind"] = c, B, MOVES[move]

            def submit(slot, seed):
                for c in range(R):
                    draws[c]["seed"] = (seed * 64 + c) & 0xFFFFFFFFFFFFFFFF
                mctx._chk(mctx.lib.lvbgpu_chains_submit(mctx.h, slot, R, draws.ctypes.data))

            def run(n, first_seed):
                for j in range(min(2, n)):
                    submit(j % 2, first_seed + j)
                for i in range(n):
                    mctx._chk(mctx.lib.lvbgpu_chains_collect(mctx.h, i % 2, outs[i % 2]))
                    if i + 2 < n:
                        submit(i % 2, first_seed + i + 2)
            run(60, 1)
            mctx.synchronize()
            t1 = time.perf_counter()
            run(steps, 1000)
            mctx.synchronize()
            dt = time.perf_counter() - t1
            out[f"chains{R}_B{B}"] = {"value": R * B * steps / dt, "unit": "trees/s", "ms_per_step": 1e3 * dt / steps, "chains": R,
                                      "candidates_per_step": R * B,
                                      "what": f"{R} chains x {B} candidates per step: one generator launch and one walk for all of them, "
                                              "two steps in flight"}
            for t in trees:
                t.close()
            mctx.close()
    tree.close()
    ctx.close()
    return out


# ----------------------------------------------------------------------------------------- one rank

def rank_main(args) -> None:
    from lvb_amd.launch import Ranks, call_with_deadline, pin_to_share_of_cores
    # this rank's share of the host's cores, before the scoring library (its thread pool) is loaded
    host_share = pin_to_share_of_cores()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("NCCL_DEBUG", "WARN")   # read when RCCL is first initialised: a failing CommInitRank says why
    ranks = Ranks()                       # torch.distributed (nccl = RCCL) only when WORLD_SIZE > 1
    rank, world = ranks.rank, ranks.world
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from lvb_amd import api, host

    kind = MOVES[args.move]
    B = args.batch
    t_setup = time.perf_counter()
    raw_rows = synth_rows(args.taxa, args.sites, args.seed, args.dist)
    rows, min_len = host.prepare_alignment(raw_rows)
    ctx = api.FitchContext(text_rows=rows, device=ranks.device)    # encode on the device
    tree = host.HostTree(args.taxa, seed=ranks.restart_seed(args.seed))  # each rank: its own restart
    tree.upload(ctx)
    length = random_walk(ctx, tree, kind, args.walk)                # short random walk, as BASELINE.md
    fresh_arrays = tree_arrays_of(tree)
    setup_s = time.perf_counter() - t_setup

    own_comm = False
    comm_init_stuck = False
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
                # ncclCommInitRank has no deadline of its own: a rank that never arrives would hang all the others.  Waited
                # for at most --comm-init-seconds; a rank that gives up says so, every rank then takes the torch fallback
                # (the agreement below), and the process ends through os._exit (the stuck call keeps its thread)
                done, res = call_with_deadline(lambda: ctx.comm_init(world, rank, uid), args.comm_init_seconds)
                if not done:
                    comm_init_stuck = True
                    print(f"[rank {rank}] lvbgpu_comm_init did not return within {args.comm_init_seconds:.0f} s; "
                          "min-reduce falls back to torch.distributed", file=sys.stderr)
                elif isinstance(res, BaseException):   # keep the scaling run alive: same reduction through torch's RCCL
                    print(f"[rank {rank}] lvbgpu_comm_init failed ({res}); min-reduce falls back to torch.distributed",
                          file=sys.stderr)
                else:
                    own_comm = True
        own_comm = bool(ranks.sum_over_ranks(int(own_comm)) == world)

    def reduce_best(best_local: int) -> int:
        if world == 1:
            return best_local
        if own_comm:
            return ctx.allreduce_min(best_local)[0]
        return int(-ranks.max_over_ranks(-float(best_local)))

    # ---- the timed region: K steps of submit -> lengths on the host (+ the min-reduce over ranks)
    head = submit_to_lengths(ctx, ranks, B, kind, args.steps, args.warmup, 1000, reduce_best if world > 1 else None,
                             settle=args.settle, long_steps=args.long_steps)
    per_step = head["scored_per_step"] if head["scored_per_step"] else B
    total_trees = per_step * args.steps * world
    per_rank = ranks.all_values(per_step * args.steps / head["elapsed_local"])   # each rank's own clock around ITS region
    print(f"[rank {rank}] timed region {1e3 * head['elapsed_s']:.2f} ms for {args.steps} steps "
          f"(local loop {1e3 * head['steps_s_local']:.2f} ms), walk {head['launch_ms'] * 1e3:.1f} us x {head['walks']}",
          file=sys.stderr)

    out = {
        "metric": "candidate trees scored/sec (Fitch getplen), 500 taxa x 50k sites",
        "value": total_trees / head["elapsed_s"],
        "unit": "trees/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * head["elapsed_s"] / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.taxa} taxa x {args.sites} sites synthetic DNA "
                        f"({'tree-like, 10% substitutions' if args.dist == 'tree' else 'i.i.d. uniform'}), "
                        f"{args.move.upper()} neighbourhood, incremental getplen semantics, B={B} candidates per step; "
                        f"a step = submit -> lengths on the host (neighbours drawn, programmed and scored on the GPU, "
                        f"B lengths read back; two steps in flight: lvbgpu_chains_submit / _collect); "
                        f"start tree + {args.walk} accepted moves",
            "taxa": args.taxa, "sites_after_constant_cut": len(rows[0]), "nwords": ctx.nwords,
            "batch": B, "move": args.move, "mean_dirty_nodes": round(head["mean_dirty"], 2),
            "parallelism": f"{world} independent restart(s), one per GPU; best length min-reduced over RCCL"
                           + ("" if world == 1 else (" by lvbgpu_allreduce_min" if own_comm else " (torch fallback)")),
            "best_length": head["best"], "min_len_tree": min_len, "setup_seconds": round(setup_s, 2),
            "reducer": "none (one rank)" if world == 1 else ("lvbgpu_allreduce_min (the library's own RCCL communicator)"
                                                             if own_comm else "torch.distributed all_reduce (fallback)"),
            "comm_size": ctx.comm_size() if own_comm else world, "reduce_ms": round(head["reduce_ms"], 4),
            "per_rank_trees_per_s": [round(v) for v in per_rank],
            "cores_per_rank": host_share["cores_per_rank"], "host_threads_per_rank": host_share["threads"],
            "pinned_to_share_of_cores": host_share["pinned"],
            "timed_region": "one untimed min-reduce + barrier, t0, K steps, min-reduce of the best length (reduce_ms), "
                            "synchronize, barrier, t1; value = all ranks' candidates / max over ranks of (t1 - t0)",
        },
        "roofline": roofline_block(ctx, B, head["alg_bytes"], head["launch_ms"], head["mean_dirty"],
                                   load_traffic(args, False), paired=head["paired"]),
        # did a step of the timed region stall?  Its longest and median step (lengths on the host to lengths on the host,
        # this rank), and the same loop for `long_steps` more steps BEHIND the timed region (this rank's own clock)
        "timed_region_steps": {"longest_ms": round(head["longest_step_ms"], 4), "median_ms": round(head["median_step_ms"], 4)},
    }
    if head["long_steps"]:
        out["value_long"] = {"value": per_step * head["long_steps"] * world / head["long_seconds"], "unit": "trees/s",
                             "steps": head["long_steps"], "ms_per_step": 1e3 * head["long_seconds"] / head["long_steps"],
                             "what": "the same pipelined loop for more steps, right behind the timed region (rank 0's clock, "
                                     "times the number of ranks)"}
    steps_side = max(20, min(args.steps, 100))
    if not args.headline_only:
        one = submit_to_lengths(ctx, ranks, B, kind, steps_side, 5, 3000, depth=1)
        out["one_at_a_time"] = {"value": one["scored_per_step"] * steps_side / one["elapsed_s"], "unit": "trees/s",
                                "ms_per_step": 1e3 * one["elapsed_s"] / steps_side, "walk_ms": one["launch_ms"],
                                "what": "the same step with ONE batch in flight (lvbgpu_propose_score): a step's latency"}
        out["kernel_only"] = kernel_only(ctx, tree, B, kind, args.nbatches, steps_side, min(args.warmup, 10))

    extras = rank == 0 and world == 1 and not args.headline_only
    if extras and not args.no_shapes:
        shapes = {}
        for b in (256, 1024, 16384):
            r = submit_to_lengths(ctx, ranks, b, kind, steps_side if b <= 4096 else max(20, steps_side // 2), 5, 5000)
            nst = steps_side if b <= 4096 else max(20, steps_side // 2)
            shapes[f"B{b}"] = {"value": r["scored_per_step"] * nst / r["elapsed_s"], "unit": "trees/s",
                               "ms_per_step": 1e3 * r["elapsed_s"] / nst, "walk_ms": r["launch_ms"],
                               "mean_dirty_nodes": round(r["mean_dirty"], 2)}
        if args.dist == "tree":
            # SURVEY.md 8(d) "U": i.i.d. uniform cells (almost every combine is a union), same start tree
            urows, _ = host.prepare_alignment(synth_rows(args.taxa, args.sites, args.seed, "uniform"))
            uctx = api.FitchContext(text_rows=urows, device=ranks.device)
            _, ul, ur = (a.copy() for a in tree.arrays())
            uctx.set_tree(ul, ur, tree.root)
            r = submit_to_lengths(uctx, ranks, B, kind, steps_side, 5, 1000)
            shapes["uniform"] = {"value": r["scored_per_step"] * steps_side / r["elapsed_s"], "unit": "trees/s",
                                 "ms_per_step": 1e3 * r["elapsed_s"] / steps_side, "walk_ms": r["launch_ms"],
                                 "mean_dirty_nodes": round(r["mean_dirty"], 2), "batch": B,
                                 "sites_after_constant_cut": len(urows[0])}
            uctx.close()
        out["shapes"] = shapes

    if extras and not args.no_configs:
        # BASELINE.json configs[1] and configs[4] at one GPU, measured like the headline (same code path, same events)
        out["configs"] = {
            "cfg2": config_leg(ranks, ranks.device, "BASELINE configs[1]: 64 taxa x 10 000 sites, NNI neighbourhood, batch = 1024",
                               64, 10000, "nni", (1024,), args.seed, steps_side, multi_chain=(4, 8)),
            "cfg5": config_leg(ranks, ranks.device, "BASELINE configs[4] at one GPU: 2000 taxa x 200 000 sites, TBR neighbourhood, "
                               "matrix and tree HBM-resident", 2000, 200000, "tbr", (1024, 4096), args.seed, max(20, steps_side // 2)),
        }

    if extras and args.mixed_walk > 0:
        # the shape an annealing run spends its time on: the walk drifts toward uniform-random trees whose
        # root-ward paths are 2-3x longer.  Same legs, same tree for GPU and CPU.
        mtree = host.HostTree(left=tree.arrays()[1], right=tree.arrays()[2], root=tree.root, seed=args.seed * 31 + 7)
        mlen = random_walk(ctx, mtree, kind, args.mixed_walk)
        m = submit_to_lengths(ctx, ranks, B, kind, steps_side, 5, 9000)
        mw = {
            "accepted_moves": args.walk + args.mixed_walk, "mean_dirty_nodes": round(m["mean_dirty"], 2),
            "value": m["scored_per_step"] * steps_side / m["elapsed_s"], "unit": "trees/s",
            "ms_per_step": 1e3 * m["elapsed_s"] / steps_side, "steps": steps_side,
            "roofline": roofline_block(ctx, B, m["alg_bytes"], m["launch_ms"], m["mean_dirty"], load_traffic(args, True), 10,
                                       paired=m["paired"]),
            "two_candidates_per_wave": m["paired"],
            "kernel_only": kernel_only(ctx, mtree, B, kind, args.nbatches, steps_side, 5),
        }
        mixed_arrays, mixed_len = tree_arrays_of(mtree), mlen
        out["mixed_walk"] = mw
        # back to the headline tree for the annealing leg
        tree.upload(ctx)

    if args.anneal_seconds > 0 and not args.headline_only and (extras or world > 1):
        # second half of the metric: best length vs wall clock, whole host loop included (starting temperature,
        # neighbours drawn + scored + committed on the GPU, accept / cool on the host) - not part of `value`.
        # R independent chains are stepped together on this GPU: a device step serves all of them.
        def params_for(c, seed_base=None):
            p = host.anneal_defaults()
            p.seed = (args.anneal_seed if seed_base is None else seed_base) * 7919 + 1000 * rank + c + 1
            p.algorithm = {"nni": 10, "spr": 11, "tbr": 12}[args.move]
            p.batch = args.anneal_batch
            p.t0 = 0.0   # estimated as StartingTemperature() does (65 % of uphill moves accepted)
            p.min_len_tree = min_len
            p.max_seconds = args.anneal_seconds
            p.log_cap = 4096
            return p
        R = max(1, args.anneal_chains)
        actx = api.FitchContext(text_rows=rows, device=ranks.device)
        atrees = [host.HostTree(args.taxa, seed=ranks.restart_seed(args.anneal_seed) * 100 + c) for c in range(R)]
        res, log = host.anneal_chains(actx, atrees, [params_for(c) for c in range(R)])
        anneal_log = list(log)
        single_log = []
        keep = log[:: max(1, len(log) // 12)] + log[-1:]
        secs = max(r["seconds"] for r in res)
        tot = lambda k: sum(r[k] for r in res)
        if rank == 0:
            out["anneal"] = {
                "chains": R, "seconds": round(secs, 3),
                "best_length": min(r["best_length"] for r in res), "best_lengths": [r["best_length"] for r in res],
                "start_lengths": [r["start_length"] for r in res],
                "scored": tot("scored"), "consumed": tot("consumed"), "accepted": tot("accepted"),
                "device_steps": max(r["device_steps"] for r in res), "chain_steps": tot("device_steps"),
                "scored_per_s": round(tot("scored") / secs), "consumed_per_s": round(tot("consumed") / secs),
                # the same while at least half of the chains were still annealing (robust against a straggler chain)
                "seconds_busy": round(res[0]["seconds_busy"], 3),
                "scored_per_s_busy": round(res[0]["scored_busy"] / max(res[0]["seconds_busy"], 1e-9)),
                "seconds_done": [round(r["seconds_done"], 3) for r in res], "seed": args.anneal_seed,
                "device_fraction": round(res[0]["seconds_device"] / secs, 3), "batch": args.anneal_batch,
                "temperatures": [r["temperatures"] for r in res], "frozen": sum(r["frozen"] for r in res),
                "best_length_vs_wallclock": [[round(t, 3), b] for t, b in keep],
                # THE annealing rate: candidates scored per second while at least half the chains were at it (a rate taken
                # until the last chain has frozen says how long that chain took)
                "rate": round(res[0]["scored_busy"] / max(res[0]["seconds_busy"], 1e-9)), "rate_unit": "trees/s",
                "what": f"{R} independent chains (own seeds and start trees) dealt to lanes (contexts of their own, two from 16 "
                        "chains on) that one host thread serves in turn; a lane's chains are stepped together: one post launch "
                        "(commit walk, table rebuilds, the next generator) and one scoring walk per step for all of them; "
                        "starting temperatures included",
            }
            # ... and on another seed base (the default one was chosen so that no chain is a straggler: bench.py --anneal-seed)
            if args.anneal_second_seed >= 0:
                strees = [host.HostTree(args.taxa, seed=ranks.restart_seed(args.anneal_second_seed) * 100 + c) for c in range(R)]
                res2, log2 = host.anneal_chains(actx, strees, [params_for(c, args.anneal_second_seed) for c in range(R)])
                secs2 = max(r["seconds"] for r in res2)
                out["anneal"]["second_seed"] = {
                    "seed": args.anneal_second_seed, "seconds": round(secs2, 3), "best_length": min(r["best_length"] for r in res2),
                    "scored_per_s": round(sum(r["scored"] for r in res2) / secs2), "seconds_busy": round(res2[0]["seconds_busy"], 3),
                    "rate": round(res2[0]["scored_busy"] / max(res2[0]["seconds_busy"], 1e-9)),
                    "frozen": sum(r["frozen"] for r in res2), "seconds_done_max": round(max(r["seconds_done"] for r in res2), 3)}
                for t in strees:
                    t.close()
            # one chain alone, for comparison: the same loop with R = 1
            # (seeded as in rounds 1 and 2 - from --seed, not --anneal-seed; with --single-chain-levels 0 it is the same chain
            # as in their lines, with runs of accepted moves - the default - the hot phase draws by the host's law)
            fresh = host.HostTree(args.taxa, seed=ranks.restart_seed(args.seed) * 100)
            p1 = params_for(0)
            p1.seed = args.seed * 7919 + 1000 * rank + 1
            p1.run_levels = args.single_chain_levels   # a lone chain: runs of acceptances in one step while it is hot
            one, single_log = host.anneal_chains(actx, [fresh], [p1])
            fresh.close()
            out["anneal"]["single_chain"] = {
                "seconds": round(one[0]["seconds"], 3), "best_length": one[0]["best_length"], "scored": one[0]["scored"],
                "consumed": one[0]["consumed"], "device_steps": one[0]["device_steps"],
                "scored_per_s": round(one[0]["scored"] / one[0]["seconds"]), "frozen": one[0]["frozen"],
                "run_levels": args.single_chain_levels,
                "best_length_vs_wallclock": [[round(t, 3), b] for t, b in (single_log[:: max(1, len(single_log) // 8)] + single_log[-1:])],
                "what": "one chain alone in its context; run_levels > 0: while it accepts most of what it sees its candidates are "
                        "cumulative (host-drawn, up to that many accepted moves per scoring walk: lvbhost_anneal_params::run_levels)"}
        for t in atrees:
            t.close()
        actx.close()
    # LVB's own CPU path last: 32 reference processes at once leave the host's CPU quota throttled for a while, which
    # the legs above (host threads beside the GPU) would feel
    if extras and not args.no_cpu_baseline and args.dist == "tree":
        cb = cpu_reference_on_tree(rows, kind, args.cpu_seconds, fresh_arrays, length,
                                   (args.taxa, args.sites, args.seed, args.dist))
        out["cpu_baseline"] = cb if cb is not None else cpu_port_baseline(rows, kind, args.cpu_seconds, tree)

    if extras and not args.no_cpu_baseline and args.dist == "tree" and "anneal" in out:
        out["anneal"]["reference_cpu"] = cpu_reference_anneal(rows, min(2.5 * args.cpu_seconds, 30.0), args.seed, anneal_log)
        ref = out["anneal"]["reference_cpu"]
        if ref and "length" in ref and single_log:   # ... and when the chain that ran alone passed that length
            hit = next((t for t, b in single_log if b <= ref["length"]), None)
            ref["gpu_single_chain_seconds_to_same_length"] = None if hit is None else round(hit, 4)

    if extras and not args.no_cpu_baseline and args.dist == "tree" and "mixed_walk" in out:
        out["mixed_walk"]["cpu_baseline"] = cpu_reference_on_tree(rows, kind, args.cpu_seconds, mixed_arrays, mixed_len,
                                                                  (args.taxa, args.sites, args.seed, args.dist), all_cores=False)
    tree.close()
    if not comm_init_stuck:   # (destroying the context would reach into a communicator that is still being set up)
        ctx.close()
    ranks.barrier()   # rank 0 has more legs than the others: nobody tears the process group down under it
    ranks.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if comm_init_stuck:       # the call that never came back keeps its thread: end the process under it
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


def main() -> int:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)          # before anything here can touch a GPU
    if args.dry_ranks:
        dry_rank_main(args)
    else:
        rank_main(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
