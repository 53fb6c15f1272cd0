"""A whole search on the MI355X path: read a PHYLIP alignment, cut constant columns, anneal, write
the best trees.  Mirrors the reference program's flow (Main.c:60-155) and result lines, with our own
host (batched SA over the device scorer); it is a convenience around the library, not a port of the CLI.

  python -m lvb_amd.search -i alignment.phy [-s seed] [-a 0|1] [-o outtree] [--batch B] [--device D]
  python -m lvb_amd.search -i alignment.phy --chains 32         # 32 restarts stepped together on one GPU
  python -m lvb_amd.search -i alignment.phy -s seed --exact     # the reference's own trajectory (-a 0|1|2)
  python -m torch.distributed.run --nproc-per-node 8 -m lvb_amd.search -i alignment.phy   # one restart per GPU

--exact reproduces the reference program decision for decision (same random stream, start trees,
proposals, cooling, treestack: lvb_amd/csrc/refsearch.cpp), so "Rearrangements evaluated", "Tree score",
"Topologies recovered" and the output trees are the reference's for that seed.  Without it the search
is the batched one (lvb_amd/csrc/anneal.cpp): same algorithm and statistics, its own random stream,
far more candidates per second.
"""
from __future__ import annotations

import argparse
import sys
import time

from . import api, host


def run_chains(path: str, seed: int = 1, chains: int = 16, algorithm: int = 1, batch: int = 4096, device: int = 0,
               out: str | None = "outtree", max_seconds: float = 0.0, cooling: int = 0, verbose: bool = True,
               fmt: str = "phylip", run_levels: int | None = None) -> dict:
    """`chains` independent restarts stepped together on ONE GPU (lvbhost_anneal_chains): every device step draws,
    scores and commits for all of them.  The trees written are the distinct topologies of the best length over all
    chains.  run_levels: runs of accepted moves per scoring walk while a chain is hot (lvbhost_anneal_params::run_levels);
    None: 3 for one or two chains - where it pays - else off."""
    if run_levels is None:
        run_levels = 3 if chains <= 2 else 0
    t0 = time.perf_counter()
    names, rows = host.read_alignment(path, fmt)
    n, m_read = len(rows), len(rows[0])
    if n < 5:
        raise ValueError("The data matrix must have at least 5 sequences.")  # Wrapper.c:54 (MIN_N)
    rows, min_len = host.prepare_alignment(rows)
    ctx = api.FitchContext(text_rows=rows, device=device)
    trees = [host.HostTree(n, seed=seed * 1000 + c) for c in range(chains)]
    params = []
    for c in range(chains):
        p = host.anneal_defaults()
        p.seed = seed * 1000 + c + 1
        p.algorithm, p.cooling_schedule, p.batch = algorithm, cooling, batch
        p.min_len_tree, p.max_seconds, p.t0, p.log_cap = min_len, max_seconds, 0.0, 4096
        p.run_levels = run_levels
        params.append(p)
    per_chain, log = host.anneal_chains(ctx, trees, params)
    best = min(r["best_length"] for r in per_chain)
    seen, newick = set(), []
    for r, t in zip(per_chain, trees):
        if r["best_length"] != best:
            continue
        for bt in t.best_trees():
            key = bt.canonical()               # exact, rooting-independent identity: one line per topology
            if key not in seen:
                seen.add(key)
                newick.append(host.newick(bt, names))
    if out:
        with open(out, "w") as f:
            f.writelines(newick)
    res = dict(chains=chains, best_length=best, best_lengths=[r["best_length"] for r in per_chain],
               consumed=sum(r["consumed"] for r in per_chain), scored=sum(r["scored"] for r in per_chain),
               topologies=len(newick), frozen=sum(r["frozen"] for r in per_chain), taxa=n, sites_read=m_read,
               sites_used=len(rows[0]), min_len_tree=min_len, wall_seconds=time.perf_counter() - t0, log=log, outtree=out,
               newick=newick)
    if verbose:
        ci = min_len / best
        print("\nSearch Results:")
        print(f"  Chains on this GPU:       {chains}  ({res['frozen']} frozen)")
        print(f"  Rearrangements evaluated: {res['consumed']}")
        print(f"  Candidates scored (GPU):  {res['scored']}")
        print(f"  Topologies recovered:     {res['topologies']}")
        print(f"  Tree score:               {best}")
        print(f"  Consistency index:        {ci:.2f}")
        print(f"  Homoplasy index:          {1 - ci:.2f}")
        print(f"  Total runtime (seconds):  {res['wall_seconds']:.2f}")
        if out:
            print(f"\nAll topologies written to '{out}'")
    for t in trees:
        t.close()
    ctx.close()
    return res


def run(path: str, seed: int = 1, algorithm: int = 1, batch: int = 4096, device: int = 0, out: str | None = "outtree",
        max_seconds: float = 0.0, cooling: int = 0, verbose: bool = True, device_proposals: int = 2,
        fmt: str = "phylip") -> dict:
    t0 = time.perf_counter()
    names, rows = host.read_alignment(path, fmt)
    n, m_read = len(rows), len(rows[0])
    if n < 5:
        raise ValueError("The data matrix must have at least 5 sequences.")  # Wrapper.c:54 (MIN_N)
    rows, min_len = host.prepare_alignment(rows)
    ctx = api.FitchContext(text_rows=rows, device=device)
    tree = host.HostTree(n, seed=seed)
    start = tree.upload(ctx)
    p = host.anneal_defaults()
    p.seed = seed
    p.algorithm = algorithm
    p.cooling_schedule = cooling
    p.batch = batch
    p.min_len_tree = min_len
    p.max_seconds = max_seconds
    p.t0 = 0.0  # StartingTemperature()
    p.device_proposals = int(device_proposals)
    p.log_cap = 4096
    res, log = host.anneal(ctx, tree, p)
    best = tree.best_trees()
    newick = [host.newick(t, names) for t in best]
    if out:
        with open(out, "w") as f:
            f.writelines(newick)
    res.update(taxa=n, sites_read=m_read, sites_used=len(rows[0]), min_len_tree=min_len, start_length=start,
               wall_seconds=time.perf_counter() - t0, log=log, outtree=out, newick=newick)
    if verbose:
        ci = min_len / res["best_length"]
        print("\nSearch Results:")
        print(f"  Rearrangements evaluated: {res['consumed']}")
        print(f"  Candidates scored (GPU):  {res['scored']}")
        print(f"  Topologies recovered:     {res['topologies']}")
        print(f"  Tree score:               {res['best_length']}")
        print(f"  Consistency index:        {ci:.2f}")
        print(f"  Homoplasy index:          {1 - ci:.2f}")
        print(f"  Total runtime (seconds):  {res['wall_seconds']:.2f}")
        if out:
            print(f"\nAll topologies written to '{out}'")
    tree.close()
    ctx.close()
    return res


def run_exact(path: str, seed: int, algorithm: int = 1, cooling: int = 0, device: int = 0, out: str | None = "outtree",
              verbose: bool = True, max_batch: int = 0, max_trees: int = 0, fmt: str = "phylip") -> dict:
    """The reference's trajectory for this seed on the device scorer (Main.c:60-155 flow)."""
    t0 = time.perf_counter()
    names, rows = host.read_alignment(path, fmt)
    n, m_read = len(rows), len(rows[0])
    if n < 5:
        raise ValueError("The data matrix must have at least 5 sequences.")
    rows, min_len = host.prepare_alignment(rows)
    ctx = api.FitchContext(text_rows=rows, device=device)
    p = host.refsearch_defaults()
    p.seed, p.algorithm, p.cooling_schedule, p.min_len_tree, p.max_trees = seed, algorithm, cooling, min_len, max_trees
    if max_batch:
        p.max_batch = max_batch
    res, tree = host.reference_search(ctx.h, p)
    best = tree.best_trees()
    if out:
        with open(out, "w") as f:
            for t in best:
                if t.root != 0:  # PrintTreestack re-roots at the first taxon (Treestack.c:402-403)
                    t.apply(t.reroot_edits(0), 0)
                f.write(host.newick(t, names))
    res.update(taxa=n, sites_read=m_read, sites_used=len(rows[0]), min_len_tree=min_len,
               wall_seconds=time.perf_counter() - t0, outtree=out)
    if verbose:
        ci = min_len / res["best_length"]
        print(f"  SA Starting Temperature: {res['t0']:.8f}")
        print("\nSearch Results:")
        print(f"  Rearrangements evaluated: {res['rearrangements']}")
        print(f"  Topologies recovered:     {res['trees']}")
        print(f"  Tree score:               {res['best_length']}")
        print(f"  Consistency index:        {ci:.2f}")
        print(f"  Homoplasy index:          {1 - ci:.2f}")
        print(f"  Total runtime (seconds):  {res['wall_seconds']:.2f}")
        if out:
            print(f"\nAll topologies written to '{out}'")
    tree.close()
    ctx.close()
    return res


def run_restarts(ranks, path: str, seed: int, out: str | None = "outtree", verbose: bool = True, **kw) -> dict:
    """One independent restart per rank (= per GPU), as SURVEY.md 8e shards the path: every rank anneals from
    its own start tree and seed on its own device; the only exchange is the best length (a min-reduce) and
    the rank that holds it writes the trees.  `ranks` is lvb_amd.launch.Ranks()."""
    res = run(path, seed=ranks.restart_seed(seed), device=ranks.device, out=None, verbose=False, **kw)
    best = int(-ranks.max_over_ranks(-float(res["best_length"])))
    # lowest rank among those holding the best length
    winner = int(-ranks.max_over_ranks(-float(ranks.rank if res["best_length"] == best else ranks.world)))
    total_scored = ranks.sum_over_ranks(int(res["scored"]))
    res.update(global_best_length=best, winner_rank=winner, restarts=ranks.world, scored_all_ranks=total_scored)
    if ranks.rank == winner:
        if out:
            with open(out, "w") as f:
                f.writelines(res["newick"])
        if verbose:
            ci = res["min_len_tree"] / best
            print("\nSearch Results:")
            print(f"  Restarts (one per GPU):   {ranks.world}")
            print(f"  Candidates scored (GPU):  {total_scored}")
            print(f"  Topologies recovered:     {res['topologies']}")
            print(f"  Tree score:               {best}  (restart {winner})")
            print(f"  Consistency index:        {ci:.2f}")
            print(f"  Homoplasy index:          {1 - ci:.2f}")
            print(f"  Total runtime (seconds):  {res['wall_seconds']:.2f}")
            if out:
                print(f"\nAll topologies written to '{out}'")
    ranks.barrier()
    return res


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m lvb_amd.search")
    ap.add_argument("-i", dest="infile", default="infile")
    ap.add_argument("-o", dest="out", default="outtree")
    ap.add_argument("-s", dest="seed", type=int, default=int(time.time()) % 900000000)
    ap.add_argument("-a", dest="algorithm", type=int, choices=[0, 1, 2], default=1)
    ap.add_argument("-c", dest="cooling", choices=["g", "l"], default="g")
    ap.add_argument("-f", dest="fmt", choices=list(host.FORMATS), default="phylip")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--max-seconds", type=float, default=0.0)
    ap.add_argument("--host-proposals", action="store_true", help="draw neighbours on the host instead of the GPU")
    ap.add_argument("--chains", type=int, default=1, help="independent restarts stepped together on this GPU (1 .. 64)")
    ap.add_argument("--run-levels", type=int, default=None,
                    help="with --chains: runs of up to this many accepted moves per scoring walk while a chain is hot "
                         "(default: 3 for one or two chains, else off)")
    ap.add_argument("--exact", action="store_true", help="reproduce the reference program's run for this seed")
    ap.add_argument("-N", dest="max_trees", type=int, default=0)
    a = ap.parse_args(argv)
    if (a.chains > 1 or a.run_levels is not None) and (a.exact or a.host_proposals or a.max_trees):
        ap.error("--chains / --run-levels run the multi-chain loop, which has no --exact, --host-proposals or -N")
    try:
        if a.exact:
            run_exact(a.infile, a.seed, a.algorithm, 0 if a.cooling == "g" else 1, a.device, a.out,
                      max_trees=max(a.max_trees, 0), fmt=a.fmt)
            return 0
        from .launch import Ranks
        ranks = Ranks()  # WORLD_SIZE > 1 (python -m torch.distributed.run ... -m lvb_amd.search): one restart per GPU
        kw = dict(algorithm=a.algorithm, batch=a.batch, max_seconds=a.max_seconds, cooling=0 if a.cooling == "g" else 1,
                  device_proposals=0 if a.host_proposals else 2, fmt=a.fmt)
        if ranks.world > 1:
            run_restarts(ranks, a.infile, a.seed, a.out, **kw)
            ranks.close()
        elif a.chains > 1 or a.run_levels is not None:
            run_chains(a.infile, a.seed, a.chains, algorithm=a.algorithm, batch=a.batch, device=a.device, out=a.out,
                       max_seconds=a.max_seconds, cooling=0 if a.cooling == "g" else 1, fmt=a.fmt, run_levels=a.run_levels)
        else:
            run(a.infile, a.seed, device=a.device, out=a.out, **kw)
    except (api.LvbGpuError, ValueError, OSError) as exc:
        print(f"\nFATAL ERROR: {exc}")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
