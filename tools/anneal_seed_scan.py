"""Which seed bases give a 32-chain annealing run of the bench's alignment WITHOUT a straggler?  (One chain in sixty has
its starting temperature land on the reference's second 1e-5 increment, accepts 87 % of what it sees and needs ~100x
longer to freeze - faithful to StartingTemperature.c, DESIGN.md section 7c - and a throughput measured until ALL chains
have frozen then says nothing about the scorer.)  Prints seconds, frozen chains and the smallest cooling-step count."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lvb_amd import api, host
from tests.synth import treelike_rows

n, m, R = 500, 50000, 32
rows, min_len = host.prepare_alignment(treelike_rows(n, m, 3))
for base in [int(a) for a in sys.argv[1:]] or range(3, 11):
    ctx = api.FitchContext(text_rows=rows)
    trees = [host.HostTree(n, seed=(base * 1000 + 1) * 100 + c) for c in range(R)]
    ps = []
    for c in range(R):
        p = host.anneal_defaults()
        p.seed = base * 7919 + c + 1; p.algorithm = 11; p.batch = 4096; p.t0 = 0.0; p.min_len_tree = min_len
        p.max_seconds = 2.5; p.log_cap = 16
        ps.append(p)
    res, log = host.anneal_chains(ctx, trees, ps)
    print(f"seed base {base}: {max(r['seconds'] for r in res):.2f} s, frozen {sum(r['frozen'] for r in res)}, "
          f"min temperatures {min(r['temperatures'] for r in res)}, scored/s {sum(r['scored'] for r in res) / max(r['seconds'] for r in res) / 1e6:.1f} M, "
          f"while half the chains anneal: {res[0]['seconds_busy']:.2f} s, {res[0]['scored_busy'] / max(res[0]['seconds_busy'], 1e-9) / 1e6:.1f} M/s", flush=True)
    for t in trees:
        t.close()
    ctx.close()
