"""Host logic under AddressSanitizer + UBSan (CPU only: the pool has no GPU sanitizers).  Driven by
tests/manual/sanitize_host.sh, which builds the host sources against the scorer's test double
(tests/cpu_double) with -fsanitize=address,undefined and preloads the runtimes: whole
reference-trajectory runs, the program builder under random moves, the alignment readers."""
import sys, json, ctypes as C, numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent.parent))
from lvb_amd import host
from oracle import binding
from pathlib import Path
lib = host.bind(C.CDLL(sys.argv[1]))
lib.lvbgpu_double_new.restype = C.c_void_p
lib.lvbgpu_double_new.argtypes = [C.c_long, C.c_long, np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")]
lib.lvbgpu_double_free.argtypes = [C.c_void_p]
GOLD = Path(__file__).resolve().parent.parent / 'golden'
cases = json.loads((GOLD / 'ref_trajectories.json').read_text())['cases']
for case in cases:
    if case['expect']['rearrangements'] > 100000: continue
    names, rows = host.read_alignment(GOLD / 'ref_tests' / case['infile'], lib=lib)
    rows, ml = host.prepare_alignment(rows, lib)
    enc = binding.encode_rows(rows)
    ctx = C.c_void_p(lib.lvbgpu_double_new(enc.shape[0], enc.shape[1], np.ascontiguousarray(enc)))
    p = host.refsearch_defaults(lib); p.seed, p.algorithm, p.min_len_tree = case['seed'], case['algorithm'], ml
    p.cooling_schedule = 0 if case['cooling'] == 'g' else 1; p.max_trees = case['max_trees']; p.device_moves_min = -1
    res, tree = host.reference_search(ctx, p, lib)
    assert res['rearrangements'] == case['expect']['rearrangements'], (case, res)
    n = len(tree.best_trees())
    tree.close(); lib.lvbgpu_double_free(ctx)
    print(case['infile'], case['seed'], case['algorithm'], 'ok', n)
# the multi-chain annealing loop (anneal_chains.cpp) on the double's chains: 5 chains, every move schedule
from tests import synth
for alg, levels in ((0, 0), (1, 0), (2, 0), (0, 3), (1, 3), (2, 4)):   # levels > 0: runs of acceptances (host-drawn hot phase)
    rows5, ml5 = host.prepare_alignment(synth.treelike_rows(20, 500, 33), lib)
    enc5 = binding.encode_rows(rows5)
    ctx5 = C.c_void_p(lib.lvbgpu_double_new(enc5.shape[0], enc5.shape[1], np.ascontiguousarray(enc5)))
    trees5 = [host.HostTree(20, seed=50 + c, lib=lib) for c in range(5)]
    pars5 = []
    for c in range(5):
        q = host.anneal_defaults(lib); q.seed, q.algorithm, q.batch, q.t0, q.min_len_tree = 90 + c, alg, 48, 0.0, ml5
        q.max_proposals, q.log_cap, q.run_levels = 2500, 32, levels
        pars5.append(q)
    res5, log5 = host.anneal_chains(ctx5, trees5, pars5, lib=lib)
    assert all(r['consumed'] == 2500 for r in res5), res5
    for t5 in trees5: t5.close()
    lib.lvbgpu_double_free(ctx5)
    assert levels == 0 or sum(r['host_steps'] for r in res5) > 0
    print('anneal_chains', alg, levels, [r['best_length'] for r in res5], 'ok')
# program builder fuzz through the sanitized library
t = host.HostTree(40, seed=3, lib=lib)
for k in range(3000):
    e = t.propose(k % 3)
    t.program(mode=0, edits=e)
    if k % 7 == 0: t.apply(e)
print('fuzz ok')
for f, fmt in (("stock_100x1000.fas", "fasta"), ("stock_100x1000.nex", "nexus"), ("stock_100x1000.aln", "clustal"),
               ("lib_phylip_interleaved.phy", "phylip"), ("blackbox/test_min_m_2.infile", "phylip")):
    try:
        names, rows = host.read_alignment(GOLD / 'ref_tests' / f, fmt, lib)
        print(f, len(rows), len(rows[0]))
    except ValueError as exc:
        print(f, 'rejected:', str(exc).splitlines()[0])
print('readers ok')
# the readers on damaged input: every format's stock file with random bytes changed, lines dropped, doubled or cut,
# and the file truncated - the parser may refuse (ValueError) but must not read or write out of bounds
import random, tempfile, os
rnd = random.Random(5)
tried = refused = 0
with tempfile.TemporaryDirectory() as td:
    for f, fmt in (("stock_100x1000.fas", "fasta"), ("stock_100x1000.nex", "nexus"), ("stock_100x1000.aln", "clustal"),
                   ("lib_phylip_interleaved.phy", "phylip")):
        data = (GOLD / 'ref_tests' / f).read_bytes()
        for k in range(150):
            b = bytearray(data)
            how = k % 5
            if how == 0:
                for _ in range(rnd.randint(1, 30)):
                    b[rnd.randrange(len(b))] = rnd.choice(b"ACGT-?N;:>#\n\t 0123456789[]'\"=()")
            elif how == 1:
                b = b[:rnd.randrange(1, len(b))]
            else:
                lines = bytes(b).split(b"\n")
                i = rnd.randrange(len(lines))
                if how == 2: del lines[i]
                elif how == 3: lines.insert(i, lines[i])
                else: lines[i] = lines[i][:rnd.randrange(0, len(lines[i]) + 1)]
                b = bytearray(b"\n".join(lines))
            path = os.path.join(td, "damaged")
            with open(path, "wb") as fh: fh.write(bytes(b))
            tried += 1
            try:
                names, rows = host.read_alignment(path, fmt, lib)
                assert len({len(r) for r in rows}) == 1 and len(names) == len(rows)
            except ValueError:
                refused += 1
print(f'damaged files: {tried} tried, {refused} refused, none crashed')
