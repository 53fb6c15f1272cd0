/* lvbhost.h - C ABI of the host-side search mirror (lvb_amd/liblvbhost.so).
 *
 * The reference's search host (Solve.c Anneal, TreeOperations.c mutate_*) stays on the CPU.  This
 * library is our own C++ restatement of just enough of it to drive and measure the device path
 * behind include/lvbgpu.h: a topology object, NNI/SPR/TBR proposal generators that emit EDITS
 * (not tree copies), a re-root, and a batched simulated-annealing loop with the reference's
 * acceptance rule and cooling schedule.  It contains no scoring code: every length comes from
 * lvbgpu_* calls.
 *
 * Reference interfaces mirrored (file:line under the reference's src/):
 *   TreeOperations.c:160-209  mutate_nni   -> lvbhost_propose(kind 0)
 *   TreeOperations.c:236-335  mutate_spr   -> lvbhost_propose(kind 1)
 *   TreeOperations.c:337-541  mutate_tbr   -> lvbhost_propose(kind 2)
 *   TreeOperations.c:576-656  lvb_reroot / arbreroot -> lvbhost_reroot_edits
 *   TreeOperations.c:799-811  PullRandomTree -> lvbhost_tree_random
 *   Solve.c:144-479           Anneal       -> lvbhost_anneal
 *   StartingTemperature.c:49-195            -> lvbhost_starting_temperature
 */
#ifndef LVBHOST_H
#define LVBHOST_H

#include <stdint.h>

#include "lvbgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lvbhost_tree lvbhost_tree;

enum lvbhost_move
{
    LVBHOST_NNI = 0,
    LVBHOST_SPR = 1,
    LVBHOST_TBR = 2
};

/* ---- topology object (own random stream: xorshift64*) --------------------------------- */
lvbhost_tree *lvbhost_tree_random(int32_t n, uint64_t seed);
lvbhost_tree *lvbhost_tree_from_arrays(int32_t n, const int32_t *left, const int32_t *right, int32_t root,
                                       uint64_t seed);
void lvbhost_tree_free(lvbhost_tree *t);
int32_t lvbhost_tree_n(const lvbhost_tree *t);
int32_t lvbhost_tree_root(const lvbhost_tree *t);
void lvbhost_tree_arrays(const lvbhost_tree *t, int32_t *parent, int32_t *left, int32_t *right);
void lvbhost_tree_reseed(lvbhost_tree *t, uint64_t seed);

/* one random move of `kind` as edits against the tree (tree unchanged); returns the number of
 * edits written (<= cap) or a negative lvbgpu status */
int lvbhost_propose(lvbhost_tree *t, int kind, lvbgpu_edit *edits, int32_t cap);
/* B random moves; edit_offsets[B+1]; returns total edits or negative status.  kind -1 = cycle
 * NNI, SPR, TBR */
int lvbhost_propose_batch(lvbhost_tree *t, int kind, int32_t B, int32_t *edit_offsets, lvbgpu_edit *edits,
                          int32_t cap);
/* deterministic forms */
int lvbhost_nni_edits(const lvbhost_tree *t, int32_t u, int swap_right, lvbgpu_edit *edits, int32_t cap);
int lvbhost_spr_edits(const lvbhost_tree *t, int32_t src, int32_t dest, lvbgpu_edit *edits, int32_t cap);
int lvbhost_tbr_edits(const lvbhost_tree *t, int32_t src, int32_t dest, int32_t newroot_leaf, lvbgpu_edit *edits,
                      int32_t cap);
int lvbhost_reroot_edits(const lvbhost_tree *t, int32_t newroot, lvbgpu_edit *edits, int32_t cap);
/* apply edits (+ new root, or -1) to the tree: what SwapTrees achieves after an accept */
int lvbhost_tree_apply(lvbhost_tree *t, const lvbgpu_edit *edits, int32_t n_edits, int32_t new_root);

/* ---- alignment input / tree output (reference MSAInput.cpp:274-432 read_phylip,
 *      TreeOperations.c:1156-1220 ur_print) ------------------------------------------- */
typedef struct lvbhost_alignment lvbhost_alignment;
/* PHYLIP, sequential or interleaved: header "n m", 10-character name field, digits and blanks
 * inside sequences ignored, text upper-cased (MSAInput.cpp:829).  NULL + message on error. */
lvbhost_alignment *lvbhost_alignment_read_phylip(const char *path, char *err, int32_t errcap);
/* any of the reference's four input formats (-f; DataStructure.h:50-53: 0 phylip, 1 fasta, 2 nexus, 3 clustal),
 * followed by the conditions its reader puts on all of them (MSAInput.cpp:780-849: at least two sequences, equal
 * lengths, upper case, accepted characters).  FASTA / NEXUS / CLUSTAL are parsed by an own record-stream parser
 * (chunks appended to the taxon of that name); results equal the reference's reader on what it accepts. */
lvbhost_alignment *lvbhost_alignment_read(const char *path, int format, char *err, int32_t errcap);
void lvbhost_alignment_free(lvbhost_alignment *a);
int64_t lvbhost_alignment_n(const lvbhost_alignment *a);
int64_t lvbhost_alignment_m(const lvbhost_alignment *a);
const char *lvbhost_alignment_row(const lvbhost_alignment *a, int64_t i);
const char *lvbhost_alignment_name(const lvbhost_alignment *a, int64_t i);
/* the tree as one line of bracketed text in the reference's unrooted form:
 * "(rootname,<left subtree>,<right subtree>);\n", names right-trimmed, no branch lengths.
 * Returns the number of bytes written (excluding the terminator) or a negative status. */
int64_t lvbhost_tree_newick(const lvbhost_tree *t, const char *const *names, char *out, int64_t cap);

/* ---- alignment preparation (reference matchange, DataOperations.c:272-405, 53-105) ---- */
/* keep[k] = 1 for columns whose raw characters are not all equal to row 0's (constchar); returns
 * the number kept.  The caller drops the others before encoding, as cutcols does. */
int64_t lvbhost_variable_columns(int64_t n, int64_t m, const char *const *rows, uint8_t *keep);
/* MinimumTreeLength: sum over columns of (#distinct characters other than - ? N X) - 1, 5 when
 * more than MAXSTATES (5) distinct ones occur */
int64_t lvbhost_min_tree_length(int64_t n, int64_t m, const char *const *rows);

/* the distinct best topologies the last lvbhost_anneal run found (the reference's treestack, Treestack.c:231-306;
 * found by hash, told apart by an exact comparison of canonical forms): how many there are - all are kept, _count
 * and _kept agree - and their arrays */
/* rooting- and numbering-independent identity of the topology (sum of hashed bipartition keys) */
uint64_t lvbhost_tree_topology_hash(const lvbhost_tree *t);
/* the exact identity behind it: the tree re-rooted at taxon 0, every node's subtrees ordered by their smallest
 * taxon, in preorder (-1 opens an internal node, a taxon number is a leaf): 2n-3 entries.  Equal for two trees iff
 * they are the same unrooted topology.  Returns the number of entries written or a negative status. */
int32_t lvbhost_tree_canonical(const lvbhost_tree *t, int32_t *out, int32_t cap);
int32_t lvbhost_tree_best_count(const lvbhost_tree *t);
int32_t lvbhost_tree_best_kept(const lvbhost_tree *t);
int lvbhost_tree_best_get(const lvbhost_tree *t, int32_t i, int32_t *left, int32_t *right, int32_t *root);

/* ---- program introspection (what the device will walk), for tests --------------------- */
/* mode 0: edits relative to the tree, mode 1: whole tree (all dirty), mode 2: explicit dirty flags.
 * Writes tokens/dsts (caps in elements); returns 0 or a negative lvbgpu status. */
int lvbhost_program(const lvbhost_tree *t, int mode, const lvbgpu_edit *edits, int32_t n_edits, int32_t new_root,
                    const uint8_t *dirty_flags, uint32_t *toks, int32_t tok_cap, int32_t *ntok, int32_t *dsts,
                    int32_t dst_cap, int32_t *ndst, int32_t *max_stack, int32_t *n_dirty);

/* ---- batched simulated annealing over the device path --------------------------------- */
typedef struct
{
    uint64_t seed;
    int32_t algorithm;        /* 0: alternate NNI/SPR (reference -a 0, Solve.c:288-297);
                                 1: TBR with probability t/t0, else NNI/SPR (-a 1, Solve.c:421-426);
                                 2: probabilities from the three move counters (-a 2, Solve.c:253-259, 452-466;
                                    refreshed per batch);
                                 10/11/12: NNI only / SPR only / TBR only */
    int32_t cooling_schedule; /* 0 geometric (0.99^n t0), 1 linear (Solve.c:409-443) */
    int32_t batch;            /* candidates scored per device step (speculative) */
    int32_t reroot_interval;  /* REROOT_INTERVAL, LVB.h:99 (1000); 0 = never */
    double t0;                /* starting temperature; <= 0: lvbhost_starting_temperature() */
    int64_t maxaccept;        /* LVB.h:146-150: 5 */
    int64_t maxpropose;       /* 2000 */
    int64_t maxfail;          /* 40 */
    int64_t min_len_tree;     /* MinimumTreeLength of the alignment (energy scale, Solve.c:233,303) */
    int64_t max_proposals;    /* stop after this many consumed proposals (0 = until frozen) */
    double max_seconds;       /* stop after this wall time (0 = no limit) */
    int64_t max_device_steps; /* stop after this many batches (0 = no limit) */
    int32_t sync_every;       /* every this many device steps: RCCL min-reduce of the best length (0 = never).
                                 Lockstep mode: every rank must then run exactly max_device_steps (> 0)
                                 batches, so the other stop criteria are ignored */
    int32_t log_cap;          /* capacity of log_seconds/log_best */
    int32_t device_proposals; /* 0: neighbours are drawn and programmed by this library on the host;
                                 1: on the GPU (lvbgpu_propose_score*); 2: on the GPU for steps of
                                 >= 1024 candidates, on the host for smaller ones (default) */
    int32_t run_levels;       /* lvbhost_anneal_chains: while a chain accepts most of what it sees (>= 0.30 acceptances per
                                 proposal; back to device draws below 0.15) its candidates are drawn by the host's
                                 generators and are CUMULATIVE - level d holds alternatives drawn on the tree the
                                 first alternatives of levels 1 .. d-1 leave - so that one scoring walk advances the
                                 chain by a run of accepted moves instead of one; the hot chains of a step share one
                                 scoring walk and one commit walk.  0: off (default), n >= 1:
                                 that many levels (3 is a good choice; 1: host-drawn, one move per step).  The
                                 trajectory does not depend on n >= 1, nor on the chains beside it; the value of the
                                 first chain's parameters counts for all.  Pays for one or two chains per context
                                 (500 x 50 000: 0.36 -> 0.28 s for one, 0.43 -> 0.38 s for two); even at four, a loss
                                 beyond: the host draws and builds ~10 us of programs per chain and step */
    int32_t lanes;            /* lvbhost_anneal_chains: the chains are dealt to this many LANES - contexts of their own
                                 (lvbgpu_fork: own stream and batches) that ONE host thread serves in turn, whichever
                                 lane's lengths have arrived - so that one lane's scoring walk covers another's host work
                                 and post launch.  0: automatic (2 from 16 chains on, else 1; LVBHOST_LANES overrides).
                                 A chain's trajectory does not depend on it.  The first chain's value counts for all;
                                 lockstep runs (sync_every > 0) keep one lane */
} lvbhost_anneal_params;

typedef struct
{
    int64_t start_length;
    int64_t best_length;
    int64_t final_length;
    int64_t global_best_length; /* after the last RCCL min-reduce (== best_length without a communicator) */
    int64_t scored;             /* candidates scored on the device */
    int64_t consumed;           /* proposals the serial-equivalent chain consumed ("rearrangements evaluated") */
    int64_t accepted;           /* accepted moves (commits) */
    int64_t topologies;         /* distinct topologies of the best length ("Topologies recovered") */
    int64_t device_steps;       /* batches launched */
    int64_t reroots;
    int64_t dirty_nodes;        /* sum of D over all scored candidates */
    int64_t temperatures;       /* cooling steps taken */
    double t_final;
    double seconds;
    double seconds_device;      /* inside lvbgpu_* calls */
    int32_t n_log;              /* entries written to the log arrays */
    int32_t frozen;             /* 1 if the freezing criterion ended the run */
    double seconds_done;        /* wall time (from the run's start) at which THIS chain stopped (froze or ran out of proposals) */
    /* lvbhost_anneal_chains, in every chain's result: the run while at least half of its chains were still annealing -
     * its wall time and the candidates scored in it.  One chain in some dozens has its starting temperature land on the
     * reference's second 1e-5 increment (StartingTemperature.c:173), accepts most of what it sees and needs ~100x longer
     * to freeze; a rate measured until the LAST chain has frozen then says how long that chain took, not what the
     * scorer does. */
    double seconds_busy;
    int64_t scored_busy;
    int64_t host_steps;         /* of device_steps: host-drawn steps with cumulative candidates (run_levels > 0) */
} lvbhost_anneal_result;

void lvbhost_anneal_defaults(lvbhost_anneal_params *p);
/* Starts from (and updates) `tree`, which must be the tree resident in ctx (lvbgpu_set_tree).
 * log_seconds/log_best receive one (wall time, best length) pair per improvement. */
int lvbhost_anneal(lvbgpu_ctx *ctx, lvbhost_tree *tree, const lvbhost_anneal_params *params,
                   lvbhost_anneal_result *result, double *log_seconds, int64_t *log_best);
int lvbhost_starting_temperature(lvbgpu_ctx *ctx, lvbhost_tree *tree, const lvbhost_anneal_params *params,
                                 double *t0_out);
/* R independent chains (restarts) on one GPU, stepped together: every device step draws, scores and commits for all
 * of them at once (lvbgpu_chains_propose_score / lvbgpu_chains_commit), so a step's latency is paid once instead of
 * R times.  trees[c] is chain c's start tree (made resident in slot c here; the context gets R slots if it has another
 * number), params[c] its parameters (own seed; t0 <= 0: estimated per chain as StartingTemperature() does, in the
 * same steps), results[c] its outcome; neighbours are always drawn on the device.  A chain's trajectory depends on its
 * own parameters only, not on R.  params[0] supplies what concerns the whole run: max_seconds, max_device_steps,
 * sync_every (lockstep RCCL min-reduce of the best length over ranks, as lvbhost_anneal) and log_cap of the shared
 * log (wall time, best length over all chains).  1 <= R <= 64. */
int lvbhost_anneal_chains(lvbgpu_ctx *ctx, int32_t R, lvbhost_tree *const *trees, const lvbhost_anneal_params *params,
                          lvbhost_anneal_result *results, double *log_seconds, int64_t *log_best, int32_t *n_log);
/* make `tree` resident in ctx (full evaluation) */
int lvbhost_tree_upload(lvbgpu_ctx *ctx, const lvbhost_tree *tree, int64_t *length_out);

/* ---- the reference's own trajectory on the device scorer ----------------
 * Same seed and options as the reference program => same start trees, starting temperature,
 * proposals, decisions, cooling, re-roots and treestack, hence the same "Rearrangements
 * evaluated", "Tree score", "Topologies recovered" and output trees.  Pieces first (so each can be
 * checked against the compiled reference on its own), then the whole search. */
typedef struct lvbhost_refrng lvbhost_refrng; /* uni()/randpint(), RandomNumberGenerator.c:87-256 */
int lvbhost_refrng_new(lvbhost_refrng **out, int32_t seed /* rinit: 0..900000000 */);
void lvbhost_refrng_free(lvbhost_refrng *r);
double lvbhost_refrng_uni(lvbhost_refrng *r);
int64_t lvbhost_refrng_randpint(lvbhost_refrng *r, int64_t upper);
/* PullRandomTree (TreeOperations.c:799-811): child arrays [2n-3] of a tree rooted at taxon 0 */
int lvbhost_ref_random_tree(lvbhost_refrng *r, int32_t n, int32_t *left, int32_t *right);
/* mutate_nni/spr/tbr (kind 0/1/2) of `tree` with the reference's draws, as edits */
int lvbhost_ref_propose(const lvbhost_tree *tree, lvbhost_refrng *r, int kind, lvbgpu_edit *edits, int32_t cap,
                        int32_t *n_edits);
/* the same in two halves: draw only the move's parameters (all of the stream consumption), and turn
 * parameters into edits (no draws).  The parameters are what lvbgpu_score_moves takes. */
int lvbhost_ref_draw_move(const lvbhost_tree *tree, lvbhost_refrng *r, int kind, lvbgpu_move *out);
int lvbhost_move_edits(const lvbhost_tree *tree, const lvbgpu_move *move, lvbgpu_edit *edits, int32_t cap,
                       int32_t *n_edits);
/* arbreroot (TreeOperations.c:639-656) as edits + the new root */
int lvbhost_ref_arbreroot(const lvbhost_tree *tree, lvbhost_refrng *r, lvbgpu_edit *edits, int32_t cap,
                          int32_t *n_edits, int32_t *new_root);

typedef struct
{
    int32_t seed;             /* -s */
    int32_t algorithm;        /* -a 0 | 1 | 2 (Solve.c:251-298, 452-466) */
    int32_t cooling_schedule; /* -c g (0) | l (1) */
    int32_t max_batch;        /* ceiling on proposals drawn ahead per device step (< 1000) */
    int64_t min_len_tree;     /* MinimumTreeLength of the alignment */
    int64_t max_trees;        /* -N: stop when the treestack holds this many (0 = keep all) */
    int64_t maxaccept, maxpropose, maxfail; /* 5, 2000, 40 */
    int64_t device_moves_min; /* batches at least this long are scored from their 16-byte move parameters
                                 (lvbgpu_score_moves: the device builds the programs); shorter ones from
                                 host-built programs (lvbgpu_score_batch).  0: default (128); < 0: never */
    int64_t reserved[3];
} lvbhost_refsearch_params;

typedef struct
{
    double t0;                  /* "SA Starting Temperature" */
    int64_t rearrangements;     /* "Rearrangements evaluated" (Anneal's iteration count) */
    int64_t best_length;        /* "Tree score" */
    int64_t trees;              /* "Topologies recovered" */
    int64_t start_length, final_length;
    int64_t accepted_moves, reroots, temperatures;
    int64_t st_rearrangements;  /* proposals consumed by the starting-temperature search */
    int64_t scored, device_steps;       /* candidates scored / lvbgpu_score_batch calls, overall */
    int64_t st_scored, st_device_steps; /* ... of which while finding t0 */
    int64_t device_move_steps;          /* steps scored through lvbgpu_score_moves */
    double t_final, seconds, seconds_device;
} lvbhost_refsearch_result;

void lvbhost_refsearch_defaults(lvbhost_refsearch_params *p);
/* ctx holds the alignment (constant columns already cut).  *tree_out receives a new tree handle
 * (free with lvbhost_tree_free) holding the final tree and, as its kept best trees in treestack
 * order (lvbhost_tree_best_get), the trees the reference would print. */
int lvbhost_reference_search(lvbgpu_ctx *ctx, const lvbhost_refsearch_params *params, lvbhost_refsearch_result *result,
                             lvbhost_tree **tree_out);

#ifdef __cplusplus
}
#endif
#endif /* LVBHOST_H */
