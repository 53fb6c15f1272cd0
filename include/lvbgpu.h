/* lvbgpu.h - C ABI of the MI355X-native Fitch parsimony scoring path for LVB.
 *
 * This is the drop-in boundary for ONE path of phylolvb/lvb: tree evaluation (getplen) and the
 * packed state-set data it drives.  The search host (SA schedule, NNI/SPR/TBR proposals,
 * treestack) stays on the CPU and calls through this header; everything behind it is
 * hand-written HIP for gfx950 (lvb_amd/csrc/).  No CPU fallback exists: every entry point that
 * needs the device returns LVBGPU_E_NODEVICE / LVBGPU_E_HIP when it is missing.
 *
 * Reference interfaces replaced (file:line under the reference's src/):
 *   LVB.h:187  long getplen(Dataptr, TREESTACK_TREE_NODES*, Parameters, long root, long*, long*, int*)
 *              -> lvbgpu_getplen_compat()            (strict, stateless, B = 1)
 *              -> lvbgpu_set_tree()/lvbgpu_score_batch()/lvbgpu_commit()  (resident, batched)
 *   LVB.h:188-189 alloc_memory_to_getplen / free_memory_to_getplen -> nothing to allocate; the
 *              C++ adapter (lvb_amd/csrc/getplen_adapter.cpp) keeps them for link compatibility
 *   LVB.h:235  DNAToBinary(Dataptr, Lvb_bit_length**)  -> lvbgpu_encode_text()/lvbgpu_create_from_text()
 *   LVB.h:228  words_per_row(long)                     -> lvbgpu_words_per_row()
 *   LVB.h:222  ss_init(Dataptr, tree, enc_mat)         -> lvbgpu_create() (leaf rows become resident)
 *   TreeOperations.c:88-103 make_dirty_below + the "sitestate[0]==0 means dirty" convention
 *              -> candidates are *edits* (node, new left, new right); the library derives the
 *                 dirty set (edited nodes + their ancestors below the root) itself
 *
 * Data conventions (identical to the reference):
 *   - n taxa, nodes 0..n-1 are leaves (leaf i = taxon i), n..2n-4 internal; every index is in
 *     the tree; the tree is rooted at a leaf `root` whose record holds the two top children.
 *   - a state-set row is nwords uint64 words, 16 sites per word, nibble k of word j = site
 *     16j+k, bit0=A bit1=C bit2=G bit3=T (LVB.h:73-76); positions >= m hold 0xF.
 *   - children of a leaf other than the root are LVBGPU_UNSET (-1).
 *
 * Error convention: every function returns LVBGPU_OK (0) or a negative code and never exits;
 * the reference's "FATAL ERROR ... exit(1)" convention (Error.c:49-67) is reproduced only by
 * the C++ adapter that carries the reference's own getplen signature.
 *
 * Threading: a context is bound to one device and owns one HIP stream; it is not thread-safe
 * (the reference's callers are single-threaded and non-reentrant).
 */
#ifndef LVBGPU_H
#define LVBGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LVBGPU_UNSET (-1)

enum lvbgpu_status
{
    LVBGPU_OK = 0,
    LVBGPU_E_ARG = -1,       /* bad argument (null pointer, size out of range) */
    LVBGPU_E_NODEVICE = -2,  /* no usable HIP device / HIP runtime not functional */
    LVBGPU_E_HIP = -3,       /* a HIP call failed; see lvbgpu_last_error() */
    LVBGPU_E_NOMEM = -4,     /* host or device allocation failed */
    LVBGPU_E_STATE = -5,     /* call order violated (e.g. score before set_tree) */
    LVBGPU_E_TOPOLOGY = -6,  /* arrays/edits do not describe a binary tree rooted at a leaf */
    LVBGPU_E_SYMBOL = -7,    /* alignment text holds a symbol DNAToBinary rejects */
    LVBGPU_E_ZEROLEN = -8,   /* computed length <= 0 (the reference asserts > 0, TreeEvaluation.c:267) */
    LVBGPU_E_COMM = -9       /* RCCL communicator failure */
};

typedef struct lvbgpu_ctx lvbgpu_ctx;
typedef struct lvbgpu_batch lvbgpu_batch;

/* One child-pair rewrite relative to the resident current tree: node `node` gets children
 * (left, right).  An NNI is 2 edits, an SPR 3, a TBR 3 + the re-rooted path inside the moved
 * subtree, a re-root the path between old and new root (the old root leaf gets (-1,-1)). */
typedef struct
{
    int32_t node;
    int32_t left;
    int32_t right;
} lvbgpu_edit;

/* what one batch costs (for the roofline line): all counts are per launch of the batch */
typedef struct
{
    int64_t candidates;   /* B */
    int64_t combines;     /* sum over candidates of (D + 2) word-combines per word */
    int64_t rows_read;    /* sum over candidates of (D + 3) clean rows read */
    int64_t dirty_nodes;  /* sum over candidates of D */
    int64_t max_stack;    /* deepest operand stack any candidate needs */
    int64_t algorithmic_bytes; /* rows_read * nwords * 8 */
} lvbgpu_batch_stats;

/* ---- library / device --------------------------------------------------------------- */
const char *lvbgpu_strerror(int status);
const char *lvbgpu_last_error(const lvbgpu_ctx *ctx); /* text of the last failing HIP/RCCL call */
int lvbgpu_device_count(void);                        /* >= 0, or a negative status */
int lvbgpu_abi_version(void);

/* ---- encoding (DataOperations.c:164-249, 446-467) ------------------------------------ */
long lvbgpu_words_per_row(long m);
/* text rows (upper-case, m chars each) -> packed rows on the device, copied to out[n][nwords].
 * Returns LVBGPU_E_SYMBOL for a character the reference would crash() on (235-241). */
int lvbgpu_encode_text(int device, long n, long m, const char *const *rows, uint64_t *out);

/* ---- context: alignment resident in HBM ---------------------------------------------- */
/* leaf_matrix: host pointer, n rows of nwords words, row i at leaf_matrix + i*row_stride_words */
int lvbgpu_create(lvbgpu_ctx **out, int device, long n, long nwords, const uint64_t *leaf_matrix,
                  long row_stride_words);
int lvbgpu_create_from_text(lvbgpu_ctx **out, int device, long n, long m, const char *const *rows);
/* a second context on the same alignment and device (the leaf rows are copied on the device; one tree slot, no tree):
 * own stream, own batches - what two groups of chains need whose device work is to overlap (lvbhost_anneal_chains) */
int lvbgpu_fork(lvbgpu_ctx *src, lvbgpu_ctx **out);
void lvbgpu_destroy(lvbgpu_ctx *ctx);
long lvbgpu_n(const lvbgpu_ctx *ctx);
long lvbgpu_nwords(const lvbgpu_ctx *ctx);

/* ---- several resident trees (chains) in one context --------------------------------------
 * Independent annealing chains on one GPU share the alignment: lvbgpu_set_chains(ctx, R) gives the context R
 * tree slots (state sets of the leaves once, R blocks of internal node sets; any resident tree is dropped),
 * lvbgpu_select_chain says which slot the single-tree calls below (set_tree, score_batch, commit,
 * propose_score, current_length, get_*) mean.  A fresh context has one slot, selected. */
int lvbgpu_set_chains(lvbgpu_ctx *ctx, int32_t nchains); /* 1 .. 64 */
int lvbgpu_select_chain(lvbgpu_ctx *ctx, int32_t chain);
int32_t lvbgpu_chains(const lvbgpu_ctx *ctx);

/* ---- resident current tree ------------------------------------------------------------ */
/* Full evaluation (every internal node recomputed: getplen's all-dirty case after ss_init /
 * lvb_reroot, TreeEvaluation.c:204-264) of the tree given by child arrays of 2n-3 entries;
 * node sets and per-node `changes` become resident.  *length_out = tree length. */
int lvbgpu_set_tree(lvbgpu_ctx *ctx, const int32_t *left, const int32_t *right, int32_t root,
                    int64_t *length_out);
int lvbgpu_current_length(lvbgpu_ctx *ctx, int64_t *length_out);
int lvbgpu_get_topology(lvbgpu_ctx *ctx, int32_t *parent, int32_t *left, int32_t *right, int32_t *root);
int lvbgpu_get_changes(lvbgpu_ctx *ctx, int64_t *changes /* [2n-3]; leaves hold 0 */);
int lvbgpu_get_sets(lvbgpu_ctx *ctx, int32_t node, uint64_t *out /* [nwords] */);

/* ---- batched incremental scoring -------------------------------------------------------
 * Candidate b is the current tree with edits[edit_offsets[b] .. edit_offsets[b+1]) applied and
 * rooted at roots[b] (roots == NULL or roots[b] < 0: same root).  Length semantics are
 * getplen's incremental case: only dirty nodes are recomputed, clean nodes contribute their
 * cached `changes` (TreeEvaluation.c:191-202); nothing resident is modified.  lengths_out[b]
 * belongs to candidate b whatever the library does inside: small steps come back without the
 * copy engine, big batches are walked longest program first and in pieces that overlap program
 * building with the device (DESIGN.md sections 3 and 7a).  Programs are built by up to
 * LVBGPU_THREADS host threads, which spin for 0.3 ms after a call before they sleep.   */
int lvbgpu_score_batch(lvbgpu_ctx *ctx, int32_t B, const int32_t *edit_offsets,
                       const lvbgpu_edit *edits, const int32_t *roots, int64_t *lengths_out);

/* the same in three steps, so a batch can stay resident and be re-launched (benchmarks) or be
 * overlapped with host work: build (host: dirty sets + postorder programs; H2D), launch
 * (asynchronous on the context's stream), lengths (synchronise + D2H). */
int lvbgpu_batch_build(lvbgpu_ctx *ctx, int32_t B, const int32_t *edit_offsets,
                       const lvbgpu_edit *edits, const int32_t *roots, lvbgpu_batch **out);
int lvbgpu_batch_launch(lvbgpu_ctx *ctx, lvbgpu_batch *batch);
int lvbgpu_batch_lengths(lvbgpu_ctx *ctx, lvbgpu_batch *batch, int64_t *lengths_out); /* LVBGPU_E_STATE before a launch */
int lvbgpu_batch_get_stats(const lvbgpu_batch *batch, lvbgpu_batch_stats *out);
void lvbgpu_batch_free(lvbgpu_batch *batch);

/* ---- neighbourhoods generated on the device ----------------------------------------------
 * B random neighbours of the resident tree are DRAWN, turned into programs and scored on the
 * GPU (no per-candidate host work, no program upload).  kind: 0 NNI, 1 SPR, 2 TBR (the rules of
 * mutate_nni/spr/tbr, TreeOperations.c:160-541), -1 = candidate b gets kind b % 3.  The draw is a
 * function of (seed, b) only.  lengths_out[b] is INT64_MAX for the rare candidate whose move
 * does not fit the library's fixed per-candidate buffers (never accept those).
 * lvbgpu_proposal_edits() returns candidate b of the LAST lvbgpu_propose_score() call as edits
 * (for lvbgpu_commit and for the host's own topology mirror); info4, if not NULL, receives
 * {kind, a, b, c}: NNI {0, u, swapped-right?, -}, SPR {1, src, dest, -}, TBR {2, src, dest, leaf x}. */
int lvbgpu_propose_score(lvbgpu_ctx *ctx, int32_t B, int32_t kind, uint64_t seed, int64_t *lengths_out);
/* the reference's two move schedules: parity >= 0: candidate b is an SPR if (parity + b) is odd, else
 * an NNI (-a 0, Solve.c:288-297); parity < 0: each candidate draws NNI with probability p_nni, SPR
 * with p_spr, TBR otherwise (-a 1, Solve.c:262-283) */
int lvbgpu_propose_score_mixed(lvbgpu_ctx *ctx, int32_t B, double p_nni, double p_spr, int64_t parity,
                               uint64_t seed, int64_t *lengths_out);
int lvbgpu_proposal_edits(lvbgpu_ctx *ctx, int32_t b, lvbgpu_edit *edits, int32_t cap, int32_t *n_edits,
                          int32_t *info4);
/* ---- several chains in one step ----------------------------------------------------------------
 * What lvbgpu_select_chain + lvbgpu_propose_score* would do chain by chain, in ONE generator launch, ONE walk
 * and ONE read-back: draws[i] asks for `count` neighbours of chain `chain`'s resident tree (kind as
 * lvbgpu_propose_score: 0 / 1 / 2, -1 = candidate j gets kind j % 3; -2 = SPR if (mix_a + j) is odd else NNI;
 * -3 = drawn per candidate, NNI below mix_a, SPR below mix_b, else TBR, thresholds scaled to 2^32).  Candidate j
 * of a chain is a function of (seed, j) only - the same move, the same length as the single-tree call gives
 * with that seed.  lengths_out holds the chains' candidates one chain after the other, in the order of draws[].
 * Chains must be distinct.  An annealing host keeps R independent chains busy with one step's latency. */
typedef struct
{
    int32_t chain, count, kind;
    uint32_t mix_a, mix_b;
    uint64_t seed;
} lvbgpu_chain_draw;
int lvbgpu_chains_propose_score(lvbgpu_ctx *ctx, int32_t k, const lvbgpu_chain_draw *draws, int64_t *lengths_out);
/* the same in two halves, so that the host can work on one batch while the device works on another: _submit enqueues
 * everything up to the lengths' read-back and returns at once, _collect waits for THAT batch alone and hands the
 * lengths over.  Two batches may be in flight, in slots 0 and 1 (the plain call uses slot 0); the second one's
 * neighbours are drawn while the first one is being scored.  A chain may be in both as long as its tree does not
 * change in between: a commit makes the other batch's candidates of that chain stale (picking them is
 * LVBGPU_E_STATE).  lvbgpu_chains_commit picks from the batch collected last. */
int lvbgpu_chains_submit(lvbgpu_ctx *ctx, int32_t slot, int32_t k, const lvbgpu_chain_draw *draws);
int lvbgpu_chains_collect(lvbgpu_ctx *ctx, int32_t slot, int64_t *lengths_out);
/* accept candidate `b` (index within its chain's draw of the LAST lvbgpu_chains_propose_score /
 * lvbgpu_propose_score* call) for each listed chain, at most one per chain: the candidates' own device-built
 * programs are walked in commit form (one launch for all picks), and the chains' topologies follow (the moves'
 * rewrites are fetched from the device).  Asynchronous like lvbgpu_commit(.., NULL): the lengths are known
 * from scoring.  A pick of a candidate that overflowed (INT64_MAX) is LVBGPU_E_ARG.
 * COLLECTED, not launched: what this call, lvbgpu_chains_reroot and lvbgpu_chains_commit_edits ask for goes to the device
 * as ONE launch (commit walk + rebuild of the chains' generator tables + the moves' records to the host) with the NEXT
 * call on the context - with a submit into the OTHER batch slot that batch's generator rides in the same launch (take the
 * two slots in turn); every other call lets the device catch up first, so no call ever sees a tree half moved. */
typedef struct
{
    int32_t chain, b;
} lvbgpu_chain_pick;
int lvbgpu_chains_commit(lvbgpu_ctx *ctx, int32_t k, const lvbgpu_chain_pick *picks);
/* re-root the listed chains (distinct) at the given leaves in one commit walk: arbreroot (TreeOperations.c:639-656)
 * for several chains at once; what lvbgpu_select_chain + lvbgpu_commit(rewrites along the old-root .. new-root path,
 * new_root, NULL) does chain by chain.  Asynchronous (a re-root does not change the length). */
typedef struct
{
    int32_t chain, new_root;
} lvbgpu_chain_root;
int lvbgpu_chains_reroot(lvbgpu_ctx *ctx, int32_t k, const lvbgpu_chain_root *reqs);
/* fn(i, arg) for i in [0, n) on the context's host threads (the ones that build programs; LVBGPU_THREADS), the caller
 * among them; returns when all are done.  For host loops that prepare many chains' candidates between two calls. */
typedef void (*lvbgpu_task_fn)(int32_t index, void *arg);
int lvbgpu_parallel_for(lvbgpu_ctx *ctx, int32_t n, lvbgpu_task_fn fn, void *arg);
/* Host-made candidates of several chains in ONE walk: candidate b is a set of child-pair rewrites of chain
 * chain_of[b]'s resident tree - one move, or the cumulative rewrites of a run of moves (any set that gives a tree);
 * edit_offsets[B + 1] index `edits`.  What lvbgpu_select_chain + lvbgpu_score_batch does chain by chain
 * (TreeEvaluation.c:191-264 per candidate).  Keep one chain's candidates adjacent. */
int lvbgpu_chains_score_edits(lvbgpu_ctx *ctx, int32_t B, const int32_t *chain_of, const int32_t *edit_offsets,
                              const lvbgpu_edit *edits, int64_t *lengths_out);
/* ... and accept one such candidate per listed chain (distinct chains) in ONE commit walk, the generator's tables
 * following on the device: what lvbgpu_select_chain + lvbgpu_commit(n, edits, -1, NULL) does chain by chain (SwapTrees
 * after getplen, Solve.c:323/370).  Asynchronous: the caller scored the candidates and knows the lengths. */
int lvbgpu_chains_commit_edits(lvbgpu_ctx *ctx, int32_t k, const int32_t *chains, const int32_t *edit_offsets,
                               const lvbgpu_edit *edits);
/* the rewrites of pick j of the LAST lvbgpu_chains_commit (for a host that mirrors the topologies): already on the
 * host, no device access */
int lvbgpu_chains_picked_edits(lvbgpu_ctx *ctx, int32_t j, lvbgpu_edit *edits, int32_t cap, int32_t *n_edits);

/* ---- a whole annealing step in one submission: the accept decision rides with the batch ----------------------------
 * lvbgpu_chains_step_submit is lvbgpu_chains_submit plus, per draw, the RULE by which that chain accepts (Solve.c:303-378):
 * candidate j of the draw (in order) is taken if its length is <= cur_length, or with probability
 * exp(-deltah / temperature), deltah = min_len_tree / cur_length - min_len_tree / length capped at 1, and never once
 * -deltah < temperature * log(1e-11); its uniform draw is a function of (accept_seed, j).  The chain's pick is the FIRST
 * taken candidate; the library decides at the collect (lvb_amd/csrc/decide.h: one function for the library and the scorer's
 * test double) and commits the picks (lvbgpu_chains_commit's work).
 * lvbgpu_chains_step_collect hands over the lengths (as lvbgpu_chains_collect) and picks_out[i] = index of the accepted
 * candidate within draw i, or -1; lvbgpu_chains_step_edits(i) = the accepted move's rewrites (waits for them if they are
 * still on their way).  A slot holds one step at a time; a step and plain batches may not be in flight together. */
typedef struct
{
    int64_t cur_length;
    double temperature;
    double min_len_tree;
    uint64_t accept_seed;
} lvbgpu_chain_rule;
int lvbgpu_chains_step_submit(lvbgpu_ctx *ctx, int32_t slot, int32_t k, const lvbgpu_chain_draw *draws,
                              const lvbgpu_chain_rule *rules);
int lvbgpu_chains_step_collect(lvbgpu_ctx *ctx, int32_t slot, int64_t *lengths_out, int32_t *picks_out);
/* *ready = 1 if that slot's submitted batch (or step) has its lengths on the host - the collect would not wait -, else 0;
 * never waits: for a host that serves several contexts and takes whichever is done first */
int lvbgpu_chains_ready(lvbgpu_ctx *ctx, int32_t slot, int32_t *ready);
int lvbgpu_chains_step_edits(lvbgpu_ctx *ctx, int32_t i, lvbgpu_edit *edits, int32_t cap, int32_t *n_edits);

/* counts of the LAST device-built batch (as lvbgpu_batch_get_stats gives for host-built ones; candidates that
 * overflowed are left out): reads the batch's descriptors back - for measurement, not for the search */
int lvbgpu_proposal_stats(lvbgpu_ctx *ctx, lvbgpu_batch_stats *out);
/* the same with the moves NAMED by the host (a search that draws its own neighbours, e.g. with the
 * reference's random stream, but wants no host work per candidate beyond 16 bytes): the device turns
 * each move into its child-pair rewrites and its program, as mutate_nni/spr/tbr + the dirty marking
 * would, and scores it.  kind 0 NNI: a = internal node u, b = 1 to give away u's right child (0: left;
 * TreeOperations.c:184-205); kind 1 SPR: a = src, b = dest (256-327); kind 2 TBR: a = src, b = dest,
 * c = leaf of src's subtree it is re-rooted at, not a child of src (-1, or a subtree of <= 2 leaves: as SPR;
 * 436-513).  A move the reference's generators could not have made is LVBGPU_E_TOPOLOGY.  Afterwards
 * lvbgpu_proposal_edits(ctx, b, ...) gives candidate b's rewrites for lvbgpu_commit. */
typedef struct
{
    int32_t kind, a, b, c;
} lvbgpu_move;
int lvbgpu_score_moves(lvbgpu_ctx *ctx, int32_t B, const lvbgpu_move *moves, int64_t *lengths_out);

/* B whole topologies scored from the leaf rows alone (every internal node recomputed, nothing
 * resident read or written): left/right are [B][2n-3]. */
int lvbgpu_score_full_batch(lvbgpu_ctx *ctx, int32_t B, const int32_t *left, const int32_t *right,
                            const int32_t *roots, int64_t *lengths_out);

/* Accept a candidate (SwapTrees after getplen, Solve.c:323/370): apply the edits to the
 * resident tree, recompute its dirty nodes' sets and per-node changes in place.
 * length_out == NULL makes the commit asynchronous: it is only enqueued (ordered before any later
 * call on this context) and nothing is read back - for callers that know the length already
 * because they scored the candidate. */
int lvbgpu_commit(lvbgpu_ctx *ctx, int32_t n_edits, const lvbgpu_edit *edits, int32_t root,
                  int64_t *length_out);

/* ---- strict compatibility: the reference's own tree block ------------------------------
 * `tree` is the reference's BranchArray: 2n-3 records of 40 bytes {long parent,left,right,
 * changes; uint64_t *sitestate} (LVB.h:121-128).  Dirty == sitestate[0]==0.  Recomputes the
 * dirty nodes on the device and writes their sets and `changes` back into the host block, so
 * the caller's next treecopy/getplen sees exactly what the reference would have left there.
 * Stateless with respect to the resident tree (operand rows are uploaded per call). */
int lvbgpu_getplen_compat(lvbgpu_ctx *ctx, void *tree, long root, int64_t *length_out);

/* ---- timing on the context's stream (HIP events) -------------------------------------- */
int lvbgpu_timer_start(lvbgpu_ctx *ctx);
int lvbgpu_timer_stop(lvbgpu_ctx *ctx, float *elapsed_ms); /* synchronises */
int lvbgpu_synchronize(lvbgpu_ctx *ctx);
void *lvbgpu_stream(lvbgpu_ctx *ctx); /* hipStream_t, for interop */
/* Where the library polls for the device itself (a step's lengths, a watcher's flags, the picked moves of a commit)
 * it gives up after this many seconds and the call returns LVBGPU_E_HIP with the stream's state in
 * lvbgpu_last_error(): a stuck stream does not become an endless host spin.  Default 30 s (or LVBGPU_WAIT_SECONDS
 * when the context is created).  What was waited for is lost: synchronise and resubmit, or destroy the context. */
int lvbgpu_set_wait_limit(lvbgpu_ctx *ctx, double seconds);
/* test hook: keeps the context's stream busy for about `ms` milliseconds (1 .. 2000) with a kernel that only watches
 * the clock, so that what is enqueued behind it cannot complete before then */
int lvbgpu_debug_stall(lvbgpu_ctx *ctx, int32_t ms);
/* diagnostic (LVBGPU_POST_PROFILE set): out[0] = workgroups of the last post launch, then {role (1 table rebuild, 2 commit
 * walk, 3 generator), start, -, end} of each of the first 1000 (100 MHz clock), then 8 x 8 stamps of the first
 * rebuilding workgroups' phases: tools/post_profile.py */
int lvbgpu_debug_post_stamps(lvbgpu_ctx *ctx, unsigned long long *out4065);
/* diagnostic: who walked with whom in the last device-built batch of `slot` (LVBGPU_PAIR): out[2 p], out[2 p + 1] = the
 * candidates of pair p (0xFFFFFFFF: walked alone); *npairs = 0: that batch was walked one candidate per wave */
int lvbgpu_debug_pairs(lvbgpu_ctx *ctx, int32_t slot, uint32_t *out, int32_t cap_pairs, int32_t *npairs);
/* test hook: counters of what results cannot show (they are the same either way): scoring walks launched two
 * candidates per wave (LVBGPU_PAIR=n when the context was created; fitch_walk_pair);
 * lvbgpu_chains_commit_edits calls that walked the SCORED programs of the last lvbgpu_chains_score_edits call
 * instead of building the accepted candidates' programs again; post launches (commit walk + table rebuilds of the
 * chains' accepted moves and re-roots in one launch), and those of them in which the next batch's generator rode along */
#define LVBGPU_COUNT_PAIRED_WALKS 0
#define LVBGPU_COUNT_COMMITS_REUSING_PROGRAMS 1
#define LVBGPU_COUNT_POST_LAUNCHES 2
#define LVBGPU_COUNT_POST_LAUNCHES_WITH_GENERATOR 3
int lvbgpu_debug_count(lvbgpu_ctx *ctx, int32_t what, int64_t *count);

/* per-kernel timing for the roofline line: while enabled (every > 0), every `every`-th scoring walk the
 * library launches (batch launches, lvbgpu_score_batch, lvbgpu_propose_score*, lvbgpu_score_moves) is
 * bracketed by HIP events on the context's stream - sampled, because a pair of events between launches
 * costs the step ~20 us; _read waits for the walks in flight and returns the sum of the timed walks' durations
 * and their number since timing was enabled.  every = 0 turns it off.  Commit walks are not counted. */
int lvbgpu_walk_timing(lvbgpu_ctx *ctx, int every);
int lvbgpu_walk_timing_read(lvbgpu_ctx *ctx, double *total_ms, int64_t *launches);
/* what the memory path the walk is bound by delivers on this device, measured now: a pure-load kernel with
 * the walk's launch geometry and access pattern (one wave per (tile, candidate), XCD-aware order,
 * rows_per_wave pseudo-random rows of the resident block per tile, 4 or 8 loads of 1 KiB in flight,
 * one XOR per 16 bytes) over B candidates, `reps` launches; *gb_per_s = the better of the two. */
int lvbgpu_probe_l2(lvbgpu_ctx *ctx, int32_t B, int32_t rows_per_wave, int32_t reps, double *gb_per_s);

/* ---- multi-GPU: best length over independent restarts (one context per process/GPU) ----
 * id is an opaque 128-byte RCCL unique id created on rank 0 and distributed by the launcher. */
/* LVBGPU_OK if the RCCL library can be loaded in this process (call it on every rank and agree
 * before any rank enters lvbgpu_comm_init, which is collective) */
int lvbgpu_comm_available(void);
int lvbgpu_comm_unique_id(void *id128);
int lvbgpu_comm_init(lvbgpu_ctx *ctx, int nranks, int rank, const void *id128);
int lvbgpu_comm_size(const lvbgpu_ctx *ctx); /* ranks of the context's communicator; 1 without one */
/* values of 2^47 and more (a rank that has no length yet passes INT64_MAX) take part as "no length": they lose
 * against every real length, come back as INT64_MAX if no rank had one, and then *argmin_rank = -1 */
int lvbgpu_allreduce_min(lvbgpu_ctx *ctx, int64_t *value /* in: local best, out: global best */,
                         int32_t *argmin_rank /* may be NULL */);
/* the other sharding of the path (the reference's OpenMP site slices, TreeEvaluation.c:95-97): rank r
 * of k creates its context from columns [m r / k, m (r + 1) / k) of the alignment (cut at multiples of 2048 sites for
 * whole tiles); every rank scores the SAME candidates on its slice, and since a length is a sum over sites the
 * candidates' lengths are the sums of the ranks' values: values[count] in, sums out (one RCCL sum per step).  For one
 * chain on an alignment too large for one GPU to score fast enough; restarts (lvbgpu_allreduce_min) remain the
 * primary way to use several GPUs. */
int lvbgpu_allreduce_sum(lvbgpu_ctx *ctx, int64_t *values, int32_t count);
int lvbgpu_comm_destroy(lvbgpu_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* LVBGPU_H */
