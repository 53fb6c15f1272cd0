// kernels.hpp - launch interface between the C ABI (api_*.cpp) and fitch_kernels.hip
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "program.hpp"
#include "decide.h"

namespace lvbgpu
{

#ifndef LVB_WALK_WAVES
#define LVB_WALK_WAVES 4
#endif
constexpr uint32_t WALK_WAVES = LVB_WALK_WAVES;   // waves per workgroup
constexpr uint32_t WALK_THREADS = 64 * WALK_WAVES;
constexpr uint32_t TILE_WORDS = 128;              // 64 lanes x 16 B = 128 reference words (2048 sites) per wave tile
constexpr uint32_t MAX_LDS_BYTES = 160 * 1024;    // gfx950: 160 KiB per CU

// One tile per wave until a launch has this many (tile, candidate) waves; only beyond it are tiles grouped (up to
// MAX_TILES_PER_WAVE per wave).  A wave that walks several tiles puts as many column slices into its XCD's L2 at the
// same time as it has tiles (the waves of an XCD are at different tiles of their groups): with 5 tiles per wave the
// slices of 500 x 50k (1 MB each) no longer fit the 4 MB L2 - 2.3 GB of L2 misses per launch instead of 33 MB
// (tools/l2_probe5.hip), B = 65 536: 1452 -> 1216 us per launch, 2000 x 200k at B = 4096: 621 -> 460 us.  Round 1 aimed
// for 65 536 waves.
constexpr uint32_t TARGET_WAVES = 1u << 22;
constexpr uint32_t MAX_TILES_PER_WAVE = 8;
constexpr uint64_t FLIP_MIN_BYTES = 64ull << 20;  // tree blocks beyond this alternate the direction of scoring launches

// "length" the device generator gives a candidate it could not represent (per-candidate buffers
// too short, no admissible move): the host turns anything this large into INT64_MAX
constexpr long long PROPOSAL_OVERFLOW_LENGTH = 1ll << 61;
static_assert(PROPOSAL_OVERFLOW_LENGTH == LVB_OVERFLOW_LENGTH, "decide.h restates it for C");
constexpr uint32_t CAND_RESIDENT_BASE = 1u;       // base += *s_all - sum(node_changes[dst])
// CandDesc::flags bits 8..: the CHAIN (resident tree slot) the candidate belongs to.  Programs name nodes by their
// numbers in ONE tree (0..2n-4); the state sets of several resident trees share the leaf rows and keep their internal
// rows one block after the other, so the walk turns node v of chain c into row v + c * chain_rows for v >= bias_from.
constexpr uint32_t CAND_CHAIN_SHIFT = 8;
// watcher launches: bits 48..59 of a length slot count the waves that have added to it (lengths are < 2^47; bit 61 is
// PROPOSAL_OVERFLOW_LENGTH)
constexpr uint32_t WATCH_COUNT_SHIFT = 48;
constexpr uint32_t WATCH_MAX_GROUPS = 2047;
constexpr uint32_t WATCH_WAVES = 8u * WALK_WAVES; // eight extra workgroups
constexpr int32_t MAX_CHAINS = 64;
constexpr uint32_t PICK_NONE = LVB_PICK_NONE; // decide.h: "this chain accepted nothing"

struct CandDesc
{
    uint32_t tok_off, ntok;   // tokens of this candidate in WalkArgs::toks
    uint32_t dst_off, ncomb;  // produced nodes (one per combine) in WalkArgs::dsts
    long long base;           // changes of clean nodes supplied by the host (strict compat) or 0
    uint32_t flags;
    uint32_t nfresh;          // chain starts (tokens with TOK_FRESH) in this program
};
static_assert(sizeof(CandDesc) == 32, "CandDesc layout");

struct WalkArgs
{
    const uint4 *rows_in;       // state-set rows (BIT-PLANE layout), row r at rows_in + r * in_stride4
    uint4 *rows_out;            // COMMIT: where produced sets go (may alias rows_in)
    const uint32_t *toks;
    const int32_t *dsts;
    const CandDesc *cands;
    const long long *node_changes;      // resident per-node changes (CAND_RESIDENT_BASE)
    const long long *s_all;             // resident sum of all internal changes
    unsigned long long *len_out;        // [B], zeroed by the caller
    unsigned long long *changes_out;    // COMMIT: per-node change accumulators, zeroed for dsts
    unsigned long long *s_all_out;      // COMMIT on the resident tree: running sum of all internal changes (may be null)
    // Where a state set lies: group g (16 bytes = 32 sites) of row r in tile t is at
    //     rows + t * tile_stride + r * stride4 + g          (16-byte units)
    // The RESIDENT block is tile-major - [tile][row][64 groups]: stride4 = 64, tile stride = rows x 64 - so that a tile's
    // column slice, which is what the waves of an XCD read together, is ONE contiguous piece of memory (500 x 50k: 1 MB,
    // 2000 x 200k: 4 MB) instead of a thousand 1 KiB pieces a row apart: DRAM pages, TLB entries and L2 sets are used
    // whole (pure-load probe, same reads: 2000 x 200k B = 4096 608 -> 522 us, 500 x 50k B = 16 384 400 -> 372 us).
    // Staging arenas (strict compat) are row-major: stride4 = row length, tile stride = 64.
    uint32_t in_stride4, out_stride4;   // row strides in 16-byte units
    uint64_t in_tile_bytes;             // input tile stride in BYTES (1024 for row-major blocks)
    uint64_t out_tile4;                 // output tile stride in 16-byte units (64 for row-major blocks)
    uint64_t block_bytes;               // size of the block rows_in points into (picks the launch's direction flip)
    uint32_t nrows;                     // rows addressable through rows_in (picks 32- or 64-bit row offsets)
    // several resident trees in one block: rows / change slots of node v of chain c sit at v + c * chain_rows for
    // v >= bias_from (the internal nodes; bias_from = UINT32_MAX turns the mapping off: staging arenas), the root
    // slot of chain c at root_slot + c, its S_all at s_all[4 c]
    uint32_t bias_from, chain_rows;
    uint32_t B, ntiles;
    uint32_t ngroups;                   // tiles are dealt to this many groups; one wave walks one (group, candidate)
    uint32_t nitems;                    // B * ngroups
    // filled by launch_walk so that no wave divides: group g walks tiles_per (+1 if g < tiles_rem)
    // tiles; item / B = mulhi(item, inv_B) or that + 1
    uint32_t tiles_per, tiles_rem, inv_B;
    uint32_t stack_depth;               // operand-stack levels per wave in LDS (>= 1)
    uint32_t root_slot;                 // COMMIT: changes_out slot that collects the two root combines
    // direct steps (small batches, host_len != null): the last wave of the launch to finish copies the B lengths
    // into host-mapped memory, clears the length slots and the counter, and releases *host_flag = step_seq.
    // The host polls that word: no copy engine and no stream wait on the step's critical path.
    unsigned long long *host_len;       // [B] in pinned host memory
    uint32_t *done_count;               // device word, zero between launches
    uint32_t *host_flag;                // pinned host word
    uint32_t step_seq;
    // watcher (big scoring launches of device-built batches, host_len != null): every wave's one atomic adds its
    // partial length AND 1 << WATCH_COUNT_SHIFT to its candidate's slot, so the slot itself says how many of the
    // candidate's ngroups waves have arrived; eight extra workgroups at the END of the grid (dealt last, i.e. into the
    // launch's tail) wait for every slot to be complete and store the lengths into host_len; each of their WATCH_WAVES
    // waves then sets ITS word of host_flag[WATCH_WAVES] to step_seq (the host waits for all of them).  The
    // walking waves pay nothing for it (no returning atomic, no fence, no counter of their own), and a step needs no
    // read-back copy behind the walk.
    uint32_t watcher;
    uint32_t watch_starve; // test hook (LVBGPU_DEBUG_STARVE_WATCHER): the watchers wait for one wave more than there are and look only
                           // a few thousand times - they must give up (flag 0xFFFFFFFF) and the host must say so
    // fused commits (COMMIT, tmp_changes != null): the waves accumulate combine k's changes in tmp_changes[k]
    // (zero between launches); the last wave to finish moves them into changes_out[], keeps *s_all_out current
    // with the difference to what those nodes held before, and clears tmp_changes and done_count again - no
    // separate zeroing launch in front of a commit.
    unsigned long long *tmp_changes;
    // several candidates committed in one launch (one per chain): launch candidate j walks cands[pick_idx[j]]
    // (use_pick == 0: cands[j]), accumulates in tmp_changes + j * tmp_stride and ticks done_count[j]; the last of ITS
    // waves settles that candidate.  A single commit is j = 0, tmp_stride irrelevant.  The picks travel IN the kernel
    // arguments: from pinned host memory every commit wave (and the table rebuild, and the gather) began with a read
    // over the host link.
    uint32_t use_pick;
    uint32_t pick_idx[MAX_CHAINS];
    uint32_t tmp_stride;
    // ... and, from launch candidate n_first on, the programs of a SECOND block (host-built: re-roots, host-made
    // candidates), candidate j being cands2[j - n_first] with tokens / destinations relative to toks2 / dsts2: accepted
    // device moves and re-roots of other chains in ONE commit walk.  n_first = UINT32_MAX (resident_args): no second block
    uint32_t n_first;
    const CandDesc *cands2;
    const uint32_t *toks2;
    const int32_t *dsts2;
    uint32_t flip; // walk each XCD's share of the items from its far end (filled by launch_walk)
    // two candidates per wave (fitch_walk_pair; scoring launches): pairs[2 p], pairs[2 p + 1] = the candidates of pair p
    // (the second PICK_NONE: walked alone); an item is then (tile group, pair) and nitems = npairs * ngroups
    const uint32_t *pairs;
    uint32_t npairs;
    // COMMIT: produced sets and their counts wait in LDS, this many per wave, and go out in bursts (filled by
    // launch_walk, >= 1 for COMMIT).  A store or atomic inside the chain makes every wait for a
    // row a full drain of the memory counter (reads and writes return out of order with respect to each other),
    // which turns the 4-deep load ring into one round trip per token.
    uint32_t defer_slots;
};

// how many tile groups to cut ntiles into for a batch of B candidates
inline uint32_t choose_groups(uint32_t B, uint32_t ntiles, uint32_t target_waves = TARGET_WAVES)
{
    uint32_t per_wave = (uint32_t)(((uint64_t)B * ntiles) / target_waves);
    if (per_wave < 1)
        per_wave = 1;
    if (per_wave > MAX_TILES_PER_WAVE)
        per_wave = MAX_TILES_PER_WAVE;
    return (ntiles + per_wave - 1) / per_wave;
}

// ---- device-side proposals (propose_kernels.hip)
struct lvbgpu_edit_dev
{
    int32_t node, left, right;
};
// a move named by the host (lvbgpu_score_moves): NNI a = internal node u, b = 1 to swap out u's right
// child; SPR a = src, b = dest; TBR a = src, b = dest, c = leaf the moved subtree is re-rooted at
// (-1: as SPR).  Same layout as lvbgpu_move in include/lvbgpu.h.
struct lvbgpu_move_dev
{
    int32_t kind, a, b, c;
};

struct ProposalInfo
{
    int32_t kind;     // 0 NNI, 1 SPR, 2 TBR
    int32_t a, b, c;  // NNI: u,-,- ; SPR: src,dest,- ; TBR: src,dest,x (-1: subtree too small, plain SPR)
    int32_t flag;     // NNI: which child of u was swapped out
    int32_t n_edits;  // edits written (may exceed the stride when overflow is set)
    int32_t overflow; // the move did not fit the fixed strides: its length is meaningless, never accept it
    int32_t ncomb;    // combines of its program (D + 2)
};
// What the generator reads of one topology: one blob of IdxT (uint16_t while 2n-3 <= 65535, else int32_t)
//   parent[nb] left[nb] right[nb] nleaf[nb] depth[nb] tin[nb] first_leaf[nb] leaf_order[leaf_order_len] up[K][nb]
// nleaf = leaves below (1 for a leaf); depth = edges to the root leaf (0 for it); tin = preorder number (root 0,
// a subtree is the interval [tin, tin + 2 nleaf - 1)); leaf_order = the non-root leaves in preorder and
// first_leaf[v] the position of v's first one; up[k][v] = the 2^k-th ancestor of v, the root beyond it.
// Built by the host when the resident topology changes (api_propose.cpp), padded to 16 bytes.
constexpr uint32_t GEN_WAVES = 16; // one copy of the tables into LDS serves 16 candidates at a time (4 waves: 72 % of a
                                    // candidate's time was its workgroup's copy - a thousand workgroups reading the same 34 KB)
constexpr uint32_t GEN_THREADS = 64 * GEN_WAVES;
constexpr uint32_t MAX_GEN_SEGS = 64; // = MAX_CHAINS: one segment of a launch per resident tree
// one run of candidates of ONE resident tree inside a generator launch (36 bytes: 64 of them travel as kernel arguments,
// in the post launch beside the commit walk's and the table rebuild's)
struct GenSeg
{
    uint32_t start, count;   // candidates [start, start + count) of the batch
    uint32_t mix_a, mix_b;
    uint32_t seed_lo, seed_hi; // the draw is a function of (seed, index within the segment)
    int32_t root;
    uint32_t blk_start;      // first workgroup of the launch that works on this segment (filled by the launcher)
    int8_t kind_all;         // 0 NNI, 1 SPR, 2 TBR; -1 / -2 / -3: see propose_kernels.hip
    uint8_t chain;           // its tables: GenArgs::tables + chain * table_stride
    uint8_t wait;            // post launch: this tree's tables are being rebuilt by the same launch - wait for GenArgs::table_ready[chain]
    uint8_t fused;           // post launch: ... and the rebuilding workgroup draws this segment itself, from its LDS (few candidates: no generator workgroups)
};
static_assert(sizeof(GenSeg) == 36, "GenSeg layout");
struct GenArgs
{
    const void *tables;
    uint32_t idx_bytes;
    uint32_t table_stride, table_bytes; // every tree's slot / what of it the tables fill (the same for all trees of these taxa)
    int32_t K;                          // ancestor tables hold 2^0 .. 2^(K-1)
    int32_t n, nb;
    uint32_t leaf_order_len;
    uint32_t stride_t, stride_e;
    uint32_t *toks;
    int32_t *dsts;
    lvbgpu_edit_dev *edits;
    CandDesc *cands;
    ProposalInfo *info;
    unsigned long long *len_out; // [B] length slots of the batch, cleared by the generator
    // two candidates per wave (fitch_walk_pair): every generating workgroup pairs the sixteen candidates it has just
    // drawn among themselves - whose programs END alike the longest, found on the programs' last 64 tokens in LDS - and
    // writes the pairs where the walk looks for them: pairs[2 p], pairs[2 p + 1] (PICK_NONE: walked alone), segment s's
    // pairs behind those of the segments before it (ceil(count / 2) each).  No pass over the batch, no second kernel, no
    // workgroup waiting for another.  null: one candidate per wave
    uint32_t *pairs;
    uint32_t n_gen_blocks;        // filled by the launcher: generating workgroups of the launch
    const lvbgpu_move_dev *moves; // single segment only: candidate b IS moves[b]
    unsigned long long *prof;     // LVBGPU_GEN_PROFILE: [256][8] clock stamps of the first candidates (else null)
    const uint32_t *table_ready;  // post launch: [MAX_CHAINS], = ready_seq once that chain's tables have been rebuilt (GenSeg::wait)
    uint32_t ready_seq;
    uint32_t wait_spins;          // how often a waiting workgroup looks at that word before it gives up (0: 2^24, ~seconds; tests: few)
    int32_t use_lds;              // filled by the launcher
    uint32_t nseg;
    GenSeg seg[MAX_GEN_SEGS];
};
static_assert(sizeof(GenArgs) <= 2600, "GenArgs travels as a kernel argument, in the post launch beside three more structs");
hipError_t launch_propose(const GenArgs &args, hipStream_t stream);

// who walks with whom (GenArgs::pairs): a generating workgroup's LDS beside its tables - the last 64 tokens of its sixteen
// programs (rows of 65 words: different candidates in different banks), their lengths, the 16 x 16 shared lengths
constexpr uint32_t PAIR_ROW = 65;
constexpr uint32_t PAIR_LDS_BYTES = (16 * PAIR_ROW + 16 + 256) * 4;

// rebuild the generator's tables of the picked candidates' chains on the device (one workgroup per pick)
constexpr uint32_t REBUILD_THREADS = 1024;
constexpr uint32_t REBUILD_MAX_NODES = (MAX_LDS_BYTES - 32) / 24; // four int32 arrays + one uint2 array in LDS; larger trees keep the host path
// a move that is not a candidate of a device batch (a re-root named by the host): its chain, its new root and its
// rewrites ext_edits[edit_off .. edit_off + n_edits)
struct RebuildExt
{
    int32_t chain, new_root, edit_off, n_edits;
};
struct RebuildArgs
{
    void *tables;            // all chains' slots
    uint32_t table_stride, idx_bytes;
    int32_t n, nb, K;
    uint32_t leaf_order_len;
    // workgroup j < n_pick rebuilds for the picked candidate at batch position pick_idx[j] (its descriptor names the
    // chain, its rewrites are edits + pick_idx[j] * stride_e); workgroup j >= n_pick for ext[j - n_pick] (a move that is
    // not a candidate of a device batch: a re-root named by the host)
    uint32_t n_pick;
    uint32_t rebuild_picks;        // 0: the picked candidates' workgroups only send the records to the host (their tables cannot follow on the device)
    uint32_t pick_idx[MAX_CHAINS]; // (in the kernel arguments, see WalkArgs)
    const CandDesc *cands;   // the picked candidates' descriptors (flags carry the chain)
    const ProposalInfo *info;
    const lvbgpu_edit_dev *edits;
    uint32_t stride_e;
    const RebuildExt *ext;
    const lvbgpu_edit_dev *ext_edits;
    // post launch: the generator of the SAME launch waits for a chain's tables - they are written through (agent-scope
    // stores) and table_ready[chain] = ready_seq says when (null: plain stores, the next launch reads them)
    uint32_t *table_ready;
    uint32_t ready_seq;
    uint32_t withhold_ready;  // test hook (LVBGPU_DEBUG_WITHHOLD_READY): the words are never set - whoever waits must give up
    uint64_t wait_mask;      // the chains a generator workgroup of this launch waits for (the others' tables leave as plain stores)
    // the new tables are made in LDS (in the layout the generator reads) and copied out in one piece; a chain whose next
    // draw is small is drawn by the rebuilding workgroup itself, straight from there (GenSeg::fused).  0: trees whose
    // tables do not fit LDS beside the rebuild's own arrays - every entry is stored where it belongs as it is computed
    uint32_t stage_tables, table_bytes;
    unsigned long long *prof; // LVBGPU_POST_PROFILE: [8 workgroups][8] clock stamps of the rebuild's phases (else null)
};
// what lvbgpu_chains_commit needs on the host of every picked candidate, written straight into pinned memory
// (gather.hpp) by wave 0 of that pick's workgroup of the post launch, before it rebuilds the chain's tables
struct GatherArgs
{
    uint32_t pick_idx[MAX_CHAINS]; // batch positions of the picked candidates (in the kernel arguments)
    uint32_t k;
    const ProposalInfo *info;
    const lvbgpu_edit_dev *edits;
    uint32_t stride_e;
    char *out;                     // pinned host memory: out + j * out_stride = [ProposalInfo][n_edits rewrites]
    uint32_t out_stride;
    uint32_t *flag;                // pinned: = seq once all k records are on the host
    uint32_t seq;
    uint32_t *arrived;             // device word, zero between launches
};

// The POST launch (propose_kernels.hip): what lies between two scoring walks of an annealing step - the table rebuilds
// of the chains that moved, the commit walk of those moves, and the NEXT step's generator - as roles of ONE launch.
struct PostArgs
{
    GenArgs gen;      // nseg == 0: no generator role
    WalkArgs commit;  // nitems == 0: no commit walk (launch_post shapes it: tiles per wave, burst slots)
    RebuildArgs reb;
    GatherArgs gat;   // k == 0: nothing goes to the host
    uint32_t n_reb;   // rebuild workgroups: reb.n_pick picked candidates, then n_reb - reb.n_pick moves named by the host
    uint32_t n_cblk;  // filled by launch_post
    unsigned long long *prof; // LVBGPU_POST_PROFILE: [1 + 4 x workgroups] clock stamps of the last launch (else null): count, then {role, start, mid, end}
};
static_assert(sizeof(PostArgs) <= 4000, "PostArgs travels as a kernel argument");
// can the generator ride in a post launch (its tables must fit LDS beside nothing else)?
bool post_can_generate(const GenArgs &g);
hipError_t launch_post(const PostArgs &args, hipStream_t stream);
// fills what a walk's launcher owes the kernel (tiles per group, the division constant, and for a commit walk the burst
// slots that fit `lds_budget` bytes per workgroup of `nwaves` waves beside the operand stacks); *lds_out = dynamic LDS
hipError_t shape_walk(WalkArgs &a, bool commit, uint32_t nwaves, size_t lds_budget, size_t *lds_out);
bool walk_needs_wide(const WalkArgs &a); // row offsets as 64-bit bytes (tree blocks of 64 GiB and more; LVBGPU_WIDE_OFFSETS=1)
hipError_t upload_iupac_table();
hipError_t raise_lds_limit();
hipError_t launch_walk(const WalkArgs &a, bool commit, hipStream_t stream, uint32_t *flip_state = nullptr);
// bounded busy kernel for the wait-limit test (lvbgpu_debug_stall): keeps `stream` busy for about `ms` milliseconds
hipError_t launch_stall(hipStream_t stream, uint32_t ms);
hipError_t launch_zero_changes(unsigned long long *changes, const int32_t *dsts, uint32_t n, unsigned long long *root_slot,
                               unsigned long long *len_slot, unsigned long long *s_all, uint32_t bias_from, uint32_t row_bias,
                               hipStream_t stream);
// scalars[0] = sum of changes[first, last), scalars[1] = scalars[0] + changes[root_slot] (tree length)
hipError_t launch_sum_changes(const unsigned long long *changes, uint32_t first, uint32_t last, uint32_t root_slot,
                              long long *scalars, hipStream_t stream);
hipError_t launch_fill_pad(uint64_t *rows, uint32_t nrows, uint32_t nwords, uint32_t stride_words,
                           uint32_t first_full_row, hipStream_t stream);
// reference nibble layout <-> device bit-plane layout, in place, rows [0, nrows)
hipError_t launch_relayout(uint4 *rows, uint32_t nrows, uint32_t stride4, bool to_planes, hipStream_t stream);
// rows [0, nrows) of a row-major block (stride4 = ntiles * 64) -> rows [0, nrows) of a tile-major block of total_rows rows
hipError_t launch_rows_to_tiles(const uint4 *src, uint4 *dst, uint32_t nrows, uint32_t total_rows, uint32_t ntiles, hipStream_t stream);
// one bit-plane row of the tile-major resident block -> nibble layout, contiguous, in `out`
hipError_t launch_export_row(const uint4 *rows, uint32_t row, uint32_t total_rows, uint32_t ntiles, uint4 *out, hipStream_t stream);
// ceiling probe of the walk's memory path: one launch; *loads_out = 1 KiB row loads it issues
// (stride4 == 64: the block is tile-major, as the resident block is; else row-major with that row stride)
hipError_t launch_l2_probe(const uint4 *rows, uint32_t stride4, uint32_t nrows, uint32_t ntiles, uint32_t ngroups,
                           uint32_t B, uint32_t ntok, int ring, uint4 *sink, uint64_t *loads_out, hipStream_t stream);
hipError_t launch_encode_text(const uint8_t *text, uint32_t n, uint64_t m, uint32_t nwords, uint32_t stride_words,
                              uint64_t *rows, unsigned long long *bad, hipStream_t stream);

} // namespace lvbgpu
