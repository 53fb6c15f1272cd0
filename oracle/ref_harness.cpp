/* oracle/ref_harness.cpp - TEST INFRASTRUCTURE ONLY.
 *
 * A thin extern "C" door into the *real* reference (phylolvb/lvb v4.2), compiled
 * by oracle/Makefile from the sources where they lie under /root/reference/src.
 * Nothing here restates reference logic: every function below only *calls* the
 * reference (LVB.h prototypes) and copies results into plain arrays so that
 * Python (ctypes) can (a) pin oracle/fitch_oracle.c against the reference and
 * (b) generate the golden fixtures under tests/golden/ (tests/golden/gen_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the
 * library built from this file (oracle/_ref/liblvbref.so).  The product path
 * (lvb_amd/) never does.
 *
 * Call order follows the reference's own wiring:
 *   Main.c:86-101   phylip_dna_matrin -> matchange -> calc_distribution_processors -> rinit
 *   Solve.c:510-544 treealloc, DNAToBinary, PullRandomTree, ss_init
 *   Solve.c:199-200 alloc_memory_to_getplen, getplen
 *   Solve.c:240-300 arbreroot, mutate_{nni,spr,tbr}, getplen, SwapTrees
 */
#include "LVB.h"
#include "DataOperations.h"
#include "MemoryOperations.h"

#include <new>

namespace {

struct RefHandle
{
    DataStructure msa;
    Parameters rc;
    Lvb_bit_length **enc;        /* [n][nwords] */
    TREESTACK_TREE_NODES *tree[2]; /* 0 = current, 1 = proposed */
    long root[2];
    long *todo;
    long *todo_sum;
    int *runs;
};

void finish_setup(RefHandle *h, int seed, int nproc)
{
    memset(&h->rc, 0, sizeof(h->rc));
    h->rc.seed = seed;
    h->rc.n_processors_available = nproc;
    h->rc.verbose = LVB_FALSE;
    h->rc.cooling_schedule = 0;
    h->rc.algorithm_selection = 0;

    matchange(&h->msa, h->rc);                    /* Main.c:93 */
    calc_distribution_processors(&h->msa, h->rc); /* Main.c:95 */
    rinit(seed);                                  /* Main.c:101 */

    h->enc = (Lvb_bit_length **)malloc(h->msa.n * sizeof(Lvb_bit_length *));
    for (long i = 0; i < h->msa.n; i++)
        h->enc[i] = (Lvb_bit_length *)alloc(h->msa.bytes, "state sets");
    DNAToBinary(&h->msa, h->enc); /* Solve.c:516-519 */

    h->tree[0] = treealloc(&h->msa, LVB_TRUE);
    h->tree[1] = treealloc(&h->msa, LVB_TRUE);
    alloc_memory_to_getplen(&h->msa, &h->todo, &h->todo_sum, &h->runs);
    PullRandomTree(&h->msa, h->tree[0]); /* Solve.c:537 */
    ss_init(&h->msa, h->tree[0], h->enc); /* Solve.c:538 */
    h->root[0] = 0;
    treecopy(&h->msa, h->tree[1], h->tree[0], LVB_TRUE);
    h->root[1] = 0;
}

} // namespace

extern "C" {

/* Build a reference data structure from n upper-case rows of m characters. */
void *refh_new_from_rows(long n, long m, const char *const *rows, int seed, int nproc)
{
    RefHandle *h = new (std::nothrow) RefHandle();
    if (!h)
        return NULL;
    memset(&h->msa, 0, sizeof(h->msa));
    h->msa.n = n;
    h->msa.m = m;
    h->msa.original_m = m;
    h->msa.numberofpossiblebranches = 2 * n - 3; /* CommandLineParser.cpp:66 (brcnt) */
    h->msa.nsets = n - 3;
    h->msa.mssz = n - 2;
    h->msa.max_length_seq_name = 10;
    h->msa.row = (char **)malloc(n * sizeof(char *));
    h->msa.rowtitle = (char **)malloc(n * sizeof(char *));
    for (long i = 0; i < n; i++)
    {
        h->msa.row[i] = (char *)malloc(m + 1);
        memcpy(h->msa.row[i], rows[i], m);
        h->msa.row[i][m] = '\0';
        h->msa.rowtitle[i] = (char *)malloc(16);
        snprintf(h->msa.rowtitle[i], 16, "t%ld", i);
    }
    finish_setup(h, seed, nproc);
    return h;
}

/* Read an alignment with the reference's own reader (Wrapper.c:49). */
void *refh_new_from_file(const char *path, int fmt, int seed, int nproc)
{
    RefHandle *h = new (std::nothrow) RefHandle();
    if (!h)
        return NULL;
    memset(&h->msa, 0, sizeof(h->msa));
    phylip_dna_matrin((char *)path, fmt, &h->msa);
    finish_setup(h, seed, nproc);
    return h;
}

void refh_free(void *vh)
{
    RefHandle *h = (RefHandle *)vh;
    free_memory_to_getplen(&h->todo, &h->todo_sum, &h->runs);
    free(h->tree[0]);
    free(h->tree[1]);
    for (long i = 0; i < h->msa.n; i++)
        free(h->enc[i]);
    free(h->enc);
    delete h;
}

/* dims[0..7] = n, m (after constant-column cut), nwords, nbranches, min_len_tree,
 * n_threads_getplen, n_slice_size_getplen, original_m */
void refh_dims(void *vh, long *dims)
{
    RefHandle *h = (RefHandle *)vh;
    dims[0] = h->msa.n;
    dims[1] = h->msa.m;
    dims[2] = h->msa.nwords;
    dims[3] = h->msa.numberofpossiblebranches;
    dims[4] = h->msa.min_len_tree;
    dims[5] = h->msa.n_threads_getplen;
    dims[6] = h->msa.n_slice_size_getplen;
    dims[7] = h->msa.original_m;
}

/* force the thread split getplen() will use (1 = serial branch) */
void refh_set_threads(void *vh, int nthreads)
{
    RefHandle *h = (RefHandle *)vh;
    Parameters rc = h->rc;
    rc.n_processors_available = nthreads;
    free_memory_to_getplen(&h->todo, &h->todo_sum, &h->runs);
    if (nthreads <= 1)
    {
        h->msa.n_threads_getplen = 1;
        h->msa.n_slice_size_getplen = 0;
    }
    else
        calc_distribution_processors(&h->msa, rc);
    alloc_memory_to_getplen(&h->msa, &h->todo, &h->todo_sum, &h->runs);
}

void refh_row_text(void *vh, long i, char *out) /* m chars, after the cut */
{
    RefHandle *h = (RefHandle *)vh;
    memcpy(out, h->msa.row[i], h->msa.m);
}

void refh_enc_row(void *vh, long i, uint64_t *out)
{
    RefHandle *h = (RefHandle *)vh;
    memcpy(out, h->enc[i], h->msa.bytes);
}

void refh_reseed(void *vh, int seed)
{
    (void)vh;
    rinit(seed);
}

/* the global random stream itself (RandomNumberGenerator.c:87, 235) */
double refh_uni(void) { return uni(); }
long refh_randpint(long upper) { return randpint(upper); }

/* new random start tree into slot 0 (Solve.c:543-544) */
void refh_random_tree(void *vh)
{
    RefHandle *h = (RefHandle *)vh;
    PullRandomTree(&h->msa, h->tree[0]);
    ss_init(&h->msa, h->tree[0], h->enc);
    h->root[0] = 0;
}

/* make tree `which` the topology given by the arrays (the node numbering is the caller's), rooted at leaf
 * `root`: only the records' scalars are written here; every internal node is marked dirty the way the
 * reference does it (sitestate[0] = 0, TreeOperations.c:49-59), the leaves' sets are re-initialised by the
 * reference's ss_init, and the caller's next refh_getplen is the reference's full evaluation */
void refh_set_topology(void *vh, int which, const long *parent, const long *left, const long *right, long root)
{
    RefHandle *h = (RefHandle *)vh;
    TREESTACK_TREE_NODES *t = h->tree[which];
    for (long i = 0; i < h->msa.numberofpossiblebranches; i++)
    {
        t[i].parent = parent[i];
        t[i].left = left[i];
        t[i].right = right[i];
        t[i].changes = 0;
    }
    ss_init(&h->msa, t, h->enc);
    h->root[which] = root;
}

long refh_root(void *vh, int which) { return ((RefHandle *)vh)->root[which]; }

long refh_getplen(void *vh, int which)
{
    RefHandle *h = (RefHandle *)vh;
    return getplen(&h->msa, h->tree[which], h->rc, h->root[which], h->todo, h->todo_sum, h->runs);
}

/* kind: 0 NNI, 1 SPR, 2 TBR ; proposed(1) <- mutate(current(0)) as Solve.c:262-297 */
void refh_mutate(void *vh, int kind)
{
    RefHandle *h = (RefHandle *)vh;
    h->root[1] = h->root[0];
    if (kind == 0)
        mutate_nni(&h->msa, h->tree[1], h->tree[0], h->root[0]);
    else if (kind == 1)
        mutate_spr(&h->msa, h->tree[1], h->tree[0], h->root[0]);
    else
        mutate_tbr(&h->msa, h->tree[1], h->tree[0], h->root[0]);
}

void refh_swap(void *vh) /* accept: Solve.c:323 */
{
    RefHandle *h = (RefHandle *)vh;
    SwapTrees(&h->tree[0], &h->root[0], &h->tree[1], &h->root[1]);
}

long refh_arbreroot(void *vh) /* Solve.c:242 */
{
    RefHandle *h = (RefHandle *)vh;
    h->root[0] = arbreroot(&h->msa, h->tree[0], h->root[0]);
    return h->root[0];
}

/* topology + per-node changes + dirty flag (sitestate[0]==0) of tree `which` */
void refh_get_tree(void *vh, int which, long *parent, long *left, long *right, long *changes, int *dirty)
{
    RefHandle *h = (RefHandle *)vh;
    const TREESTACK_TREE_NODES *t = h->tree[which];
    for (long i = 0; i < h->msa.numberofpossiblebranches; i++)
    {
        parent[i] = t[i].parent;
        left[i] = t[i].left;
        right[i] = t[i].right;
        changes[i] = t[i].changes;
        dirty[i] = (t[i].sitestate[0] == 0U);
    }
}

void refh_get_sets(void *vh, int which, long node, uint64_t *out)
{
    RefHandle *h = (RefHandle *)vh;
    memcpy(out, h->tree[which][node].sitestate, h->msa.bytes);
}

/* raw tree block, so a layout-compatible getplen (oracle or adapter) can run on it in place */
void *refh_tree_block(void *vh, int which) { return ((RefHandle *)vh)->tree[which]; }
void *refh_msa(void *vh) { return &((RefHandle *)vh)->msa; }
long refh_tree_bytes(void *vh) { return ((RefHandle *)vh)->msa.tree_bytes; }

/* the reference's own tree text (lvb_treeprint -> ur_print, TreeOperations.c:1149-1220) and row titles */
int refh_treeprint(void *vh, int which, const char *path)
{
    RefHandle *h = (RefHandle *)vh;
    FILE *f = fopen(path, "w");
    if (!f)
        return -1;
    lvb_treeprint(&h->msa, f, h->tree[which], h->root[which]);
    fclose(f);
    return 0;
}

void refh_row_title(void *vh, long i, char *out, long cap)
{
    RefHandle *h = (RefHandle *)vh;
    snprintf(out, (size_t)cap, "%s", h->msa.rowtitle[i]);
}

/* struct layout facts the drop-in adapter relies on */
void refh_layout(long *out)
{
    out[0] = sizeof(TREESTACK_TREE_NODES);
    out[1] = offsetof(TREESTACK_TREE_NODES, parent);
    out[2] = offsetof(TREESTACK_TREE_NODES, left);
    out[3] = offsetof(TREESTACK_TREE_NODES, right);
    out[4] = offsetof(TREESTACK_TREE_NODES, changes);
    out[5] = offsetof(TREESTACK_TREE_NODES, sitestate);
    out[6] = sizeof(DataStructure);
    out[7] = offsetof(DataStructure, n_threads_getplen);
    out[8] = offsetof(DataStructure, n_slice_size_getplen);
    out[9] = offsetof(DataStructure, n);
    out[10] = offsetof(DataStructure, numberofpossiblebranches);
    out[11] = offsetof(DataStructure, nwords);
    out[12] = sizeof(Parameters);
    out[13] = offsetof(DataStructure, bytes);
    out[14] = offsetof(DataStructure, m);
}

/* time `reps` incremental proposals (mutate + getplen) of one kind on the reference's own
 * CPU path; every `accept_every`-th proposal is accepted (SwapTrees).  Returns seconds spent
 * in getplen only via *t_getplen and in mutate (incl. treecopy) via *t_mutate; the sum of the
 * proposed lengths via *checksum; sum of dirty counts via *dirty_sum. */
void refh_time_proposals(void *vh, int kind, long reps, long accept_every, double *t_getplen,
                         double *t_mutate, long *checksum, long *dirty_sum)
{
    RefHandle *h = (RefHandle *)vh;
    double tg = 0.0, tm = 0.0;
    long cs = 0, ds = 0;
    struct timespec a, b, c;
    for (long r = 0; r < reps; r++)
    {
        clock_gettime(CLOCK_MONOTONIC, &a);
        refh_mutate(vh, kind);
        clock_gettime(CLOCK_MONOTONIC, &b);
        for (long i = h->msa.n; i < h->msa.numberofpossiblebranches; i++)
            ds += (h->tree[1][i].sitestate[0] == 0U);
        struct timespec b2;
        clock_gettime(CLOCK_MONOTONIC, &b2);
        cs += refh_getplen(vh, 1);
        clock_gettime(CLOCK_MONOTONIC, &c);
        tm += (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
        tg += (c.tv_sec - b2.tv_sec) + 1e-9 * (c.tv_nsec - b2.tv_nsec);
        if (accept_every > 0 && (r % accept_every) == accept_every - 1)
            refh_swap(vh);
    }
    *t_getplen = tg;
    *t_mutate = tm;
    *checksum = cs;
    *dirty_sum = ds;
}

/* time `reps` full evaluations (all internal nodes dirty) of the current tree */
double refh_time_full(void *vh, long reps, long *checksum)
{
    RefHandle *h = (RefHandle *)vh;
    struct timespec a, b;
    long cs = 0;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (long r = 0; r < reps; r++)
    {
        for (long i = h->msa.n; i < h->msa.numberofpossiblebranches; i++)
            h->tree[0][i].sitestate[0] = 0U;
        cs += refh_getplen(vh, 0);
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    *checksum = cs;
    return (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
}

} /* extern "C" */
