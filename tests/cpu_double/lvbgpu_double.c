/* tests/cpu_double/lvbgpu_double.c - TEST DOUBLE, never part of the product.
 *
 * A CPU stand-in for the handful of lvbgpu_* entry points the search host (liblvbhost) calls, so
 * that HOST LOGIC - above all the reference-trajectory search, lvb_amd/csrc/refsearch.cpp - can be
 * checked against the reference program in the `-m "not gpu"` tier, where there is no device.  It
 * scores with the oracle (oracle/fitch_oracle.c), which is allowed here and only here: the file
 * lives under tests/, is compiled by tests/cpu_double/build.py into a temporary library together
 * with the host sources, and nothing under lvb_amd/ or include/ knows it exists.  The product's
 * liblvbhost.so links liblvbgpu.so (HIP) and has no such path.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lvbgpu.h"
#include "../../oracle/fitch_oracle.h"

struct lvbgpu_ctx
{
    long n, nb, nwords;
    uint64_t *enc;
    lvbo_node *cur, *cand;
    long root;
    long *todo;
    int64_t cur_len;
    int have_tree;
    struct dbl_multi *g; /* its multi-chain state (below), made when first needed */
};

lvbgpu_ctx *lvbgpu_double_new(long n, long nwords, const uint64_t *enc)
{
    lvbgpu_ctx *c = (lvbgpu_ctx *)calloc(1, sizeof(*c));
    c->n = n;
    c->nb = 2 * n - 3;
    c->nwords = nwords;
    c->enc = (uint64_t *)malloc((size_t)(n * nwords) * 8);
    memcpy(c->enc, enc, (size_t)(n * nwords) * 8);
    c->cur = lvbo_treealloc(c->nb, nwords);
    c->cand = lvbo_treealloc(c->nb, nwords);
    c->todo = (long *)malloc((size_t)(c->nb + 1) * sizeof(long));
    return c;
}

static void chains_follow_selected(lvbgpu_ctx *c); /* the selected chain's tree changed through a single-tree call */
static void chains_release(lvbgpu_ctx *c);

void lvbgpu_double_free(lvbgpu_ctx *c)
{
    if (!c)
        return;
    chains_release(c); /* (frees the multi-chain state too) */
    free(c->enc);
    free(c->cur);
    free(c->cand);
    free(c->todo);
    free(c);
}

long lvbgpu_n(const lvbgpu_ctx *c) { return c->n; }

static void set_children(lvbo_node *t, long node, long l, long r)
{
    t[node].left = l;
    t[node].right = r;
    if (l >= 0)
        t[l].parent = node;
    if (r >= 0)
        t[r].parent = node;
}

int lvbgpu_set_tree(lvbgpu_ctx *c, const int32_t *left, const int32_t *right, int32_t root, int64_t *length_out)
{
    for (long i = 0; i < c->nb; i++)
        c->cur[i].parent = c->cur[i].left = c->cur[i].right = LVBO_UNSET;
    for (long i = 0; i < c->nb; i++)
        set_children(c->cur, i, left[i], right[i]);
    lvbo_ss_init(c->cur, c->n, c->nb, c->nwords, c->enc);
    c->root = root;
    c->cur_len = lvbo_getplen(c->cur, c->n, c->nb, c->nwords, c->root, c->todo);
    c->have_tree = 1;
    chains_follow_selected(c);
    if (length_out)
        *length_out = c->cur_len;
    return LVBGPU_OK;
}

int lvbgpu_current_length(lvbgpu_ctx *c, int64_t *length_out)
{
    if (!c->have_tree)
        return LVBGPU_E_STATE;
    *length_out = c->cur_len;
    return LVBGPU_OK;
}

/* edits + dirty marks on `t` (a copy of, or, the current tree); a root change re-evaluates everything */
static int64_t apply_and_score(lvbgpu_ctx *c, lvbo_node *t, int32_t ne, const lvbgpu_edit *e, long *root)
{
    long new_root = *root;
    for (int32_t k = 0; k < ne; k++)
        set_children(t, e[k].node, e[k].left, e[k].right);
    if (new_root != c->root)
    {
        t[new_root].parent = LVBO_UNSET;
        for (long i = c->n; i < c->nb; i++)
            lvbo_mark_dirty(t, i);
    }
    else
        for (int32_t k = 0; k < ne; k++)
            if (e[k].node >= c->n)
                lvbo_make_dirty_below(t, e[k].node);
    return lvbo_getplen(t, c->n, c->nb, c->nwords, new_root, c->todo);
}

int lvbgpu_score_batch(lvbgpu_ctx *c, int32_t B, const int32_t *off, const lvbgpu_edit *edits, const int32_t *roots,
                       int64_t *lengths_out)
{
    if (!c->have_tree)
        return LVBGPU_E_STATE;
    for (int32_t b = 0; b < B; b++)
    {
        long root = (roots && roots[b] >= 0) ? roots[b] : c->root;
        lvbo_treecopy(c->cand, c->cur, c->nb, c->nwords);
        lengths_out[b] = apply_and_score(c, c->cand, off[b + 1] - off[b], edits + off[b], &root);
    }
    return LVBGPU_OK;
}

int lvbgpu_commit(lvbgpu_ctx *c, int32_t n_edits, const lvbgpu_edit *edits, int32_t root, int64_t *length_out)
{
    if (!c->have_tree)
        return LVBGPU_E_STATE;
    long new_root = root >= 0 ? root : c->root;
    c->cur_len = apply_and_score(c, c->cur, n_edits, edits, &new_root);
    c->root = new_root;
    chains_follow_selected(c);
    if (length_out)
        *length_out = c->cur_len;
    return LVBGPU_OK;
}

/* the device-proposal and collective entry points have no CPU meaning */
int lvbgpu_proposal_edits(lvbgpu_ctx *c, int32_t b, lvbgpu_edit *e, int32_t cap, int32_t *n, int32_t *k)
{
    (void)c, (void)b, (void)e, (void)cap, (void)n, (void)k;
    return LVBGPU_E_NODEVICE;
}
int lvbgpu_propose_score(lvbgpu_ctx *c, int32_t B, int32_t kind, uint64_t seed, int64_t *l)
{
    (void)c, (void)B, (void)kind, (void)seed, (void)l;
    return LVBGPU_E_NODEVICE;
}
int lvbgpu_propose_score_mixed(lvbgpu_ctx *c, int32_t B, double a, double b, int64_t p, uint64_t seed, int64_t *l)
{
    (void)c, (void)B, (void)a, (void)b, (void)p, (void)seed, (void)l;
    return LVBGPU_E_NODEVICE;
}
int lvbgpu_score_moves(lvbgpu_ctx *c, int32_t B, const lvbgpu_move *m, int64_t *l)
{
    (void)c, (void)B, (void)m, (void)l;
    return LVBGPU_E_NODEVICE;
}
/* The min-reduce over ranks, for multi-process tests of the lockstep searches: no RCCL on the CPU, so the ranks meet
 * in a directory (LVBGPU_DOUBLE_COMM_DIR, with LVBGPU_DOUBLE_RANK / LVBGPU_DOUBLE_WORLD): call number k of rank r
 * publishes its value as file "<k>_<r>" (written under another name, then renamed) and waits for the other ranks'
 * files of the same call.  Ranks that make the same sequence of calls pair up; one that makes fewer leaves the others
 * waiting until the timeout below - which is the failure such a test is after.  Without the variables: one rank. */
#include <stdio.h>
#include <time.h>
#include <unistd.h>
int lvbgpu_allreduce_min(lvbgpu_ctx *c, int64_t *v, int32_t *r)
{
    (void)c;
    static long calls = 0;
    const char *dir = getenv("LVBGPU_DOUBLE_COMM_DIR");
    if (!dir)
    {
        if (r)
            *r = 0;
        return LVBGPU_OK;
    }
    const int rank = atoi(getenv("LVBGPU_DOUBLE_RANK")), world = atoi(getenv("LVBGPU_DOUBLE_WORLD"));
    const long k = calls++;
    char tmp[512], path[512];
    snprintf(tmp, sizeof tmp, "%s/tmp_%ld_%d", dir, k, rank);
    snprintf(path, sizeof path, "%s/%ld_%d", dir, k, rank);
    FILE *f = fopen(tmp, "w");
    if (!f)
        return LVBGPU_E_COMM;
    fprintf(f, "%lld\n", (long long)*v);
    fclose(f);
    if (rename(tmp, path) != 0)
        return LVBGPU_E_COMM;
    long long best = 0;
    int who = -1;
    const time_t deadline = time(NULL) + 60;
    for (int j = 0; j < world; j++)
    {
        snprintf(path, sizeof path, "%s/%ld_%d", dir, k, j);
        long long x;
        for (;;)
        {
            f = fopen(path, "r");
            if (f)
            {
                const int got = fscanf(f, "%lld", &x);
                fclose(f);
                if (got == 1)
                    break;
            }
            if (time(NULL) > deadline)
                return LVBGPU_E_COMM; /* a rank never made call k */
            usleep(200);
        }
        if (who < 0 || x < best)
        {
            best = x;
            who = j;
        }
    }
    *v = best;
    if (r)
        *r = who;
    return LVBGPU_OK;
}
/* ---- several chains per context -------------------------------------------------------------------------------
 * The double keeps R trees and draws the neighbours ITSELF, with the host library's own move generators on a mirror
 * topology per chain (lvbhost_propose after lvbhost_tree_reseed(hash(seed, j)): candidate j of a draw is a function
 * of (seed, j) and of the chain's tree, as the C-ABI promises - not the moves the device would draw, which the host
 * loop must not depend on anyway).  Enough for lvbhost_anneal_chains to run on the CPU: its state machines, the
 * "a chain's trajectory does not depend on R" property, and the sanitizers. */
#include "../../include/lvbhost.h"

#define DBL_MAX_CHAINS 64

typedef struct
{
    lvbo_node *cur;
    long root;
    int64_t cur_len;
    int have_tree;
    lvbhost_tree *mirror;
    uint64_t version;
} dbl_chain;

typedef struct
{
    int32_t k;                                    /* segments */
    int32_t chain[DBL_MAX_CHAINS], start[DBL_MAX_CHAINS], count[DBL_MAX_CHAINS];
    uint64_t version[DBL_MAX_CHAINS];
    int32_t total;
    int32_t *off;                                 /* [total + 1] into edits */
    lvbgpu_edit *edits;
    size_t edits_cap, off_cap;
    int64_t *len;
    int in_flight;
} dbl_slot;

/* Every context has its own multi-chain state (lvbhost_anneal_chains serves several contexts - lanes - from one thread);
 * G is the state of the context the current call is about (enter(), first thing in every function that touches it). */
#include "../../lvb_amd/csrc/decide.h"
struct dbl_multi
{
    struct
    {
        int32_t k;
        DecideRule rule[DBL_MAX_CHAINS];
        int32_t map[DBL_MAX_CHAINS];
        int active;
    } step;                                       /* the step in flight (lvbgpu_chains_step_*) */
    lvbgpu_ctx *owner;                            /* the context itself once it has chains, else NULL */
    int32_t R, sel;
    dbl_chain ch[DBL_MAX_CHAINS];
    dbl_slot slot[2];
    int last_slot;
    int32_t npicked;
    int32_t picked_off[DBL_MAX_CHAINS + 1];
    lvbgpu_edit *picked;
    size_t picked_cap;
    uint64_t versions;
};
static __thread struct dbl_multi *Gcur;
#define G (*Gcur)
static void enter(const lvbgpu_ctx *cc)
{
    lvbgpu_ctx *c = (lvbgpu_ctx *)cc;
    if (!c->g)
        c->g = (struct dbl_multi *)calloc(1, sizeof(struct dbl_multi));
    Gcur = c->g;
}


static void store_selected(lvbgpu_ctx *c)
{
    enter(c);
    if (G.owner != c)
        return;
    dbl_chain *h = &G.ch[G.sel];
    h->cur = c->cur;
    h->root = c->root;
    h->cur_len = c->cur_len;
    h->have_tree = c->have_tree;
}

static void load_selected(lvbgpu_ctx *c)
{
    enter(c);
    dbl_chain *h = &G.ch[G.sel];
    c->cur = h->cur;
    c->root = h->root;
    c->cur_len = h->cur_len;
    c->have_tree = h->have_tree;
}

static void mirror_of_selected(lvbgpu_ctx *c)
{
    enter(c);
    /* (re)build the selected chain's mirror topology from its node records */
    dbl_chain *h = &G.ch[G.sel];
    int32_t *l = (int32_t *)malloc((size_t)c->nb * 4), *r = (int32_t *)malloc((size_t)c->nb * 4);
    for (long i = 0; i < c->nb; i++)
    {
        l[i] = c->cur[i].left >= 0 ? (int32_t)c->cur[i].left : -1;
        r[i] = c->cur[i].right >= 0 ? (int32_t)c->cur[i].right : -1;
    }
    if (h->mirror)
        lvbhost_tree_free(h->mirror);
    h->mirror = lvbhost_tree_from_arrays((int32_t)c->n, l, r, (int32_t)c->root, 1);
    h->version = ++G.versions;
    free(l);
    free(r);
}

static void chains_follow_selected(lvbgpu_ctx *c)
{
    enter(c);
    if (G.owner != c)
        return;
    store_selected(c);
    mirror_of_selected(c);
}

void lvbgpu_double_chains_reset(void);
static void chains_release(lvbgpu_ctx *c)
{
    enter(c);
    if (G.owner == c)
        lvbgpu_double_chains_reset();
    free(c->g);
    c->g = NULL;
    Gcur = NULL;
}

void lvbgpu_double_chains_reset(void)
{
    if (!Gcur || !G.owner)
        return;
    lvbgpu_ctx *c = G.owner;
    store_selected(c);
    for (int32_t i = 0; i < G.R; i++)
    {
        if (i != 0)
            free(G.ch[i].cur);
        if (G.ch[i].mirror)
            lvbhost_tree_free(G.ch[i].mirror);
    }
    /* chain 0's tree block is the context's own again */
    c->cur = G.ch[0].cur;
    c->root = G.ch[0].root;
    c->cur_len = G.ch[0].cur_len;
    c->have_tree = G.ch[0].have_tree;
    for (int s = 0; s < 2; s++)
    {
        free(G.slot[s].off);
        free(G.slot[s].edits);
        free(G.slot[s].len);
    }
    free(G.picked);
    memset(Gcur, 0, sizeof(struct dbl_multi));
}

int lvbgpu_set_chains(lvbgpu_ctx *c, int32_t r)
{
    enter(c);
    if (!c || r < 1 || r > DBL_MAX_CHAINS)
        return (getenv("DBL_DEBUG") ? fprintf(stderr, "double: E_ARG at line %d\n", __LINE__) : 0, LVBGPU_E_ARG);
    if (G.owner)
        lvbgpu_double_chains_reset();
    G.owner = c;
    G.R = r;
    G.sel = 0;
    G.ch[0].cur = c->cur;
    G.ch[0].root = c->root;
    G.ch[0].cur_len = c->cur_len;
    G.ch[0].have_tree = c->have_tree;
    if (c->have_tree)
        mirror_of_selected(c);
    for (int32_t i = 1; i < r; i++)
        G.ch[i].cur = lvbo_treealloc(c->nb, c->nwords);
    return LVBGPU_OK;
}

int lvbgpu_select_chain(lvbgpu_ctx *c, int32_t k)
{
    enter(c);
    if (G.owner != c)
        return k == 0 ? LVBGPU_OK : LVBGPU_E_ARG;
    if (k < 0 || k >= G.R)
        return (getenv("DBL_DEBUG") ? fprintf(stderr, "double: E_ARG at line %d\n", __LINE__) : 0, LVBGPU_E_ARG);
    store_selected(c);
    G.sel = k;
    load_selected(c);
    return LVBGPU_OK;
}

int32_t lvbgpu_chains(const lvbgpu_ctx *c)
{
    enter(c);
    return G.owner == c ? G.R : 1;
}

/* a context that never asked for chains has one (lvbgpu_chains() == 1): the chain calls work on it all the same */
static void adopt(lvbgpu_ctx *c)
{
    enter(c);
    if (c && G.owner != c)
        lvbgpu_set_chains(c, 1);
}

static uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

int lvbgpu_chains_submit(lvbgpu_ctx *c, int32_t s, int32_t k, const lvbgpu_chain_draw *d)
{
    enter(c);
    adopt(c);
    if (G.owner != c || s < 0 || s > 1 || k < 1 || k > G.R || !d)
        return (getenv("DBL_DEBUG") ? fprintf(stderr, "double: E_ARG at line %d\n", __LINE__) : 0, LVBGPU_E_ARG);
    store_selected(c);
    dbl_slot *sl = &G.slot[s];
    if (sl->in_flight)
        return LVBGPU_E_STATE;
    int32_t total = 0;
    for (int32_t i = 0; i < k; i++)
    {
        if (d[i].chain < 0 || d[i].chain >= G.R || d[i].count < 1 || !G.ch[d[i].chain].have_tree)
            return (getenv("DBL_DEBUG") ? fprintf(stderr, "double: E_ARG at line %d\n", __LINE__) : 0, LVBGPU_E_ARG);
        for (int32_t j = 0; j < i; j++)
            if (d[j].chain == d[i].chain)
                return (getenv("DBL_DEBUG") ? fprintf(stderr, "double: E_ARG at line %d\n", __LINE__) : 0, LVBGPU_E_ARG);
        total += d[i].count;
    }
    if ((size_t)total + 1 > sl->off_cap)
    {
        sl->off_cap = (size_t)total + 1;
        sl->off = (int32_t *)realloc(sl->off, sl->off_cap * 4);
        sl->len = (int64_t *)realloc(sl->len, sl->off_cap * 8);
    }
    sl->k = k;
    sl->total = total;
    const int32_t cap = (int32_t)(2 * c->nb + 8);
    lvbgpu_edit *tmp = (lvbgpu_edit *)malloc((size_t)cap * sizeof(lvbgpu_edit));
    int32_t at = 0;
    size_t ne_total = 0;
    sl->off[0] = 0;
    for (int32_t i = 0; i < k; i++)
    {
        dbl_chain *h = &G.ch[d[i].chain];
        sl->chain[i] = d[i].chain;
        sl->start[i] = at;
        sl->count[i] = d[i].count;
        sl->version[i] = h->version;
        for (int32_t j = 0; j < d[i].count; j++, at++)
        {
            const uint64_t r = mix64(d[i].seed ^ mix64((uint64_t)j + 1));
            int kind = d[i].kind;
            if (kind == -1)
                kind = j % 3;
            else if (kind == -2)
                kind = ((d[i].mix_a + (uint32_t)j) & 1u) ? 1 : 0;
            else if (kind == -3)
            {
                const uint32_t u = (uint32_t)(r >> 32);
                kind = u < d[i].mix_a ? 0 : (u < d[i].mix_b ? 1 : 2);
            }
            lvbhost_tree_reseed(h->mirror, r | 1u);
            const int ne = lvbhost_propose(h->mirror, kind, tmp, cap);
            if (ne < 0)
            {
                free(tmp);
                return ne;
            }
            if (ne_total + (size_t)ne > sl->edits_cap)
            {
                sl->edits_cap = 2 * (ne_total + (size_t)ne) + 64;
                sl->edits = (lvbgpu_edit *)realloc(sl->edits, sl->edits_cap * sizeof(lvbgpu_edit));
            }
            memcpy(sl->edits + ne_total, tmp, (size_t)ne * sizeof(lvbgpu_edit));
            ne_total += (size_t)ne;
            sl->off[at + 1] = (int32_t)ne_total;
            /* score it against the chain's tree */
            lvbo_treecopy(c->cand, h->cur, c->nb, c->nwords);
            long root = h->root;
            const long keep_root = c->root;
            c->root = h->root; /* apply_and_score compares against the context's root */
            sl->len[at] = apply_and_score(c, c->cand, ne, tmp, &root);
            c->root = keep_root;
        }
    }
    free(tmp);
    sl->in_flight = 1;
    return LVBGPU_OK;
}

int lvbgpu_chains_collect(lvbgpu_ctx *c, int32_t s, int64_t *l)
{
    enter(c);
    if (G.owner != c || s < 0 || s > 1 || !l)
        return (getenv("DBL_DEBUG") ? fprintf(stderr, "double: E_ARG at line %d\n", __LINE__) : 0, LVBGPU_E_ARG);
    dbl_slot *sl = &G.slot[s];
    if (!sl->in_flight)
        return LVBGPU_E_STATE;
    memcpy(l, sl->len, (size_t)sl->total * 8);
    sl->in_flight = 0;
    G.last_slot = s;
    return LVBGPU_OK;
}

int lvbgpu_chains_propose_score(lvbgpu_ctx *c, int32_t k, const lvbgpu_chain_draw *d, int64_t *l)
{
    enter(c);
    const int rc = lvbgpu_chains_submit(c, 0, k, d);
    return rc != LVBGPU_OK ? rc : lvbgpu_chains_collect(c, 0, l);
}

/* edits (+ a new root) on chain h's tree and on its mirror */
static int commit_on_chain(lvbgpu_ctx *c, dbl_chain *h, int32_t ne, const lvbgpu_edit *e, long new_root, int64_t expect)
{
    enter(c);
    const long keep_root = c->root;
    c->root = h->root;
    long root = new_root >= 0 ? new_root : h->root;
    h->cur_len = apply_and_score(c, h->cur, ne, e, &root);
    c->root = keep_root;
    h->root = root;
    h->version = ++G.versions;
    if (expect >= 0 && h->cur_len != expect)
        return LVBGPU_E_STATE; /* a candidate's committed length is its scored length */
    return lvbhost_tree_apply(h->mirror, e, ne, (int32_t)new_root);
}

int lvbgpu_chains_commit(lvbgpu_ctx *c, int32_t k, const lvbgpu_chain_pick *p)
{
    enter(c);
    adopt(c);
    if (G.owner != c || k < 1 || k > G.R || !p)
        return (getenv("DBL_DEBUG") ? fprintf(stderr, "double: E_ARG at line %d\n", __LINE__) : 0, LVBGPU_E_ARG);
    store_selected(c);
    dbl_slot *sl = &G.slot[G.last_slot];
    G.npicked = 0;
    G.picked_off[0] = 0;
    for (int32_t j = 0; j < k; j++)
    {
        int32_t seg = -1;
        for (int32_t i = 0; i < sl->k; i++)
            if (sl->chain[i] == p[j].chain)
                seg = i;
        if (seg < 0 || p[j].b < 0 || p[j].b >= sl->count[seg])
            return (getenv("DBL_DEBUG") ? fprintf(stderr, "double: E_ARG at line %d\n", __LINE__) : 0, LVBGPU_E_ARG);
        dbl_chain *h = &G.ch[p[j].chain];
        if (h->version != sl->version[seg])
            return LVBGPU_E_STATE; /* the chain's tree changed since these candidates were drawn */
        const int32_t at = sl->start[seg] + p[j].b, ne = sl->off[at + 1] - sl->off[at];
        const size_t need = (size_t)G.picked_off[j] + (size_t)ne;
        if (need > G.picked_cap)
        {
            G.picked_cap = 2 * need + 64;
            G.picked = (lvbgpu_edit *)realloc(G.picked, G.picked_cap * sizeof(lvbgpu_edit));
        }
        memcpy(G.picked + G.picked_off[j], sl->edits + sl->off[at], (size_t)ne * sizeof(lvbgpu_edit));
        G.picked_off[j + 1] = G.picked_off[j] + ne;
        G.npicked = j + 1;
        const int rc = commit_on_chain(c, h, ne, sl->edits + sl->off[at], -1, sl->len[at]);
        if (rc != LVBGPU_OK)
            return rc;
    }
    load_selected(c);
    return LVBGPU_OK;
}

int lvbgpu_chains_picked_edits(lvbgpu_ctx *c, int32_t j, lvbgpu_edit *e, int32_t cap, int32_t *n)
{
    enter(c);
    if (G.owner != c || j < 0 || j >= G.npicked || !e || !n)
        return (getenv("DBL_DEBUG") ? fprintf(stderr, "double: E_ARG at line %d\n", __LINE__) : 0, LVBGPU_E_ARG);
    const int32_t ne = G.picked_off[j + 1] - G.picked_off[j];
    if (ne > cap)
        return (getenv("DBL_DEBUG") ? fprintf(stderr, "double: E_ARG at line %d\n", __LINE__) : 0, LVBGPU_E_ARG);
    memcpy(e, G.picked + G.picked_off[j], (size_t)ne * sizeof(lvbgpu_edit));
    *n = ne;
    return LVBGPU_OK;
}

/* a whole step (include/lvbgpu.h lvbgpu_chains_step_*): the rule rides with the batch; the double decides at the
 * collect with the product's own rule function (lvb_amd/csrc/decide.h: plain C) and commits the picks */
#define STEP (G.step) /* (per context, as everything about its chains) */

int lvbgpu_chains_step_submit(lvbgpu_ctx *c, int32_t s, int32_t k, const lvbgpu_chain_draw *d, const lvbgpu_chain_rule *r)
{
    enter(c);
    if (!r)
        return LVBGPU_E_ARG;
    const int rc = lvbgpu_chains_submit(c, s, k, d);
    if (rc != LVBGPU_OK)
        return rc;
    uint32_t at = 0;
    STEP.k = k;
    for (int32_t i = 0; i < k; i++)
    {
        STEP.rule[i].cur = r[i].cur_length;
        STEP.rule[i].t = r[i].temperature;
        STEP.rule[i].minlen = r[i].min_len_tree;
        STEP.rule[i].seed = r[i].accept_seed;
        STEP.rule[i].start = at;
        STEP.rule[i].count = (uint32_t)d[i].count;
        at += (uint32_t)d[i].count;
    }
    STEP.active = 1;
    return LVBGPU_OK;
}

int lvbgpu_chains_step_collect(lvbgpu_ctx *c, int32_t s, int64_t *l, int32_t *picks)
{
    enter(c);
    if (!STEP.active || !picks)
        return LVBGPU_E_STATE;
    STEP.active = 0;
    int rc = lvbgpu_chains_collect(c, s, l);
    if (rc != LVBGPU_OK)
        return rc;
    lvbgpu_chain_pick pk[DBL_MAX_CHAINS];
    int32_t np = 0;
    const dbl_slot *sl = &G.slot[s];
    for (int32_t i = 0; i < STEP.k; i++)
    {
        picks[i] = -1;
        STEP.map[i] = -1;
        for (uint32_t j = 0; j < STEP.rule[i].count; j++)
            if (lvb_take(l[STEP.rule[i].start + j] == INT64_MAX ? LVB_OVERFLOW_LENGTH : (long long)l[STEP.rule[i].start + j], &STEP.rule[i], j))
            {
                picks[i] = (int32_t)j;
                STEP.map[i] = np;
                pk[np].chain = sl->chain[i];
                pk[np].b = (int32_t)j;
                np++;
                break;
            }
    }
    return np ? lvbgpu_chains_commit(c, np, pk) : LVBGPU_OK;
}

int lvbgpu_chains_step_edits(lvbgpu_ctx *c, int32_t i, lvbgpu_edit *e, int32_t cap, int32_t *n)
{
    enter(c);
    if (i < 0 || i >= DBL_MAX_CHAINS || STEP.map[i] < 0)
        return (getenv("DBL_DEBUG") ? fprintf(stderr, "double: E_ARG at line %d (i %d k %d)\n", __LINE__, i, STEP.k) : 0, LVBGPU_E_ARG);
    return lvbgpu_chains_picked_edits(c, STEP.map[i], e, cap, n);
}

/* on threads, as the library does it (a few short-lived ones here): what the host hands to lvbgpu_parallel_for must be
 * free of races between the tasks - the ThreadSanitizer tier runs the annealing loop against this */
#include <pthread.h>
typedef struct
{
    lvbgpu_task_fn fn;
    void *arg;
    int32_t first, step, n;
} dbl_par;
static void *dbl_par_run(void *p)
{
    dbl_par *j = (dbl_par *)p;
    for (int32_t i = j->first; i < j->n; i += j->step)
        j->fn(i, j->arg);
    return NULL;
}
int lvbgpu_parallel_for(lvbgpu_ctx *c, int32_t n, lvbgpu_task_fn fn, void *arg)
{
    enter(c);
    if (!c || n < 0 || !fn)
        return LVBGPU_E_ARG;
    enum { T = 3 };
    if (n < 2)
    {
        for (int32_t i = 0; i < n; i++)
            fn(i, arg);
        return LVBGPU_OK;
    }
    pthread_t th[T];
    dbl_par job[T];
    int started[T] = {0};
    for (int t = 0; t < T; t++)
    {
        job[t].fn = fn;
        job[t].arg = arg;
        job[t].first = t;
        job[t].step = T;
        job[t].n = n;
        if (t > 0)
            started[t] = pthread_create(&th[t], NULL, dbl_par_run, &job[t]) == 0;
    }
    dbl_par_run(&job[0]);
    for (int t = 1; t < T; t++)
        if (started[t])
            pthread_join(th[t], NULL);
        else
            dbl_par_run(&job[t]);
    return LVBGPU_OK;
}

/* host-made candidates of several chains: scored against each candidate's own chain, one accepted per listed chain */
int lvbgpu_chains_score_edits(lvbgpu_ctx *c, int32_t B, const int32_t *chain_of, const int32_t *off, const lvbgpu_edit *edits,
                              int64_t *lengths_out)
{
    enter(c);
    adopt(c);
    if (G.owner != c || B < 1 || !chain_of || !off || !lengths_out)
        return LVBGPU_E_ARG;
    store_selected(c);
    const long keep_root = c->root;
    int rc = LVBGPU_OK;
    for (int32_t b = 0; b < B && rc == LVBGPU_OK; b++)
    {
        if (chain_of[b] < 0 || chain_of[b] >= G.R || !G.ch[chain_of[b]].have_tree)
        {
            rc = chain_of[b] < 0 || chain_of[b] >= G.R ? LVBGPU_E_ARG : LVBGPU_E_STATE;
            break;
        }
        dbl_chain *h = &G.ch[chain_of[b]];
        long root = h->root;
        c->root = h->root;
        lvbo_treecopy(c->cand, h->cur, c->nb, c->nwords);
        lengths_out[b] = apply_and_score(c, c->cand, off[b + 1] - off[b], edits + off[b], &root);
    }
    c->root = keep_root;
    load_selected(c);
    return rc;
}

int lvbgpu_chains_commit_edits(lvbgpu_ctx *c, int32_t k, const int32_t *chains, const int32_t *off, const lvbgpu_edit *edits)
{
    enter(c);
    adopt(c);
    if (G.owner != c || k < 1 || k > G.R || !chains || !off || !edits)
        return LVBGPU_E_ARG;
    store_selected(c);
    int rc = LVBGPU_OK;
    for (int32_t j = 0; j < k && rc == LVBGPU_OK; j++)
    {
        if (chains[j] < 0 || chains[j] >= G.R || off[j + 1] <= off[j])
        {
            rc = LVBGPU_E_ARG;
            break;
        }
        rc = commit_on_chain(c, &G.ch[chains[j]], off[j + 1] - off[j], edits + off[j], -1, -1);
    }
    load_selected(c);
    return rc;
}

int lvbgpu_chains_reroot(lvbgpu_ctx *c, int32_t k, const lvbgpu_chain_root *r)
{
    enter(c);
    adopt(c);
    if (G.owner != c || k < 1 || k > G.R || !r)
        return (getenv("DBL_DEBUG") ? fprintf(stderr, "double: E_ARG at line %d\n", __LINE__) : 0, LVBGPU_E_ARG);
    store_selected(c);
    const int32_t cap = (int32_t)(2 * c->nb + 8);
    lvbgpu_edit *tmp = (lvbgpu_edit *)malloc((size_t)cap * sizeof(lvbgpu_edit));
    int rc = LVBGPU_OK;
    for (int32_t j = 0; j < k && rc == LVBGPU_OK; j++)
    {
        if (r[j].chain < 0 || r[j].chain >= G.R || r[j].new_root < 0 || r[j].new_root >= c->n)
        {
            rc = LVBGPU_E_ARG;
            break;
        }
        dbl_chain *h = &G.ch[r[j].chain];
        const int64_t before = h->cur_len;
        const int ne = lvbhost_reroot_edits(h->mirror, r[j].new_root, tmp, cap);
        rc = ne < 0 ? ne : commit_on_chain(c, h, ne, tmp, r[j].new_root, before); /* a re-root keeps the length */
    }
    free(tmp);
    load_selected(c);
    return rc;
}

/* a second context on the same alignment (lvbhost_anneal_chains' lanes) */
int lvbgpu_fork(lvbgpu_ctx *src, lvbgpu_ctx **out)
{
    if (!src || !out)
        return LVBGPU_E_ARG;
    *out = lvbgpu_double_new(src->n, src->nwords, src->enc);
    return *out ? LVBGPU_OK : LVBGPU_E_NOMEM;
}

void lvbgpu_destroy(lvbgpu_ctx *c) { lvbgpu_double_free(c); }

/* everything is computed at the submit: a submitted batch is always ready */
int lvbgpu_chains_ready(lvbgpu_ctx *c, int32_t s, int32_t *ready)
{
    enter(c);
    if (G.owner != c || s < 0 || s > 1 || !ready)
        return LVBGPU_E_ARG;
    *ready = 1;
    return LVBGPU_OK;
}
