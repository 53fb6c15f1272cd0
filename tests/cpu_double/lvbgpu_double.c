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

void lvbgpu_double_free(lvbgpu_ctx *c)
{
    if (!c)
        return;
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
int lvbgpu_allreduce_min(lvbgpu_ctx *c, int64_t *v, int32_t *r)
{
    (void)c, (void)v, (void)r;
    return LVBGPU_E_NODEVICE;
}
/* several chains per context: device-only as well */
int lvbgpu_set_chains(lvbgpu_ctx *c, int32_t r)
{
    (void)c, (void)r;
    return LVBGPU_E_NODEVICE;
}
int lvbgpu_select_chain(lvbgpu_ctx *c, int32_t k)
{
    (void)c;
    return k == 0 ? LVBGPU_OK : LVBGPU_E_NODEVICE;
}
int32_t lvbgpu_chains(const lvbgpu_ctx *c)
{
    (void)c;
    return 1;
}
int lvbgpu_chains_propose_score(lvbgpu_ctx *c, int32_t k, const lvbgpu_chain_draw *d, int64_t *l)
{
    (void)c, (void)k, (void)d, (void)l;
    return LVBGPU_E_NODEVICE;
}
int lvbgpu_chains_commit(lvbgpu_ctx *c, int32_t k, const lvbgpu_chain_pick *p)
{
    (void)c, (void)k, (void)p;
    return LVBGPU_E_NODEVICE;
}
int lvbgpu_chains_picked_edits(lvbgpu_ctx *c, int32_t j, lvbgpu_edit *e, int32_t cap, int32_t *n)
{
    (void)c, (void)j, (void)e, (void)cap, (void)n;
    return LVBGPU_E_NODEVICE;
}
int lvbgpu_chains_submit(lvbgpu_ctx *c, int32_t s, int32_t k, const lvbgpu_chain_draw *d)
{
    (void)c, (void)s, (void)k, (void)d;
    return LVBGPU_E_NODEVICE;
}
int lvbgpu_chains_collect(lvbgpu_ctx *c, int32_t s, int64_t *l)
{
    (void)c, (void)s, (void)l;
    return LVBGPU_E_NODEVICE;
}
int lvbgpu_chains_reroot(lvbgpu_ctx *c, int32_t k, const lvbgpu_chain_root *r)
{
    (void)c, (void)k, (void)r;
    return LVBGPU_E_NODEVICE;
}
