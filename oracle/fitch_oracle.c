/* oracle/fitch_oracle.c - CPU restatement of LVB's Fitch-scoring hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liblvboracle.so; lvb_amd/ (the HIP path) never does
 * and fails loudly without its HIP library.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_vs_reference.py,
 * tests/test_oracle_golden.py) against
 *   - the real reference compiled into oracle/_ref/liblvbref.so (in containers that hold
 *     /root/reference), on the reference's own test alignments and on random trees/mutations;
 *   - the committed fixtures under tests/golden/ generated from that library by
 *     tests/golden/gen_golden.py (lengths, per-node changes, recomputed state sets).
 *
 * Written from the behaviour of (citations are file:line under the reference's src/):
 *   TreeEvaluation.c:182-265   getplen, serial branch          -> lvbo_getplen
 *   TreeEvaluation.c:64-181    getplen, OpenMP site-slice path -> lvbo_getplen_sliced
 *   TreeEvaluation.c:219-230   SWAR word step                  -> lvbo_combine
 *   LVB.h:72-89                bit codes, 16 sites per 64-bit word, M7/M8 masks
 *   LVB.h:121-128              node record                     -> lvbo_node
 *   DataOperations.c:164-249   DNAToBinary                     -> lvbo_encode_row
 *   DataOperations.c:446-467   words_per_row / bytes_per_row   -> lvbo_words_per_row
 *   DataOperations.c:272-295   constchar                       -> lvbo_variable_columns
 *   DataOperations.c:53-105    getstatev / MinimumTreeLength   -> lvbo_min_tree_length
 *   TreeOperations.c:61-76     tree_bytes                      -> lvbo_tree_bytes
 *   TreeOperations.c:914-955   treealloc                       -> lvbo_treealloc
 *   TreeOperations.c:1500-1513 ss_init                         -> lvbo_ss_init
 *   TreeOperations.c:88-103    make_dirty_below                -> lvbo_make_dirty_below
 *   TreeOperations.c:737-773   treecopy                        -> lvbo_treecopy
 */
#include "fitch_oracle.h"

#include <stdlib.h>
#include <string.h>

#define SITES_PER_WORD 16 /* LVB.h:83 LENGTH_WORD */
static const uint64_t M7 = 0x7777777777777777ULL; /* LVB.h:88 */
static const uint64_t M8 = 0x8888888888888888ULL; /* LVB.h:89 */

/* ---------------------------------------------------------------- encoding */

long lvbo_words_per_row(long m)
{
    /* DataOperations.c:446-458: ceil(m / 16) */
    return (m + SITES_PER_WORD - 1) / SITES_PER_WORD;
}

int lvbo_encode_char(char base)
{
    /* DataOperations.c:190-241; bit0=A bit1=C bit2=G bit3=T (LVB.h:73-76).
     * Returns the 4-bit state set, or -1 for a symbol the reference rejects with crash(). */
    enum { A = 1, C = 2, G = 4, T = 8 };
    switch (base)
    {
    case 'A': return A;
    case 'C': return C;
    case 'G': return G;
    case 'T': case 'U': return T;
    case 'Y': return C | T;
    case 'R': return A | G;
    case 'W': return A | T;
    case 'S': return C | G;
    case 'K': return T | G;
    case 'M': return C | A;
    case 'B': return C | G | T;
    case 'D': return A | G | T;
    case 'H': return A | C | T;
    case 'V': return A | C | G;
    case 'N': case 'X': case '?': case '-': return A | C | G | T;
    default: return -1;
    }
}

int lvbo_encode_row(const char *row, long m, long nwords, uint64_t *out)
{
    /* site 16*j+k occupies nibble k of word j (shift k<<2, DataOperations.c:244);
     * positions >= m are padded with 'N' (187-188) so padding never adds length. */
    for (long j = 0; j < nwords; j++)
    {
        uint64_t w = 0;
        for (long k = 0; k < SITES_PER_WORD; k++)
        {
            long site = j * SITES_PER_WORD + k;
            int s = (site < m) ? lvbo_encode_char(row[site]) : 0xF;
            if (s < 0)
                return -1;
            w |= (uint64_t)s << (4 * k);
        }
        out[j] = w;
    }
    return 0;
}

long lvbo_variable_columns(long n, long m, const char *const *rows, unsigned char *keep)
{
    /* DataOperations.c:272-295: a column is kept iff some row's RAW character differs from
     * row 0's.  Returns the number kept. */
    long kept = 0;
    for (long k = 0; k < m; k++)
    {
        keep[k] = 0;
        for (long i = 1; i < n; i++)
            if (rows[i][k] != rows[0][k])
            {
                keep[k] = 1;
                kept++;
                break;
            }
    }
    return kept;
}

long lvbo_min_tree_length(long n, long m, const char *const *rows)
{
    /* DataOperations.c:53-105: per column, (#distinct characters other than - ? N X) - 1;
     * more than 5 distinct -> 5 (MAXSTATES).  Partial ambiguity codes count as plain states. */
    long total = 0;
    for (long k = 0; k < m; k++)
    {
        char seen[8];
        int nseen = 0, overflow = 0;
        for (long i = 0; i < n && !overflow; i++)
        {
            char c = rows[i][k];
            int known = 0;
            for (int s = 0; s < nseen; s++)
                if (seen[s] == c)
                    known = 1;
            if (known)
                continue;
            if (c != '-' && c != '?' && c != 'N' && c != 'X')
                seen[nseen++] = c;
            if (nseen > 5)
                overflow = 1;
        }
        total += overflow ? 5 : (long)nseen - 1;
    }
    return total;
}

/* ---------------------------------------------------------------- word step */

uint64_t lvbo_combine(uint64_t x, uint64_t y, long *changes)
{
    /* TreeEvaluation.c:219-230.  For each of the 16 nibbles: if x&y is non-empty the result
     * is the intersection, else the union and one change is counted. */
    uint64_t both = x & y;
    uint64_t u = (((both & M7) + M7) | both) & M8; /* bit 3 of nibble set <=> intersection non-empty */
    *changes += SITES_PER_WORD - (long)__builtin_popcountll(u);
    u >>= 3;
    return both | ((x | y) & ((u + M7) ^ M8));
}

/* ---------------------------------------------------------------- tree block */

long lvbo_tree_bytes(long nbranches, long nwords)
{
    return nbranches * (long)sizeof(lvbo_node) + nbranches * nwords * 8;
}

lvbo_node *lvbo_treealloc(long nbranches, long nwords)
{
    /* one block: node records followed by all state sets (TreeOperations.c:914-955);
     * every node starts dirty (word 0 == 0) with UNSET scalars. */
    lvbo_node *t = (lvbo_node *)malloc((size_t)lvbo_tree_bytes(nbranches, nwords));
    if (!t)
        return NULL;
    uint64_t *ss = (uint64_t *)((unsigned char *)t + nbranches * sizeof(lvbo_node));
    for (long i = 0; i < nbranches; i++)
    {
        t[i].parent = t[i].left = t[i].right = t[i].changes = LVBO_UNSET;
        t[i].sitestate = ss + i * nwords;
        t[i].sitestate[0] = 0;
    }
    return t;
}

void lvbo_tree_set_topology(lvbo_node *tree, long nbranches, const long *parent, const long *left,
                            const long *right)
{
    for (long i = 0; i < nbranches; i++)
    {
        tree[i].parent = parent[i];
        tree[i].left = left[i];
        tree[i].right = right[i];
    }
}

void lvbo_ss_init(lvbo_node *tree, long n, long nbranches, long nwords, const uint64_t *enc)
{
    /* TreeOperations.c:1500-1513: leaf i gets taxon i's encoded row; internal nodes dirty */
    for (long i = 0; i < n; i++)
        memcpy(tree[i].sitestate, enc + i * nwords, (size_t)nwords * 8);
    for (long i = n; i < nbranches; i++)
        tree[i].sitestate[0] = 0;
}

void lvbo_mark_dirty(lvbo_node *tree, long node) { tree[node].sitestate[0] = 0; }

void lvbo_make_dirty_below(lvbo_node *tree, long node)
{
    /* TreeOperations.c:88-103: node and its ancestors, stopping before the root record */
    do
    {
        tree[node].sitestate[0] = 0;
        node = tree[node].parent;
    } while (tree[node].parent != LVBO_UNSET);
}

void lvbo_treecopy(lvbo_node *dest, const lvbo_node *src, long nbranches, long nwords)
{
    /* TreeOperations.c:737-773: scalars (dest keeps its own set pointers) + one memcpy of sets */
    for (long i = 0; i < nbranches; i++)
    {
        dest[i].parent = src[i].parent;
        dest[i].left = src[i].left;
        dest[i].right = src[i].right;
        dest[i].changes = src[i].changes;
    }
    memcpy(dest[0].sitestate, src[0].sitestate, (size_t)(nbranches * nwords) * 8);
}

/* ---------------------------------------------------------------- getplen */

static long root_changes(const lvbo_node *tree, long root, long w0, long w1)
{
    /* TreeEvaluation.c:242-264: combine the root's two children (counted), then combine that
     * with the root leaf's own set (counted); nothing is stored. */
    long ch = 0;
    const uint64_t *l = tree[tree[root].left].sitestate;
    const uint64_t *r = tree[tree[root].right].sitestate;
    const uint64_t *s = tree[root].sitestate;
    for (long j = w0; j < w1; j++)
    {
        uint64_t z = lvbo_combine(l[j], r[j], &ch);
        (void)lvbo_combine(z, s[j], &ch);
    }
    return ch;
}

long lvbo_getplen(lvbo_node *tree, long n, long nbranches, long nwords, long root, long *todo)
{
    /* TreeEvaluation.c:182-265 (serial branch).  `todo` is caller scratch of nbranches-n longs. */
    long total = 0, ntodo = 0, done = 0;

    for (long i = n; i < nbranches; i++)
    {
        if (tree[i].sitestate[0] == 0)
        {
            todo[ntodo++] = i;
            tree[i].changes = 0;
        }
        else
            total += tree[i].changes;
    }

    /* index-order list, swept until every dirty node has had both children ready (204-236) */
    while (done < ntodo)
    {
        for (long t = 0; t < ntodo; t++)
        {
            long b = todo[t];
            if (tree[b].sitestate[0] != 0)
                continue;
            const uint64_t *l = tree[tree[b].left].sitestate;
            const uint64_t *r = tree[tree[b].right].sitestate;
            if (l[0] == 0 || r[0] == 0)
                continue;
            long ch = 0;
            uint64_t *z = tree[b].sitestate;
            for (long j = 0; j < nwords; j++)
                z[j] = lvbo_combine(l[j], r[j], &ch);
            tree[b].changes += ch;
            total += ch;
            done++;
        }
    }

    total += root_changes(tree, root, 0, nwords);
    return total; /* the reference asserts total > 0 (267); callers check */
}

long lvbo_getplen_sliced(lvbo_node *tree, long n, long nbranches, long nwords, long root,
                         long *todo, int nslices, long slice_words)
{
    /* TreeEvaluation.c:64-181: the OpenMP branch cuts the word axis into nslices contiguous
     * slices of slice_words words (the last slice also takes the tail, 95-97); each slice walks
     * the whole dirty set on its own words and partial sums are added afterwards (166-179).
     * Restated sequentially, slice by slice: the arithmetic and the results are the same. */
    long total = 0, ntodo = 0;
    for (long i = n; i < nbranches; i++)
    {
        if (tree[i].sitestate[0] == 0)
            todo[ntodo++] = i;
        else
            total += tree[i].changes;
    }
    unsigned char *ready = (unsigned char *)malloc((size_t)(nbranches > 0 ? nbranches : 1));
    long *partial = (long *)calloc((size_t)(ntodo > 0 ? ntodo : 1), sizeof(long));
    long rootsum = 0;

    for (int s = 0; s < nslices; s++)
    {
        long w0 = slice_words * s;
        long w1 = (s == nslices - 1) ? nwords : slice_words * (s + 1);
        for (long i = 0; i < nbranches; i++)
            ready[i] = 1;
        for (long t = 0; t < ntodo; t++)
            ready[todo[t]] = 0;
        long done = 0;
        while (done < ntodo)
        {
            for (long t = 0; t < ntodo; t++)
            {
                long b = todo[t];
                if (ready[b])
                    continue;
                long li = tree[b].left, ri = tree[b].right;
                if (!ready[li] || !ready[ri])
                    continue;
                long ch = 0;
                const uint64_t *l = tree[li].sitestate, *r = tree[ri].sitestate;
                uint64_t *z = tree[b].sitestate;
                for (long j = w0; j < w1; j++)
                    z[j] = lvbo_combine(l[j], r[j], &ch);
                partial[t] += ch;
                ready[b] = 1;
                done++;
            }
        }
        rootsum += root_changes(tree, root, w0, w1);
    }
    for (long t = 0; t < ntodo; t++)
    {
        tree[todo[t]].changes = partial[t];
        total += partial[t];
    }
    total += rootsum;
    free(ready);
    free(partial);
    return total;
}

/* ---------------------------------------------------------------- independent cross-check */

static unsigned plain_node(long node, long site, long nwords, const uint64_t *enc, const long *left,
                           const long *right, long *changes)
{
    if (left[node] < 0) /* leaf */
        return (unsigned)((enc[node * nwords + site / 16] >> (4 * (site % 16))) & 0xF);
    unsigned a = plain_node(left[node], site, nwords, enc, left, right, changes);
    unsigned b = plain_node(right[node], site, nwords, enc, left, right, changes);
    if (a & b)
        return a & b;
    (*changes)++;
    return a | b;
}

long lvbo_fitch_length_plain(long n, long nwords, const uint64_t *enc, const long *left,
                             const long *right, long root)
{
    /* Textbook Fitch, one site at a time on unpacked nibbles: no SWAR, no dirty flags.
     * The tree is rooted at leaf `root`, whose record holds the two top children; the root
     * leaf's own set is combined last, as in the reference (TreeEvaluation.c:238-264). */
    (void)n;
    long changes = 0;
    for (long site = 0; site < nwords * 16; site++)
    {
        unsigned a = plain_node(left[root], site, nwords, enc, left, right, &changes);
        unsigned b = plain_node(right[root], site, nwords, enc, left, right, &changes);
        unsigned z;
        if (a & b)
            z = a & b;
        else
        {
            z = a | b;
            changes++;
        }
        unsigned s = (unsigned)((enc[root * nwords + site / 16] >> (4 * (site % 16))) & 0xF);
        if (!(z & s))
            changes++;
    }
    return changes;
}
