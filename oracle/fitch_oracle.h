/* oracle/fitch_oracle.h - TEST INFRASTRUCTURE ONLY (see fitch_oracle.c header). */
#ifndef LVB_AMD_FITCH_ORACLE_H
#define LVB_AMD_FITCH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LVBO_UNSET (-1L)

/* layout-compatible with the reference's TREESTACK_TREE_NODES (LVB.h:121-128): 40 bytes */
typedef struct
{
    long parent;
    long left;
    long right;
    long changes;
    uint64_t *sitestate;
} lvbo_node;

/* encoding (DataOperations.c) */
long lvbo_words_per_row(long m);
int lvbo_encode_char(char base);
int lvbo_encode_row(const char *row, long m, long nwords, uint64_t *out);
long lvbo_variable_columns(long n, long m, const char *const *rows, unsigned char *keep);
long lvbo_min_tree_length(long n, long m, const char *const *rows);

/* the SWAR Fitch word step (TreeEvaluation.c:219-230) */
uint64_t lvbo_combine(uint64_t x, uint64_t y, long *changes);

/* tree block (TreeOperations.c) */
long lvbo_tree_bytes(long nbranches, long nwords);
lvbo_node *lvbo_treealloc(long nbranches, long nwords);
void lvbo_tree_set_topology(lvbo_node *tree, long nbranches, const long *parent, const long *left,
                            const long *right);
void lvbo_ss_init(lvbo_node *tree, long n, long nbranches, long nwords, const uint64_t *enc);
void lvbo_mark_dirty(lvbo_node *tree, long node);
void lvbo_make_dirty_below(lvbo_node *tree, long node);
void lvbo_treecopy(lvbo_node *dest, const lvbo_node *src, long nbranches, long nwords);

/* getplen (TreeEvaluation.c) */
long lvbo_getplen(lvbo_node *tree, long n, long nbranches, long nwords, long root, long *todo);
long lvbo_getplen_sliced(lvbo_node *tree, long n, long nbranches, long nwords, long root,
                         long *todo, int nslices, long slice_words);

/* independent cross-check: plain recursive Fitch on unpacked nibbles, no dirty flags, no SWAR */
long lvbo_fitch_length_plain(long n, long nwords, const uint64_t *enc, const long *left,
                             const long *right, long root);

#ifdef __cplusplus
}
#endif
#endif
