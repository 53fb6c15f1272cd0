// getplen_adapter.cpp - the reference's own getplen() signature over the HIP path.
//
// Linking the reference's objects with this library in place of its TreeEvaluation.o (the only
// translation unit that defines getplen) makes the unmodified search host - Anneal(),
// StartingTemperature(), mutate_*, treestack - score every tree on the MI355X.  The reference
// compiles everything as C++ without extern "C", so the symbol is the mangled
//   _Z7getplenP4dataP20TREESTACK_TREE_NODES10ParameterslPlS4_Pi
// and the types below are layout- and name-compatible re-declarations of
//   DataStructure.h:63-81  struct data (120 bytes; fields read here: n @24,
//                          numberofpossiblebranches @40, nwords @72)
//   LVB.h:121-128          TREESTACK_TREE_NODES (40 bytes)
//   DataStructure.h:86-97  Parameters (4040 bytes, passed BY VALUE, unused by getplen)
// (sizes/offsets are asserted against the compiled reference by the layout test under tests/).
//
// Semantics kept: dirty == sitestate[0]==0; dirty nodes' sitestate and changes are rewritten in
// the caller's tree block; the three scratch arrays are accepted and ignored; failure prints
// "\nFATAL ERROR: ..." on stdout and exits with EXIT_FAILURE (Error.c:49-67), including the
// reference's own assertion changes > 0 (TreeEvaluation.c:267).
// There is no CPU path: without a HIP device the first call is a FATAL ERROR.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/lvbgpu.h"

typedef struct data
{
    int n_threads_getplen;
    int n_slice_size_getplen;
    long m;
    long original_m;
    long n;
    long max_length_seq_name;
    long numberofpossiblebranches;
    long bytes;
    long tree_bytes;
    long tree_bytes_without_sitestate;
    long nwords;
    long min_len_tree;
    long nsets;
    long mssz;
    char **row;
    char **rowtitle;
} *Dataptr, DataStructure;

typedef struct
{
    long parent;
    long left;
    long right;
    long changes;
    uint64_t *sitestate;
} TREESTACK_TREE_NODES;

#define LVB_FNAMSIZE 2000
typedef struct
{
    int seed;
    int cooling_schedule;
    int algorithm_selection;
    int n_file_format;
    int n_processors_available;
    long verbose;
    char file_name_in[LVB_FNAMSIZE];
    char file_name_out[LVB_FNAMSIZE];
    int n_number_max_trees;
} Parameters;

static_assert(sizeof(DataStructure) == 120, "struct data layout");
static_assert(sizeof(TREESTACK_TREE_NODES) == 40, "node layout");
static_assert(sizeof(Parameters) == 4040, "Parameters layout");

namespace
{

struct Bound
{
    lvbgpu_ctx *ctx = nullptr;
    long n = 0, nwords = 0;
    ~Bound()
    {
        if (ctx)
            lvbgpu_destroy(ctx);
    }
};
Bound g_bound;

[[noreturn]] void fatal(const char *what, const char *detail)
{
    // the reference's crash(): message on STDOUT, flush, exit(EXIT_FAILURE)
    printf("\nFATAL ERROR: %s%s%s\n", what, detail && *detail ? ": " : "", detail ? detail : "");
    fflush(stdout);
    exit(EXIT_FAILURE);
}

int device_from_env()
{
    const char *e = getenv("LVBGPU_DEVICE");
    return e ? atoi(e) : 0;
}

lvbgpu_ctx *context_for(Dataptr MSA, TREESTACK_TREE_NODES *tree)
{
    if (g_bound.ctx && g_bound.n == MSA->n && g_bound.nwords == MSA->nwords)
        return g_bound.ctx;
    if (g_bound.ctx)
    {
        lvbgpu_destroy(g_bound.ctx);
        g_bound.ctx = nullptr;
    }
    // leaf i's set is taxon i's encoded row (ss_init, TreeOperations.c:1500-1513)
    std::vector<uint64_t> leaves((size_t)MSA->n * MSA->nwords);
    for (long i = 0; i < MSA->n; i++)
        for (long j = 0; j < MSA->nwords; j++)
            leaves[(size_t)i * MSA->nwords + j] = tree[i].sitestate[j];
    const int rc = lvbgpu_create(&g_bound.ctx, device_from_env(), MSA->n, MSA->nwords, leaves.data(), MSA->nwords);
    if (rc != LVBGPU_OK)
        fatal("cannot set up the MI355X scoring path", lvbgpu_last_error(nullptr));
    g_bound.n = MSA->n;
    g_bound.nwords = MSA->nwords;
    return g_bound.ctx;
}

} // namespace

long getplen(Dataptr MSA, TREESTACK_TREE_NODES *BranchArray, Parameters rcstruct, const long root, long *p_todo_arr,
             long *p_todo_arr_sum_changes, int *p_runs)
{
    (void)rcstruct; // never read by the reference either
    (void)p_todo_arr;
    (void)p_todo_arr_sum_changes;
    (void)p_runs;
    if (MSA->numberofpossiblebranches != 2 * MSA->n - 3)
        fatal("getplen", "numberofpossiblebranches != 2n-3");
    lvbgpu_ctx *ctx = context_for(MSA, BranchArray);
    int64_t len = 0;
    const int rc = lvbgpu_getplen_compat(ctx, BranchArray, root, &len);
    if (rc == LVBGPU_E_ZEROLEN)
        fatal("assertion failed at 'getplen_adapter.cpp' (reference TreeEvaluation.c line 267)", "changes > 0");
    if (rc != LVBGPU_OK)
        fatal(lvbgpu_strerror(rc), lvbgpu_last_error(ctx));
    return (long)len;
}

// The reference keeps these in MemoryOperations.o; they are repeated here (weak) so the adapter
// library is self-contained for hosts that link it alone.  getplen above ignores the arrays.
__attribute__((weak)) void alloc_memory_to_getplen(Dataptr MSA, long **p_todo_arr, long **p_todo_arr_sum_changes,
                                                   int **p_runs)
{
    const long internal = MSA->numberofpossiblebranches - MSA->n;
    const int threads = MSA->n_threads_getplen > 0 ? MSA->n_threads_getplen : 1;
    *p_todo_arr = (long *)malloc((size_t)internal * sizeof(long));
    *p_todo_arr_sum_changes = (long *)malloc((size_t)threads * (1 + internal) * sizeof(long));
    *p_runs = (int *)malloc((size_t)threads * internal * sizeof(int));
}

__attribute__((weak)) void free_memory_to_getplen(long **p_todo_arr, long **p_todo_arr_sum_changes, int **p_runs)
{
    free(*p_todo_arr);
    free(*p_todo_arr_sum_changes);
    free(*p_runs);
}
