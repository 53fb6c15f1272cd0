/* decide.h - the accept decision of an annealing step, as ONE function for the device (fitch_kernels.hip: the scoring
 * walk's watcher waves), for the library's host fallback (api_propose.cpp) and for the scorer's test double
 * (tests/cpu_double, plain C): C99 and C++ alike.
 *
 * One rule per chain of a step: the chain's current length, its temperature and energy scale, the seed of the step's
 * Metropolis draws, and where the chain's candidates sit in the batch.  Candidate j (index within the chain's draw) is
 * TAKEN if it is no longer than the current tree, or - Solve.c:303-378 - with probability exp(-deltah / t), where
 * deltah = minlen / cur - minlen / len (capped at 1), and never once -deltah < t log(LVB_EPS).  Its uniform draw is a
 * function of (seed, j) alone, so the decision needs no state and every candidate decides for itself: the chain's pick
 * is the SMALLEST taken j, which is what consuming the candidates in order gives. */
#ifndef LVB_DECIDE_H
#define LVB_DECIDE_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LVB_HD __host__ __device__
#else
#define LVB_HD
#endif

#define LVB_OVERFLOW_LENGTH (1ll << 61) /* kernels.hpp PROPOSAL_OVERFLOW_LENGTH: "not a proposal" */
#define LVB_PICK_NONE 0xFFFFFFFFu

typedef struct
{
    long long cur;           /* current tree length */
    double t;                /* temperature */
    double minlen;           /* MinimumTreeLength of the alignment (the energy scale) */
    unsigned long long seed; /* of this step's acceptance draws */
    uint32_t start, count;   /* the chain's candidates are [start, start + count) of the batch */
} DecideRule;

LVB_HD static inline unsigned long long lvb_mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* uniform in [0, 1) with 53 bits, a function of (seed, j) */
LVB_HD static inline double lvb_uniform(unsigned long long seed, uint32_t j)
{
    const unsigned long long r = lvb_mix64(seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)j + 1ull));
    return (double)(r >> 11) * (1.0 / 9007199254740992.0);
}
LVB_HD static inline int lvb_take(long long len, const DecideRule *r, uint32_t j)
{
    double deltah;
    if (len <= 0 || len >= LVB_OVERFLOW_LENGTH)
        return 0; /* not a proposal (a candidate that did not fit the generator's buffers) */
    if (len <= r->cur)
        return 1;
    deltah = r->minlen / (double)r->cur - r->minlen / (double)len;
    if (deltah > 1.0)
        deltah = 1.0;
    if (-deltah < r->t * -25.328436022934504) /* t log(LVB_EPS), LVB_EPS = 1e-11 (LVB.h:102) */
        return 0;
    return lvb_uniform(r->seed, j) < exp(-deltah / r->t);
}
#endif /* LVB_DECIDE_H */
