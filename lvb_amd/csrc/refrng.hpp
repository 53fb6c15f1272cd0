// refrng.hpp - the reference's random stream, restated: Marsaglia & Zaman's universal generator
// ("UNI"/RANMAR: a lag-(97,33) subtractive Fibonacci sequence combined with an arithmetic
// sequence, all in 24-bit fractions held in doubles), seeded from one integer the way F. James'
// RMARIN splits it.  Follows RandomNumberGenerator.c:87-132 (uni, rstart), 151-198 (rinit) and
// 235-256 (randpint: round-to-nearest scaling, so 0 and `upper` each get half the weight of the
// values between them).  Needed only where a run has to reproduce the reference's own trajectory
// (refsearch.cpp); the batched search draws from proposals.hpp's Rng.
#pragma once

#include <cstdint>

namespace lvbgpu
{

struct Uni
{
    static constexpr int LAG_LONG = 97, LAG_SHORT = 33;
    static constexpr int32_t MAX_SEED = 900000000; // rinit's accepted range
    double u[LAG_LONG + 1];                        // 1-based, as the Fortran original
    double c;
    int i, j;

    // false when the seed is outside [0, MAX_SEED] (the reference crashes there)
    bool seed(int32_t ijkl)
    {
        if (ijkl < 0 || ijkl > MAX_SEED)
            return false;
        const int ij = ijkl / 30082, kl = ijkl - 30082 * ij;
        int s1 = (ij / 177) % 177 + 2, s2 = ij % 177 + 2, s3 = (kl / 169) % 178 + 1, s4 = kl % 169;
        for (int e = 1; e <= LAG_LONG; e++)
        {
            double frac = 0.0, bit = 0.5;
            for (int b = 0; b < 24; b++) // 24 mantissa bits, most significant first
            {
                const int m = ((s1 * s2 % 179) * s3) % 179;
                s1 = s2;
                s2 = s3;
                s3 = m;
                s4 = (53 * s4 + 1) % 169;
                if (s4 * m % 64 >= 32)
                    frac += bit;
                bit *= 0.5;
            }
            u[e] = frac;
        }
        c = 362436.0 / 16777216.0;
        i = LAG_LONG;
        j = LAG_SHORT;
        return true;
    }

    double uni()
    {
        double x = u[i] - u[j];
        if (x < 0.0)
            x += 1.0;
        u[i] = x;
        if (--i == 0)
            i = LAG_LONG;
        if (--j == 0)
            j = LAG_LONG;
        c -= 7654321.0 / 16777216.0;
        if (c < 0.0)
            c += 16777213.0 / 16777216.0;
        x -= c;
        if (x < 0.0)
            x += 1.0;
        return x;
    }

    // integer in [0, upper]
    int64_t randpint(int64_t upper)
    {
        const double scaled = uni() * (double)upper;
        int64_t r = (int64_t)(scaled + 0.5);
        if (r < 0)
            r = 0;
        else if (r > upper)
            r = upper;
        return r;
    }
};

} // namespace lvbgpu
