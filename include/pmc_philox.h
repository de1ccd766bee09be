/* pmc_philox.h -- the per-history random stream of the MI355X engine (host + device).
 *
 * The reference draws from one std::mt19937_64 per thread (SKIRT/core/Random.cpp:18-54), which makes results
 * depend on the thread/rank layout.  The engine instead gives EVERY photon history its own counter-based stream:
 * Philox4x32-10 (Salmon et al., SC'11) with key = 64-bit seed and counter = (history index, block number), so that
 * results do not depend on how histories are mapped to lanes, workgroups or GPUs.  One block yields two uniform
 * deviates in the open interval ]0,1[ with 52 random bits each -- the same interval as Random::uniform().
 *
 * This header is the single definition of that stream; the device kernels and the CPU test oracle both include it.
 */
#ifndef PMC_PHILOX_H
#define PMC_PHILOX_H

#include <stdint.h>

#if defined(__HIPCC__)
    #define PMC_HD __host__ __device__ inline
#else
    #define PMC_HD static inline
#endif

typedef struct pmc_rng
{
    uint32_t key0, key1;     /* seed */
    uint32_t ctr0, ctr1;     /* history index */
    uint32_t block;          /* next block number */
    uint32_t have;           /* 1 if 'spare' holds the second deviate of the last block */
    double   spare;
} pmc_rng;

PMC_HD void pmc_philox_round(uint32_t* c, uint32_t k0, uint32_t k1)
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

PMC_HD void pmc_philox4x32_10(uint32_t* c, uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r)
    {
        pmc_philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

PMC_HD void pmc_rng_init(pmc_rng* g, uint64_t seed, uint64_t history)
{
    g->key0 = (uint32_t)seed;
    g->key1 = (uint32_t)(seed >> 32);
    g->ctr0 = (uint32_t)history;
    g->ctr1 = (uint32_t)(history >> 32);
    g->block = 0;
    g->have = 0;
    g->spare = 0.;
}

PMC_HD double pmc_bits_to_unit(uint32_t hi, uint32_t lo)
{
    const uint64_t x = (((uint64_t)hi << 32) | lo) >> 12;      /* 52 random bits */
    return ((double)x + 0.5) * 2.220446049250313e-16;          /* (x + 1/2) * 2^-52, exact, in ]0,1[ */
}

PMC_HD double pmc_rng_uniform(pmc_rng* g)
{
    if (g->have)
    {
        g->have = 0;
        return g->spare;
    }
    uint32_t c[4] = {g->ctr0, g->ctr1, g->block, 0x504d4331u /* "PMC1" */};
    g->block += 1;
    pmc_philox4x32_10(c, g->key0, g->key1);
    g->spare = pmc_bits_to_unit(c[2], c[3]);
    g->have = 1;
    return pmc_bits_to_unit(c[0], c[1]);
}

#endif
