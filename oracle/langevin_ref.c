/* See langevin_ref.h. */
#include "langevin_ref.h"

#include <math.h>
#include <stdlib.h>

#define BOLTZ 0.0083144626181532 /* kJ mol^-1 K^-1 (math/units.h c_boltz) */
#define DOMAIN_UPDATE_COORDINATES 0x00003000ULL /* random/seed.h:95 */

static uint64_t rotl(uint64_t v, unsigned b)
{
    return (v << b) | (v >> (64 - b));
}

void oracle_threefry2x64(uint64_t key0, uint64_t key1, uint64_t ctr0, uint64_t ctr1, uint64_t out[2])
{
    static const unsigned rot[8] = { 16, 42, 12, 31, 16, 32, 24, 21 };
    const uint64_t        ks[3]  = { key0, key1, 0x1bd11bdaa9fc1a22ULL ^ key0 ^ key1 };
    uint64_t              x0 = ctr0 + ks[0], x1 = ctr1 + ks[1];
    for (unsigned r = 0; r < 20; r++)
    {
        x0 += x1;
        x1 = rotl(x1, rot[r % 8]);
        x1 ^= x0;
        if (((r + 1) & 3) == 0)
        {
            const unsigned r4 = (r + 1) >> 2;
            x0 += ks[r4 % 3];
            x1 += ks[(r4 + 1) % 3] + r4;
        }
    }
    out[0] = x0;
    out[1] = x1;
}

/* inverse error function in double: Newton-Halley iterations on erf from a logarithmic starting guess */
static double erfinv_d(double y)
{
    if (y <= -1.0) { return -INFINITY; }
    if (y >= 1.0) { return INFINITY; }
    const double a  = 0.147;
    const double ln = log(1.0 - y * y);
    const double t  = 2.0 / (M_PI * a) + 0.5 * ln;
    double       x  = copysign(sqrt(sqrt(t * t - ln / a) - t), y);
    for (int it = 0; it < 60; it++)
    {
        const double err = erf(x) - y;
        const double d   = 2.0 / sqrt(M_PI) * exp(-x * x);
        const double dx  = err / (d - x * err); /* Halley: f'' / f' = -2x */
        x -= dx;
        if (fabs(dx) <= 1e-16 * fabs(x)) { break; }
    }
    return x;
}

void oracle_normal_table(int bits, float* table)
{
    const int size = 1 << bits, half = size / 2;
    for (int i = 0; i < half - 1; i++)
    {
        const double r = (i + 0.5) / half;
        const double x = sqrt(2.0) * erfinv_d(r);
        table[half - 1 - i] = (float)(-x);
        table[half + i]     = (float)x;
    }
    double sumsq = 0;
    for (int i = 1; i < half; i++) { sumsq += (double)table[i] * (double)table[i]; }
    const double missing  = 1.0 - 2.0 * sumsq / size;
    const double extremal = sqrt(0.5 * missing * size);
    table[0]              = (float)(-extremal);
    table[size - 1]       = (float)extremal;
}

static const float* table14(void)
{
    static float* t = NULL;
    if (t == NULL)
    {
        t = (float*)malloc(sizeof(float) << 14);
        oracle_normal_table(14, t);
    }
    return t;
}

void oracle_tabulated_normal(uint64_t key0, uint64_t domain, int internalCounterBits, uint64_t ctr0, uint64_t ctr1, float mean,
                             float stddev, int n, float* out)
{
    uint64_t key1 = domain;
    if (internalCounterBits > 0)
    {
        /* the high log2(bits) + 1 bits of the key hold bits - 1 (threefry.h:676-686) */
        int lg = 0;
        while ((1 << (lg + 1)) <= internalCounterBits) { lg++; }
        key1 += (uint64_t)(internalCounterBits - 1) << (64 - (lg + 1));
    }
    const float* tab = table14();
    uint64_t     block[2];
    int          index = 2;
    uint64_t     saved = 0;
    int          left  = 0;
    int          first = 1;
    for (int i = 0; i < n; i++)
    {
        if (left < 14)
        {
            if (index >= 2)
            {
                if (!first) { ctr1 += 1ULL << (64 - internalCounterBits); } /* internal counter in the high bits */
                first = 0;
                oracle_threefry2x64(key0, key1, ctr0, ctr1, block);
                index = 0;
            }
            saved = block[index++];
            left  = 64;
        }
        out[i] = mean + tab[saved & 0x3FFFULL] * stddev;
        saved >>= 14;
        left -= 14;
    }
}

void oracle_langevin_update(int updateType, int numAtoms, float* x, float* xp, float* v, const float* f, const float* inverseMasses,
                            const unsigned short* tcGroups, int numGroups, const float* refT, const float* tauT, float dt, int seed,
                            int step)
{
    const float* tab = table14();
    float*       em  = (float*)malloc(sizeof(float) * numGroups);
    float*       sv  = (float*)malloc(sizeof(float) * numGroups);
    for (int g = 0; g < numGroups; g++)
    {
        em[g]          = (tauT[g] > 0) ? (float)exp(-dt / tauT[g]) : 1.0F;
        const float kT = (float)(BOLTZ * refT[g]);
        sv[g]          = sqrtf(kT * (1 - em[g] * em[g]));
    }
    for (int a = 0; a < numAtoms; a++)
    {
        if (updateType == 0)
        {
            const float imdt = inverseMasses[a] * dt;
            for (int d = 0; d < 3; d++)
            {
                xp[3 * a + d] = x[3 * a + d];
                v[3 * a + d]  = v[3 * a + d] + f[3 * a + d] * imdt;
                x[3 * a + d] += v[3 * a + d] * dt;
            }
        }
        else
        {
            uint64_t block[2];
            oracle_threefry2x64((uint64_t)(int64_t)seed, DOMAIN_UPDATE_COORDINATES, (uint64_t)(int64_t)step, (uint64_t)a, block);
            const int   g   = tcGroups[a];
            const float ism = sqrtf(inverseMasses[a]);
            for (int d = 0; d < 3; d++)
            {
                const float xi = tab[(block[0] >> (14 * d)) & 0x3FFFULL];
                const float vn = v[3 * a + d];
                const float vv = vn * em[g] + ism * sv[g] * xi;
                x[3 * a + d] += 0.5F * (vv - vn) * dt;
                v[3 * a + d] = vv;
            }
        }
    }
    free(em);
    free(sv);
}
