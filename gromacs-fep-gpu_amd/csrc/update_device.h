/*
 * Device helpers shared by the update kernels (update_langevin.hip, update_constrain.hip): small vector algebra, the
 * wave-wide virial reduction, the SETTLE triangle solve and the random numbers of the stochastic-dynamics update.
 */
#ifndef NBNXM_UPDATE_DEVICE_H
#define NBNXM_UPDATE_DEVICE_H

#include <hip/hip_runtime.h>

#include "device_utils.h"
#include "pbc_aiuc.h"

namespace nbnxm_hip
{

constexpr int                c_updateBlock       = 256;
constexpr int                c_tableBits         = 14; /* langevin_gpu.h:79 */
constexpr unsigned long long c_domainUpdateCoord = 0x00003000ULL; /* random/seed.h:95 */

__device__ __forceinline__ float3 operator+(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ float3 operator-(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 operator*(float s, float3 a) { return make_float3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float  dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float3 cross3(float3 a, float3 b)
{
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* sum over the 64 lanes of a wave, result in every lane */
__device__ __forceinline__ float waveSum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { v += __shfl_xor(v, m); }
    return v;
}

/* Virial accumulators: c_numVirialSlots copies of [XX XY XZ YY YZ ZZ - -], one 32-byte record each, summed on the host.
 * Hundreds of waves adding to ONE set of six addresses serialise in L2 at ~10 ns per add: measured 41 us instead of 3.9 us
 * for SETTLE on 32k waters (tools/settle_virial_probe.py). */
constexpr int c_numVirialSlots   = 64;
constexpr int c_virialSlotStride = 8;
constexpr int c_virialFloats     = c_numVirialSlots * c_virialSlotStride;

/* the six independent components of a symmetric tensor summed over the wave, added to one of the accumulator copies once per wave */
__device__ __forceinline__ void addWaveVirial(float* virial, const float (&c)[6])
{
    const int lane = static_cast<int>(threadIdx.x) & 63;
    const int wave = static_cast<int>(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    float*    slot = virial + (wave & (c_numVirialSlots - 1)) * c_virialSlotStride;
#pragma unroll
    for (int d = 0; d < 6; d++)
    {
        const float s = waveSum(c[d]);
        if (lane == 0 && s != 0.0F) { atomicAdd(&slot[d], s); }
    }
}

/* host: sum of the accumulator copies into six numbers */
inline void sumVirialSlots(const float* slots, float* six)
{
    for (int d = 0; d < 6; d++)
    {
        double s = 0;
        for (int k = 0; k < c_numVirialSlots; k++) { s += slots[k * c_virialSlotStride + d]; }
        six[d] = static_cast<float>(s);
    }
}

/* the temperature-scaling factors of a step travel as kernel arguments (no staging copy, nothing to wait for) when
 * they fit; more groups than this go through a device buffer */
constexpr int c_maxLambdasInArgs = 32;
struct TcLambdas
{
    float v[c_maxLambdasInArgs];
};

struct SettlePars /* mdlib/settle.h:68-100, the part the coordinate constraint needs */
{
    float mO, mH, wh, ra, rb, rc, irc2;
};

/* SETTLE for one water (settle_gpu_internal.cu:140-305): dist21 / dist31 = H - O before the update, doh2 / doh3 = H - O after
 * it (minimum images); returns the displacement of the three atoms that restores the triangle */
__device__ __forceinline__ void settleTriangle(const SettlePars& pars, float3 dist21, float3 dist31, float3 doh2, float3 doh3, float3& dxO,
                                               float3& dxH2, float3& dxH3)
{
    /* the updated triangle relative to its mass centre (O as reference point: no centre of mass is formed) */
    const float3 a1 = (-pars.wh) * (doh2 + doh3);
    const float3 b1 = doh2 + a1;
    const float3 c1 = doh3 + a1;

    /* frame: ez normal to the old triangle, ex = a1 x ez, ey = ez x ex */
    float3 ez = cross3(dist21, dist31);
    float3 ex = cross3(a1, ez);
    float3 ey = cross3(ez, ex);
    ex        = rsqrtf(dot3(ex, ex)) * ex;
    ey        = rsqrtf(dot3(ey, ey)) * ey;
    ez        = rsqrtf(dot3(ez, ez)) * ez;

    const float b0dx = dot3(ex, dist21), b0dy = dot3(ey, dist21);
    const float c0dx = dot3(ex, dist31), c0dy = dot3(ey, dist31);
    const float a1dz = dot3(ez, a1);
    const float b1dx = dot3(ex, b1), b1dy = dot3(ey, b1), b1dz = dot3(ez, b1);
    const float c1dx = dot3(ex, c1), c1dy = dot3(ey, c1), c1dz = dot3(ez, c1);

    const float sinphi = a1dz * rsqrtf(pars.ra * pars.ra);
    float       tmp2   = fmaxf(1.0F - sinphi * sinphi, 1e-12F);
    const float tmp    = rsqrtf(tmp2);
    const float cosphi = tmp2 * tmp;
    const float sinpsi = (b1dz - c1dz) * pars.irc2 * tmp;
    tmp2               = 1.0F - sinpsi * sinpsi;
    const float cospsi = tmp2 * rsqrtf(tmp2);

    const float a2dy = pars.ra * cosphi;
    const float b2dx = -pars.rc * cospsi;
    const float t1   = -pars.rb * cosphi;
    const float t2   = pars.rc * sinpsi * sinphi;
    const float b2dy = t1 - t2;
    const float c2dy = t1 + t2;

    const float alpha  = b2dx * (b0dx - c0dx) + b0dy * b2dy + c0dy * c2dy;
    const float beta   = b2dx * (c0dy - b0dy) + b0dx * b2dy + c0dx * c2dy;
    const float gamma  = b0dx * b1dy - b1dx * b0dy + c0dx * c1dy - c1dx * c0dy;
    const float al2be2 = alpha * alpha + beta * beta;
    tmp2               = al2be2 - gamma * gamma;
    const float sinthe = (alpha * gamma - beta * tmp2 * rsqrtf(tmp2)) * rsqrtf(al2be2 * al2be2);
    tmp2               = 1.0F - sinthe * sinthe;
    const float costhe = tmp2 * rsqrtf(tmp2);

    const float3 a3d = make_float3(-a2dy * sinthe, a2dy * costhe, a1dz);
    const float3 b3d = make_float3(b2dx * costhe - b2dy * sinthe, b2dx * sinthe + b2dy * costhe, b1dz);
    const float3 c3d = make_float3(-b2dx * costhe - c2dy * sinthe, -b2dx * sinthe + c2dy * costhe, c1dz);

    dxO  = (a3d.x * ex + a3d.y * ey + a3d.z * ez) - a1;
    dxH2 = (b3d.x * ex + b3d.y * ey + b3d.z * ez) - b1;
    dxH3 = (c3d.x * ex + c3d.y * ey + c3d.z * ez) - c1;
}

/* scaled-virial contribution of one settled water (settle_gpu_internal.cu:325-345): xo = position of the oxygen before the update */
__device__ __forceinline__ void settleVirial(const SettlePars& pars, float3 xo, float3 dist21, float3 dist31, float3 dxO, float3 dxH2,
                                             float3 dxH3, float (&vir)[6])
{
    const float3 mdb = pars.mH * dxH2;
    const float3 mdc = pars.mH * dxH3;
    const float3 mdo = pars.mO * dxO + mdb + mdc;
    vir[0] -= xo.x * mdo.x + dist21.x * mdb.x + dist31.x * mdc.x;
    vir[1] -= xo.x * mdo.y + dist21.x * mdb.y + dist31.x * mdc.y;
    vir[2] -= xo.x * mdo.z + dist21.x * mdb.z + dist31.x * mdc.z;
    vir[3] -= xo.y * mdo.y + dist21.y * mdb.y + dist31.y * mdc.y;
    vir[4] -= xo.y * mdo.z + dist21.y * mdb.z + dist31.y * mdc.z;
    vir[5] -= xo.z * mdo.z + dist21.z * mdb.z + dist31.z * mdc.z;
}

__device__ __forceinline__ unsigned long long rotl64(unsigned long long v, unsigned b)
{
    return (v << b) | (v >> (64U - b));
}

/* first word of the Threefry-2x64-20 block of (key, counter) (random/threefry.h:420-600, a published counter-based generator) */
__device__ __forceinline__ unsigned long long threefry2x64First(unsigned long long k0, unsigned long long k1, unsigned long long c0,
                                                                unsigned long long c1)
{
    const unsigned long long ks[3] = { k0, k1, 0x1bd11bdaa9fc1a22ULL ^ k0 ^ k1 };
    constexpr unsigned       rot[8] = { 16, 42, 12, 31, 16, 32, 24, 21 };
    unsigned long long       x0 = c0 + ks[0], x1 = c1 + ks[1];
#pragma unroll
    for (unsigned r = 0; r < 20; r++)
    {
        x0 += x1;
        x1 = rotl64(x1, rot[r % 8]);
        x1 ^= x0;
        if (((r + 1) & 3) == 0)
        {
            const unsigned r4 = (r + 1) >> 2;
            x0 += ks[r4 % 3];
            x1 += ks[(r4 + 1) % 3] + r4;
        }
    }
    return x0;
}

/* three numbers of the tabulated unit normal for (seed, step, atom): 42 of the first 64 bits of the atom's Threefry block */
__device__ __forceinline__ float3 langevinNoise(const float* __restrict__ table, int seed, int step, int atom)
{
    const unsigned long long bits = threefry2x64First(static_cast<unsigned long long>(static_cast<long long>(seed)), c_domainUpdateCoord,
                                                      static_cast<unsigned long long>(static_cast<long long>(step)),
                                                      static_cast<unsigned long long>(atom));
    constexpr unsigned mask = (1U << c_tableBits) - 1U;
    return make_float3(table[bits & mask], table[(bits >> c_tableBits) & mask], table[(bits >> (2 * c_tableBits)) & mask]);
}

} // namespace nbnxm_hip

/* the stochastic-dynamics integrator's state (update_langevin.hip owns it; the fused update reads the device tables) */
struct LangevinGpu
{
    nbnxm_hip::DeviceStream         stream;
    int                             numGroups = 0, numAtoms = 0, atomsAlloc = 0;
    float*                          d_sdSigmaV      = nullptr;
    float*                          d_sdConstEm     = nullptr;
    float*                          d_table         = nullptr;
    float*                          d_inverseMasses = nullptr;
    unsigned short*                 d_tcGroups      = nullptr;
    nbnxm_hip::PinnedBuffer<float>          h_im;
    nbnxm_hip::PinnedBuffer<unsigned short> h_tc;
};

#endif
