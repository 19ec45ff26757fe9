/*
 * Atom-pair FEP kernels for gfx950 (wave64): the drop-in for the reference's gpu_feplist consumers
 *   nbnxn_fep_kernel_<Elec>_<Vdw>_{F,VF}_cuda        (nbnxm/cuda/nbnxm_fep_cuda_kernel.cuh:87-628)
 *   nbnxn_foreign_fep_kernel_<Elec>_<Vdw>_V_cuda     (nbnxm/cuda/nbnxm_foreign_fep_cuda_kernel.cuh:88-583)
 *
 * The reference maps one 32-lane warp to one i-entry (on average 11 of 32 lanes busy).
 * Here the ragged list is flattened: lane p of the grid owns list pair p (pairEntry[p] gives its
 * i-entry), so every lane of a wavefront works; i-forces, shift forces are combined with a
 * segmented wave reduction keyed on the i-entry (one atomic per segment instead of one per pair),
 * energies and dV/dlambda with a wave + LDS block reduction (one atomic per block per quantity;
 * the reference issues 4 atomics per warp per lambda per 32-pair chunk).
 * The foreign-lambda kernel keeps the pair geometry and parameters in registers and loops lambda in-lane.
 */
#ifndef NBNXM_FEP_KERNEL_IMPL_H
#define NBNXM_FEP_KERNEL_IMPL_H

#include "nbnxm_device_helpers.h"

constexpr int c_fepBlockSize = 256;

/* Sum of v over the run of consecutive lanes that share `key` (runs are contiguous);
 * valid in the first lane of each run. */
NB_DEVINL float segmentedSumDown(float v, int key, unsigned lane)
{
#pragma unroll
    for (int off = 1; off < c_waveSize; off <<= 1)
    {
        const float ov = __shfl_down(v, off);
        const int   ok = __shfl_down(key, off);
        if (lane + off < static_cast<unsigned>(c_waveSize) && ok == key) { v += ov; }
    }
    return v;
}

struct FepPairData
{
    bool   valid;
    int    entry, ai, aj, shiftIdx;
    bool   included;
    float3 rv;
    float  r2;
    float  qq[2], c6[2], c12[2], c6grid[2];
};

NB_DEVINL FepPairData loadFepPair(const NBAtomDataGpu& atdat, const NBParamGpu& nbp, const gpu_feplist& feplist, int p)
{
    FepPairData d;
    d.valid = p < feplist.nrj;
    d.entry = -1;
    d.ai = d.aj = 0;
    d.shiftIdx  = c_centralShiftIndex;
    d.included  = true;
    d.rv        = make_float3(0.0F, 0.0F, 0.0F);
    d.r2        = 0.0F;
    d.qq[0] = d.qq[1] = d.c6[0] = d.c6[1] = d.c12[0] = d.c12[1] = d.c6grid[0] = d.c6grid[1] = 0.0F;
    if (d.valid)
    {
        d.entry    = feplist.pairEntry[p];
        d.ai       = feplist.iinr[d.entry];
        d.shiftIdx = feplist.shift[d.entry];
        d.aj       = feplist.jjnr[p];
        d.included = (feplist.excl_fep == nullptr) || (feplist.excl_fep[p] != 0);
        const float4 xqi = atdat.xq[d.ai];
        const float4 xqj = atdat.xq[d.aj];
        const float3 sh  = atdat.shiftVec[d.shiftIdx];
        d.rv             = make_float3(xqi.x + sh.x - xqj.x, xqi.y + sh.y - xqj.y, xqi.z + sh.z - xqj.z);
        d.r2             = d.rv.x * d.rv.x + d.rv.y * d.rv.y + d.rv.z * d.rv.z;
        const float4 q4i = atdat.q4[d.ai];
        const float4 q4j = atdat.q4[d.aj];
        d.qq[0]          = nbp.epsfac * q4i.x * q4j.x;
        d.qq[1]          = nbp.epsfac * q4i.y * q4j.y;
        const int4   t4i = atdat.atomTypes4[d.ai];
        const int4   t4j = atdat.atomTypes4[d.aj];
        const float2 pA  = nbp.nbfp[atdat.numTypes * t4i.x + t4j.x];
        const float2 pB  = nbp.nbfp[atdat.numTypes * t4i.y + t4j.y];
        d.c6[0]          = pA.x;
        d.c12[0]         = pA.y;
        d.c6[1]          = pB.x;
        d.c12[1]         = pB.y;
        ljGridC6AB(nbp, t4i, t4j, d.c6grid);
    }
    return d;
}

/* The 64 list pairs p0 .. p0 + 63 by one wavefront: the body of nbnxmFepKernel, also run as trailing workgroups of the cluster kernel
 * (nbnxm_kernel_impl.h) so that a caller who keeps the reference's atom-pair list pays for no second kernel, no second stream and no
 * fork / join events.  No LDS, no barrier.  slotIndex: which energy accumulator slot this wave adds to. */
template<int ELEC, bool PSWITCH, bool ENERGY>
NB_DEVINL void fepAtomPairWave(const NBAtomDataGpu& atdat, const NBParamGpu& nbp, const gpu_feplist& feplist, const int bCalcFshift, const int p0,
                               const int slotIndex)
{
    const unsigned lane = threadIdx.x & (c_waveSize - 1);
    const int      p    = p0 + static_cast<int>(lane);
    float*         f    = reinterpret_cast<float*>(atdat.f);

    const FepLambda   L = makeFepLambda(nbp.lambda_q, nbp.lambda_v, nbp.lam_power, nbp.alpha_coul, nbp.alpha_vdw);
    const FepPairData d = loadFepPair(atdat, nbp, feplist, p);

    float E_lj = 0.0F, E_el = 0.0F, DVDL_lj = 0.0F, DVDL_el = 0.0F;
    float fscal = 0.0F;
    if (d.valid)
    {
        float      fs   = 0.0F;
        const bool done = fepPair<ELEC, PSWITCH, true, ENERGY>(nbp, L, d.r2, d.included, d.ai == d.aj, d.qq, d.c6,
                                                               d.c12, fs, E_lj, E_el, DVDL_lj, DVDL_el, d.c6grid);
        fscal           = done ? fs : 0.0F;
    }
    const float3 f_ij = make_float3(d.rv.x * fscal, d.rv.y * fscal, d.rv.z * fscal);
    if (fscal != 0.0F)
    {
        atomicAdd(&f[3 * d.aj + 0], -f_ij.x);
        atomicAdd(&f[3 * d.aj + 1], -f_ij.y);
        atomicAdd(&f[3 * d.aj + 2], -f_ij.z);
    }
    /* i-force and shift force: one atomic triple per run of equal i-entries in the wave */
    const float fix  = segmentedSumDown(f_ij.x, d.entry, lane);
    const float fiy  = segmentedSumDown(f_ij.y, d.entry, lane);
    const float fiz  = segmentedSumDown(f_ij.z, d.entry, lane);
    const int   prev = __shfl_up(d.entry, 1);
    const bool  head = d.valid && (lane == 0U || prev != d.entry);
    if (head && (fix != 0.0F || fiy != 0.0F || fiz != 0.0F))
    {
        atomicAdd(&f[3 * d.ai + 0], fix);
        atomicAdd(&f[3 * d.ai + 1], fiy);
        atomicAdd(&f[3 * d.ai + 2], fiz);
        if (bCalcFshift && d.shiftIdx != c_centralShiftIndex)
        {
            float* fs = reinterpret_cast<float*>(atdat.fShift) + 3 * d.shiftIdx;
            atomicAdd(&fs[0], fix);
            atomicAdd(&fs[1], fiy);
            atomicAdd(&fs[2], fiz);
        }
    }

    if constexpr (ENERGY)
    {
        /* into one of the accumulator slots the cluster kernel uses (same layout: E_lj, E_el, dV/dl_lj, dV/dl_el; summed on the host
         * with the staged scalars): hundreds of waves adding to ONE set of addresses serialise in L2 */
        const float v = waveSum4Transposed(E_lj, E_el, DVDL_lj, DVDL_el, lane); /* lanes 0 .. 3: the four sums */
        if (lane < 4U && v != 0.0F)
        {
            atomicAdd(atdat.energySlots + (slotIndex & (c_numEnergySlots - 1)) * c_energySlotStride + static_cast<int>(lane), v);
        }
    }
}

template<int ELEC, bool PSWITCH, bool ENERGY>
__launch_bounds__(c_fepBlockSize) __global__
        void nbnxmFepKernel(const NBAtomDataGpu atdat, const NBParamGpu nbp, const gpu_feplist feplist, const int bCalcFshift)
{
    const int wave = static_cast<int>(blockIdx.x) * (c_fepBlockSize / c_waveSize) + static_cast<int>(threadIdx.x) / c_waveSize;
    fepAtomPairWave<ELEC, PSWITCH, ENERGY>(atdat, nbp, feplist, bCalcFshift, wave * c_waveSize, wave);
}

/* Energies and dV/dlambda at lambda index 0 (current) .. n_lambda (foreign); results go to
 * eLJForeign / eElecForeign / dvdlLJForeign / dvdlElecForeign [idx]. */
template<int ELEC, bool PSWITCH>
__launch_bounds__(c_fepBlockSize) __global__
        void nbnxmFepForeignKernel(const NBAtomDataGpu atdat, const NBParamGpu nbp, const gpu_feplist feplist, const int n_lambda)
{
    const unsigned lane = threadIdx.x & (c_waveSize - 1);
    const unsigned w    = threadIdx.x / c_waveSize;
    const int      p    = static_cast<int>(blockIdx.x) * c_fepBlockSize + static_cast<int>(threadIdx.x);

    const FepPairData d = loadFepPair(atdat, nbp, feplist, p);

    /* wave sums of all lambda indices go to LDS, ONE block reduction at the end (dynamic LDS: waves x 4 x (n_lambda + 1) floats;
     * the reference's kernel does 4 atomics per lambda index and 32-pair chunk, nbnxm_foreign_fep_cuda_kernel.cuh:560-580) */
    extern __shared__ float red[];
    const int               numTerms = 4 * (n_lambda + 1);

    for (int idx = 0; idx <= n_lambda; idx++)
    {
        const float     lc = (idx == 0) ? nbp.lambda_q : nbp.allLambdaCoul[idx - 1];
        const float     lv = (idx == 0) ? nbp.lambda_v : nbp.allLambdaVdw[idx - 1];
        const FepLambda L  = makeFepLambda(lc, lv, nbp.lam_power, nbp.alpha_coul, nbp.alpha_vdw);
        float E_lj = 0.0F, E_el = 0.0F, DVDL_lj = 0.0F, DVDL_el = 0.0F, fs = 0.0F;
        if (d.valid)
        {
            fepPair<ELEC, PSWITCH, false, true>(nbp, L, d.r2, d.included, d.ai == d.aj, d.qq, d.c6, d.c12, fs, E_lj,
                                                E_el, DVDL_lj, DVDL_el, d.c6grid);
        }
        E_lj    = waveSum(E_lj);
        E_el    = waveSum(E_el);
        DVDL_lj = waveSum(DVDL_lj);
        DVDL_el = waveSum(DVDL_el);
        if (lane == 0U)
        {
            /* layout [E_lj | E_el | dV/dl_lj | dV/dl_el][lambda index], as in the accumulator slots */
            float* r                    = red + w * numTerms + idx;
            r[0]                        = E_lj;
            r[n_lambda + 1]             = E_el;
            r[2 * (n_lambda + 1)]       = DVDL_lj;
            r[3 * (n_lambda + 1)]       = DVDL_el;
        }
    }
    __syncthreads();
    for (int t = static_cast<int>(threadIdx.x); t < numTerms; t += c_fepBlockSize)
    {
        float s = 0.0F;
#pragma unroll
        for (int k = 0; k < c_fepBlockSize / c_waveSize; k++) { s += red[k * numTerms + t]; }
        /* foreign-lambda accumulator slots: hundreds of work-groups adding to ONE set of addresses serialise in L2 */
        float* out = atdat.foreignSlots + (blockIdx.x & (c_numForeignSlots - 1)) * atdat.foreignSlotStride + t;
        if (s != 0.0F) { atomicAdd(out, s); }
    }
}

#endif
