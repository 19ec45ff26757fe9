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

/* One item of the list regrouped by cluster pair (gpu_feplist::clItem) by one wavefront, lane = tidxj * 8 + tidxi as in the cluster
 * kernels: lane l owns the atom pair (i-atom tidxi of cluster ci, j-atom tidxj of cluster cj) and evaluates it if the list holds it.
 * Coalesced loads (two clusters), no dependent index chain, i- and j-force reductions over the 8 x 8 layout with 24-lane adds, the
 * four energy sums in one transposing pass.  Runs as trailing workgroups of the cluster kernel (nbnxm_kernel_impl.h). */
template<int ELEC, bool PSWITCH, bool ENERGY>
NB_DEVINL void fepListClusterItem(const NBAtomDataGpu& atdat, const NBParamGpu& nbp, const gpu_feplist& feplist, const int bCalcFshift, const int item)
{
    const unsigned lane  = threadIdx.x & (c_waveSize - 1);
    const unsigned tidxi = lane & 7U;
    const unsigned tidxj = lane >> 3;
    const int4     it    = feplist.clItem[item];
    const int      ci = __builtin_amdgcn_readfirstlane(it.x), cj = __builtin_amdgcn_readfirstlane(it.y);
    const int      shiftIdx = __builtin_amdgcn_readfirstlane(it.z);
    const uint2    listed2  = feplist.clListed[item];
    const uint2    incl2    = feplist.clIncl[item];
    const bool     listed   = (((lane < 32U ? listed2.x : listed2.y) >> (lane & 31U)) & 1U) != 0U;
    const bool     included = (((lane < 32U ? incl2.x : incl2.y) >> (lane & 31U)) & 1U) != 0U;

    float* __restrict__ f = reinterpret_cast<float*>(atdat.f);
    const __amdgpu_buffer_rsrc_t fRsrc =
            __builtin_amdgcn_make_buffer_rsrc(f, 0, atdat.numAtoms * 3 * static_cast<int>(sizeof(float)), 0x00020000);
    const int    ai  = ci * c_clSize + static_cast<int>(tidxi);
    const int    aj  = cj * c_clSize + static_cast<int>(tidxj);
    const float3 sh  = atdat.shiftVec[shiftIdx];
    const float4 xi  = atdat.xq[ai];
    const float4 xj  = atdat.xq[aj];
    const float4 q4i = atdat.q4[ai];
    const float4 q4j = atdat.q4[aj];
    const int4   t4i = atdat.atomTypes4[ai];
    const int4   t4j = atdat.atomTypes4[aj];
    const float3 rv  = make_float3(xi.x + sh.x - xj.x, xi.y + sh.y - xj.y, xi.z + sh.z - xj.z);
    const float  r2  = rv.x * rv.x + rv.y * rv.y + rv.z * rv.z;

    float E_lj = 0.0F, E_el = 0.0F, DVDL_lj = 0.0F, DVDL_el = 0.0F;
    float fscal = 0.0F;
    if (listed)
    {
        const float  qq[2]  = { nbp.epsfac * q4i.x * q4j.x, nbp.epsfac * q4i.y * q4j.y };
        const float2 pA     = nbp.nbfp[atdat.numTypes * t4i.x + t4j.x];
        const float2 pB     = nbp.nbfp[atdat.numTypes * t4i.y + t4j.y];
        const float  c6[2]  = { pA.x, pB.x };
        const float  c12[2] = { pA.y, pB.y };
        float        c6grid[2];
        ljGridC6AB(nbp, t4i, t4j, c6grid);
        const FepLambda L    = makeFepLambda(nbp.lambda_q, nbp.lambda_v, nbp.lam_power, nbp.alpha_coul, nbp.alpha_vdw);
        float           fs   = 0.0F;
        const bool      done = fepPair<ELEC, PSWITCH, true, ENERGY>(nbp, L, r2, included, ai == aj, qq, c6, c12, fs, E_lj, E_el, DVDL_lj, DVDL_el,
                                                                    c6grid);
        fscal                = done ? fs : 0.0F;
    }
    const float3 f_ij = make_float3(rv.x * fscal, rv.y * fscal, rv.z * fscal);
    {
        /* i-forces: sum over tidxj, lanes tidxj 0..2 carry x, y, z; j-forces: sum over tidxi, lanes tidxi 0..2 */
        const float fix = reduceOverTidxj(f_ij.x), fiy = reduceOverTidxj(f_ij.y), fiz = reduceOverTidxj(f_ij.z);
        const float vi  = (tidxj == 0U) ? fix : ((tidxj == 1U) ? fiy : fiz);
        const int   oi  = (tidxj < 3U) ? (3 * ai + static_cast<int>(tidxj)) * static_cast<int>(sizeof(float)) : 0x7FFFFFF0;
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(vi, fRsrc, oi, 0, 0);
        const float fjx = reduceOver8Lanes(-f_ij.x), fjy = reduceOver8Lanes(-f_ij.y), fjz = reduceOver8Lanes(-f_ij.z);
        const float vj  = (tidxi == 0U) ? fjx : ((tidxi == 1U) ? fjy : fjz);
        const int   oj  = (tidxi < 3U) ? (3 * aj + static_cast<int>(tidxi)) * static_cast<int>(sizeof(float)) : 0x7FFFFFF0;
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(vj, fRsrc, oj, 0, 0);
    }
    if (bCalcFshift && shiftIdx != c_centralShiftIndex)
    {
        const float sx = waveSum(f_ij.x);
        const float sy = waveSum(f_ij.y);
        const float sz = waveSum(f_ij.z);
        if (lane < 3U)
        {
            const float v = (lane == 0U) ? sx : ((lane == 1U) ? sy : sz);
            atomicAdd(reinterpret_cast<float*>(atdat.fShift) + 3 * shiftIdx + static_cast<int>(lane), v);
        }
    }
    if constexpr (ENERGY)
    {
        const float v = waveSum4Transposed(E_lj, E_el, DVDL_lj, DVDL_el, lane); /* lanes 0 .. 3: the four sums */
        if (lane < 4U && v != 0.0F)
        {
            atomicAdd(atdat.energySlots + (item & (c_numEnergySlots - 1)) * c_energySlotStride + static_cast<int>(lane), v);
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
