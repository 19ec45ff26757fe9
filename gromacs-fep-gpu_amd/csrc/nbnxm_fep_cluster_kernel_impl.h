/*
 * Perturbed cluster pairs of the fused mode: the (i-cluster, j-cluster) pairs of the cluster-pair list that touch a
 * perturbed atom (Grid::fepBits).  The cluster kernel masks them out of its list words (gpu_plist::groupSlowMask);
 * this kernel evaluates them — soft-core A/B pair for the perturbed lanes, the plain pair for the others — together
 * with everything else the reference does on the atom-pair list: energies, dV/dlambda, shift forces, the
 * lambda-dependent self terms and, on dH/dl steps, the energies at the foreign lambdas.
 *
 * Replaces nbnxn_fep_kernel_*_{F,VF}_cuda and nbnxn_foreign_fep_kernel_*_V_cuda (nbnxm/cuda/nbnxm_fep_cuda_kernel.cuh:87-628,
 * nbnxm_foreign_fep_cuda_kernel.cuh:88-583) in fused mode, without an atom-pair list: no make_fep_list on the host, no
 * gpu_init_feppairlist.  Pair semantics: fepPair (nbnxm_device_helpers.h) = the CPU kernel nb_free_energy_kernel<>.
 *
 * One wavefront per perturbed cluster pair (gpu_plist::slowPairs, a few thousand for a ligand-sized perturbed region, built
 * on the device once per list), lane = tidxj*8 + tidxi as in the cluster kernel: every lane owns one atom pair, nothing is
 * staged.  The waves are short latency-bound chains.  On force-only steps the body (nbnxm_fep_cluster_body.h) runs in trailing
 * workgroups of the cluster kernel instead; this kernel serves the energy and dH/dl steps, queued ahead of the cluster kernel.
 */
#ifndef NBNXM_FEP_CLUSTER_KERNEL_IMPL_H
#define NBNXM_FEP_CLUSTER_KERNEL_IMPL_H

#include "nbnxm_kernel_impl.h" /* includes nbnxm_fep_cluster_body.h */

constexpr int c_fepClusterWavesPerBlock = c_fepClusterWavesPerBlockDef;

template<int ELEC, bool TWIN, int VDW, bool ENERGY, bool FOREIGN>
__launch_bounds__(c_fepClusterWavesPerBlock* c_waveSize) __global__
        void nbnxmFepClusterKernel(const NBAtomDataGpu atdat,
                                   const NBParamGpu    nbp,
                                   const gpu_plist     plist,
                                   const int           bCalcFshift,
                                   const nbnxn_sci_t* __restrict__ sciList, /* plist.sciSorted */
                                   const nbnxn_cj_packed_t* __restrict__ cjPackedList,
                                   const nbnxn_excl_t* __restrict__ exclList,
                                   const float4* __restrict__ xq,
                                   const int* __restrict__ atomTypes,
                                   const float2* __restrict__ ljComb,
                                   const unsigned* __restrict__ fepWords,
                                   const int numForeignLambda /* FOREIGN: lambda indices 0 .. numForeignLambda */)
{
    constexpr bool USE_TABLE = VdwTraits<VDW>::useTable;
    const unsigned wave      = __builtin_amdgcn_readfirstlane(threadIdx.x / c_waveSize);

    /* LDS: the LJ parameter table of the workgroup */
    extern __shared__ __align__(16) unsigned char fepLds[];
    const int numTypes = atdat.numTypes;
    float2*   nbfpLds  = reinterpret_cast<float2*>(fepLds);
    if constexpr (USE_TABLE)
    {
        for (int t = threadIdx.x; t < numTypes * numTypes; t += blockDim.x) { nbfpLds[t] = nbp.nbfp[t]; }
        __syncthreads();
    }

    /* FOREIGN: a scratch area per wave behind the table (fepClusterPair; the host sizes the launch's LDS: fepClusterLdsBytes) */
    float* waveLds = nullptr;
    if constexpr (FOREIGN)
    {
        const int tabBytes = USE_TABLE ? ((numTypes * numTypes * static_cast<int>(sizeof(float2)) + 15) & ~15) : 0;
        waveLds            = reinterpret_cast<float*>(fepLds + tabBytes + wave * c_fepForeignLdsBytes);
    }
    /* few, short, latency-bound waves next to (or between) kernels with thousands: top priority */
    __builtin_amdgcn_s_setprio(3);
    /* the launch is sized by the host's figure (the previous list's while a new list's count is still on its way), the items are
     * counted on the device: the waves stride over them */
    const int firstItem = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * c_fepClusterWavesPerBlock + static_cast<int>(wave));
    const int numItems  = __builtin_amdgcn_readfirstlane(min(*plist.slowCount, plist.slowPairs_nalloc));
    const int stride    = static_cast<int>(gridDim.x) * c_fepClusterWavesPerBlock;
    if constexpr (FOREIGN)
    {
        /* the heavy cluster pairs at the front of the list: c_fepForeignHeavyChunks waves each, by lambda index (see the cluster kernel's tail) */
        const int numHeavy   = __builtin_amdgcn_readfirstlane(min(plist.slowCount[1], numItems));
        const int numVirtual = numItems + numHeavy * (c_fepForeignHeavyChunks - 1);
        for (int v = firstItem; v < numVirtual; v += stride)
        {
            const bool heavy = v < numHeavy * c_fepForeignHeavyChunks;
            const int  item  = heavy ? v / c_fepForeignHeavyChunks : v - numHeavy * (c_fepForeignHeavyChunks - 1);
            const int  chunk = heavy ? v - item * c_fepForeignHeavyChunks : -1;
            fepClusterPair<ELEC, TWIN, VDW, ENERGY, FOREIGN>(atdat, nbp, plist, bCalcFshift, cjPackedList, exclList, xq, ljComb, fepWords,
                                                             numForeignLambda, item, nbfpLds, waveLds, chunk);
        }
    }
    else
    {
        for (int item = firstItem; item < numItems; item += stride)
        {
            fepClusterPair<ELEC, TWIN, VDW, ENERGY, FOREIGN>(atdat, nbp, plist, bCalcFshift, cjPackedList, exclList, xq, ljComb, fepWords,
                                                             numForeignLambda, item, nbfpLds, waveLds);
        }
    }
}

#endif
