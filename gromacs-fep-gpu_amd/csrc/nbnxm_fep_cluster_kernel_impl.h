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
 * One wavefront per packed j-group that holds such pairs (gpu_plist::slowGroups, a few hundred for a ligand-sized
 * perturbed region), lane = tidxj*8 + tidxi as in the cluster kernel.  The 64 i-atoms' A/B data are staged in the wave's
 * LDS because the i-cluster is a run-time index here.  Launched on the locality's FEP stream, concurrently with the
 * cluster kernel: it is a few hundred latency-bound waves that fill issue slots the big kernel leaves.
 */
#ifndef NBNXM_FEP_CLUSTER_KERNEL_IMPL_H
#define NBNXM_FEP_CLUSTER_KERNEL_IMPL_H

#include "nbnxm_kernel_impl.h"

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
    static_assert(!FOREIGN || ENERGY, "the foreign-lambda flavour is an energy flavour");
    constexpr bool EXCL_FORCES = (ELEC != ELK_CUT) || ENERGY;
    constexpr bool USE_TABLE   = VdwTraits<VDW>::useTable;
    constexpr int  FEP_ELEC    = (ELEC == ELK_CUT) ? ELK_RF : ELEC;

    const unsigned lane  = threadIdx.x & (c_waveSize - 1);
    const unsigned wave  = __builtin_amdgcn_readfirstlane(threadIdx.x / c_waveSize);
    const unsigned tidxi = lane & 7U;
    const unsigned tidxj = lane >> 3;
    const unsigned half  = lane >> 5;

    /* LDS: the LJ parameter table of the workgroup, then per wave the 64 i-atoms { x,q (shifted) ; qA,qB ; typeA,typeB } */
    extern __shared__ __align__(16) unsigned char fepLds[];
    const int numTypes   = atdat.numTypes;
    float2*   nbfpLds    = reinterpret_cast<float2*>(fepLds);
    const int tableBytes = USE_TABLE ? ((numTypes * numTypes * static_cast<int>(sizeof(float2)) + 15) & ~15) : 0;
    float4*   xqib       = reinterpret_cast<float4*>(fepLds + tableBytes + wave * c_iStageBytes);
    float2*   qABib      = reinterpret_cast<float2*>(xqib + c_superClSize);
    int2*     tABib      = reinterpret_cast<int2*>(qABib + c_superClSize);
    if constexpr (USE_TABLE)
    {
        for (int t = threadIdx.x; t < numTypes * numTypes; t += blockDim.x) { nbfpLds[t] = nbp.nbfp[t]; }
    }
    __syncthreads();

    /* few, long-latency waves next to the cluster kernel's thousands (whose waves start at priority 3 and lower it as
     * they advance): top priority, or they would only get issue slots once that kernel is nearly done */
    __builtin_amdgcn_s_setprio(3);
    const int item = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * c_fepClusterWavesPerBlock + static_cast<int>(wave));
    /* one wave per (group, j-cluster slot): the waves are latency-bound chains, so the kernel lasts as long as its
     * longest wave; four short ones per group instead of one long one */
    if (item >= plist.numSlowGroups * c_jGroupSize) { return; }
    const int         mySlot   = item & (c_jGroupSize - 1);
    const int         group    = __builtin_amdgcn_readfirstlane(plist.slowGroups[item >> 2]);
    const nbnxn_sci_t nb_sci   = sciList[__builtin_amdgcn_readfirstlane(plist.slowGroupSci[item >> 2])];
    const int         sci      = nb_sci.sci;
    const int         shiftIdx = nb_sci.shift & NBNXM_CI_SHIFT_MASK;
    const bool        central  = (shiftIdx == c_centralShiftIndex);

    const nbnxn_cj_packed_t* __restrict__ grp = &cjPackedList[group];
    const unsigned imask = grp->imei[0].imask & __builtin_amdgcn_readfirstlane(plist.groupSlowMask[group])
                           & (0xFFU << (mySlot * c_numClPerSupercl));

    unsigned long long iFepBits = static_cast<unsigned long long>(fepWords[2 * sci]) | (static_cast<unsigned long long>(fepWords[2 * sci + 1]) << 32);
    /* the atoms' own (i == j) terms belong to the group that holds the i-entry's own clusters */
    const bool diagGroup = ENERGY && EXCL_FORCES && central && mySlot == 0 && grp->cj[0] == sci * c_numClPerSupercl && iFepBits != 0ULL;
    if (imask == 0U && !diagGroup) { return; }

    float* __restrict__ f   = reinterpret_cast<float*>(atdat.f);
    const float rcoulomb_sq = nbp.rcoulomb_sq;
    const __amdgpu_buffer_rsrc_t fRsrc =
            __builtin_amdgcn_make_buffer_rsrc(f, 0, atdat.numAtoms * 3 * static_cast<int>(sizeof(float)), 0x00020000);
    const float2* __restrict__ nbfp = nbp.nbfp;

    {
        const float3 sh = atdat.shiftVec[shiftIdx];
        const int    ai = sci * c_superClSize + static_cast<int>(lane);
        float4       xl = xq[ai];
        xl.x += sh.x;
        xl.y += sh.y;
        xl.z += sh.z;
        xl.w *= nbp.epsfac;
        xqib[lane]      = xl;
        const float4 q4 = atdat.q4[ai];
        qABib[lane]     = make_float2(q4.x * nbp.epsfac, q4.y * nbp.epsfac);
        const int4 t4   = atdat.atomTypes4[ai];
        tABib[lane]     = make_int2(t4.x, t4.y);
        /* the staged copy is private to this wave: LDS operations of one wave complete in order */
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }

    float  E_lj = 0.0F, E_el = 0.0F, DVDL_lj = 0.0F, DVDL_el = 0.0F;
    float3 fshiftAcc = make_float3(0.0F, 0.0F, 0.0F);
    [[maybe_unused]] const float selfCoef = (ELEC == ELK_CUT || ELEC == ELK_RF) ? -0.5F * nbp.c_rf : -nbp.ewald_beta * c_oneOverSqrtPi;

    if constexpr (ENERGY && EXCL_FORCES)
    {
        /* perturbed atoms carry q = 0 in xq; their lambda-dependent self term is what the i == j entry of the
         * atom-pair list contributes (nb_free_energy.cpp:1035-1052,1079-1100) */
        if (diagGroup && ((iFepBits >> lane) & 1ULL))
        {
            const float2 qAB = qABib[lane];
            const float  sA  = qAB.x * qAB.x / nbp.epsfac * selfCoef;
            const float  sB  = qAB.y * qAB.y / nbp.epsfac * selfCoef;
            E_el += (1.0F - nbp.lambda_q) * sA + nbp.lambda_q * sB;
            DVDL_el += sB - sA;
        }
    }

    const unsigned wexcl = exclList[half ? grp->imei[1].excl_ind : grp->imei[0].excl_ind].pair[lane & 31U];
    const FepLambda L    = makeFepLambda(nbp.lambda_q, nbp.lambda_v, nbp.lam_power, nbp.alpha_coul, nbp.alpha_vdw);

#pragma unroll 1
    for (int jm = 0; jm < c_jGroupSize; jm++)
    {
        const unsigned slowMask = (imask >> (jm * c_numClPerSupercl)) & 0xFFU;
        if (slowMask == 0U) { continue; }
        const int      cj       = grp->cj[jm];
        const unsigned jFepBits = (fepWords[cj >> 2] >> ((cj & 3) * 8)) & 0xFFU;
        const unsigned wexclJ   = wexcl >> (jm * c_numClPerSupercl);
        const int      aj       = cj * c_clSize + static_cast<int>(tidxj);
        const float4   xqj      = xq[aj];
        const float4   q4j      = atdat.q4[aj];
        const int4     t4j      = atdat.atomTypes4[aj];
        int            typej    = 0;
        float2         ljcp_j   = make_float2(0.0F, 0.0F);
        if constexpr (USE_TABLE) { typej = atomTypes[aj]; }
        else { ljcp_j = ljComb[aj]; }
        float3 fcj_buf = make_float3(0.0F, 0.0F, 0.0F);
#pragma unroll 1
        for (int i = 0; i < c_numClPerSupercl; i++)
        {
            if (!(slowMask & (1U << i))) { continue; }
            const unsigned iBits    = static_cast<unsigned>(iFepBits >> (i * c_clSize)) & 0xFFU;
            const int      ci       = sci * c_numClPerSupercl + i;
            const int      ai       = ci * c_clSize + static_cast<int>(tidxi);
            const float4   xi       = xqib[i * c_clSize + tidxi];
            const int2     tABi     = tABib[i * c_clSize + tidxi];
            const float3   rv       = make_float3(xi.x - xqj.x, xi.y - xqj.y, xi.z - xqj.z);
            const float    r2       = rv.x * rv.x + rv.y * rv.y + rv.z * rv.z;
            const bool     included = ((wexclJ >> i) & 1U) != 0U;
            const bool     subDiag  = central && (ci == cj) && (tidxj <= tidxi);
            const bool     pert     = (((iBits >> tidxi) | (jFepBits >> tidxj)) & 1U) != 0U;
            float          F_invr   = 0.0F;
            if (pert)
            {
                if (!subDiag)
                {
                    const float2 qABi   = qABib[i * c_clSize + tidxi];
                    const float  qq[2]  = { qABi.x * q4j.x, qABi.y * q4j.y };
                    const float2 pA     = USE_TABLE ? nbfpLds[numTypes * tABi.x + t4j.x] : nbfp[numTypes * tABi.x + t4j.x];
                    const float2 pB     = USE_TABLE ? nbfpLds[numTypes * tABi.y + t4j.y] : nbfp[numTypes * tABi.y + t4j.y];
                    const float  c6[2]  = { pA.x, pB.x };
                    const float  c12[2] = { pA.y, pB.y };
                    float        fscal  = 0.0F;
                    const bool   done   = fepPair<FEP_ELEC, VDW == VDK_PSWITCH, true, ENERGY>(nbp, L, r2, included, false, qq, c6, c12, fscal, E_lj,
                                                                                              E_el, DVDL_lj, DVDL_el);
                    F_invr = done ? fscal : 0.0F;
                }
            }
            else
            {
                /* a plain pair inside a perturbed cluster pair */
                const int intMask = included ? -1 : 0;
                bool      active;
                if constexpr (EXCL_FORCES) { active = (r2 < rcoulomb_sq) && !subDiag; }
                else { active = (r2 < rcoulomb_sq) && included; }
                if (active)
                {
                    float c6, c12;
                    if constexpr (USE_TABLE)
                    {
                        /* a non-perturbed atom's type is its A-state type */
                        const float2 c6c12 = nbfpLds[numTypes * tABi.x + typej];
                        c6                 = c6c12.x;
                        c12                = c6c12.y;
                    }
                    else { ljFromComb(VDW, ljComb[ai], ljcp_j, c6, c12); }
                    float E_lj_p = 0.0F, E_el_p = 0.0F;
                    /* no Ewald table in this kernel's LDS: the rational form of the correction */
                    nbPair<ELEC, TWIN, VDW, ENERGY, EXCL_FORCES, true, false>(nbp, nullptr, r2, intMask, xi.w * xqj.w, c6, c12, F_invr,
                                                                                     E_lj_p, E_el_p);
                    if constexpr (ENERGY)
                    {
                        E_lj += E_lj_p;
                        E_el += E_el_p;
                    }
                }
            }
            const float3 f_ij = make_float3(rv.x * F_invr, rv.y * F_invr, rv.z * F_invr);
            fcj_buf.x -= f_ij.x;
            fcj_buf.y -= f_ij.y;
            fcj_buf.z -= f_ij.z;
            const float fix = reduceOverTidxj(f_ij.x);
            const float fiy = reduceOverTidxj(f_ij.y);
            const float fiz = reduceOverTidxj(f_ij.z);
            {
                const float v   = (tidxj == 0U) ? fix : ((tidxj == 1U) ? fiy : fiz);
                const int   off = (tidxj < 3U) ? (3 * ai + static_cast<int>(tidxj)) * static_cast<int>(sizeof(float)) : c_dropLane;
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, fRsrc, off, 0, 0);
            }
            fshiftAcc.x += f_ij.x;
            fshiftAcc.y += f_ij.y;
            fshiftAcc.z += f_ij.z;
        }
        const float fjx = reduceOver8Lanes(fcj_buf.x);
        const float fjy = reduceOver8Lanes(fcj_buf.y);
        const float fjz = reduceOver8Lanes(fcj_buf.z);
        {
            const float v   = (tidxi == 0U) ? fjx : ((tidxi == 1U) ? fjy : fjz);
            const int   off = (tidxi < 3U) ? (3 * aj + static_cast<int>(tidxi)) * static_cast<int>(sizeof(float)) : c_dropLane;
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, fRsrc, off, 0, 0);
        }
    }

    if (bCalcFshift && !central)
    {
        const float sx = waveSum(fshiftAcc.x);
        const float sy = waveSum(fshiftAcc.y);
        const float sz = waveSum(fshiftAcc.z);
        if (lane < 3U)
        {
            const float v = (lane == 0U) ? sx : ((lane == 1U) ? sy : sz);
            atomicAdd(reinterpret_cast<float*>(atdat.fShift) + 3 * shiftIdx + static_cast<int>(lane), v);
        }
    }

    if constexpr (ENERGY)
    {
        E_lj    = waveSum(E_lj);
        E_el    = waveSum(E_el);
        DVDL_lj = waveSum(DVDL_lj);
        DVDL_el = waveSum(DVDL_el);
        const int   slot = item & (c_numEnergySlots - 1);
        const float v    = (lane == 0U) ? E_lj : ((lane == 1U) ? E_el : ((lane == 2U) ? DVDL_lj : DVDL_el));
        if (lane < 4U) { atomicAdd(atdat.energySlots + slot * c_energySlotStride + static_cast<int>(lane), v); }
    }

    /* ---- foreign lambdas (dH/dl steps): the same pairs' energies at every lambda index, c_foreignChunk indices per walk */
    if constexpr (FOREIGN)
    {
        constexpr int c_foreignChunk = 4;
        for (int fbase = 0; fbase <= numForeignLambda; fbase += c_foreignChunk)
        {
            FepLambda Lf[c_foreignChunk];
            float     fE_lj[c_foreignChunk], fE_el[c_foreignChunk], fDVDL_lj[c_foreignChunk], fDVDL_el[c_foreignChunk];
#pragma unroll
            for (int q = 0; q < c_foreignChunk; q++)
            {
                const int   fidx = min(fbase + q, numForeignLambda);
                const float lc   = (fidx == 0) ? nbp.lambda_q : nbp.allLambdaCoul[fidx - 1];
                const float lv   = (fidx == 0) ? nbp.lambda_v : nbp.allLambdaVdw[fidx - 1];
                Lf[q]            = makeFepLambda(lc, lv, nbp.lam_power, nbp.alpha_coul, nbp.alpha_vdw);
                fE_lj[q] = fE_el[q] = fDVDL_lj[q] = fDVDL_el[q] = 0.0F;
                if (diagGroup && ((iFepBits >> lane) & 1ULL))
                {
                    const float2 qAB = qABib[lane];
                    const float  sA  = qAB.x * qAB.x / nbp.epsfac * selfCoef;
                    const float  sB  = qAB.y * qAB.y / nbp.epsfac * selfCoef;
                    fE_el[q] += (1.0F - lc) * sA + lc * sB;
                    fDVDL_el[q] += sB - sA;
                }
            }
#pragma unroll 1
            for (int jm = 0; jm < c_jGroupSize; jm++)
            {
                const unsigned slowMask = (imask >> (jm * c_numClPerSupercl)) & 0xFFU;
                if (slowMask == 0U) { continue; }
                const int      cj       = grp->cj[jm];
                const unsigned jFepBits = (fepWords[cj >> 2] >> ((cj & 3) * 8)) & 0xFFU;
                const unsigned wexclJ   = wexcl >> (jm * c_numClPerSupercl);
                const int      aj       = cj * c_clSize + static_cast<int>(tidxj);
                const float4   xqj      = xq[aj];
                const float4   q4j      = atdat.q4[aj];
                const int4     t4j      = atdat.atomTypes4[aj];
#pragma unroll 1
                for (int i = 0; i < c_numClPerSupercl; i++)
                {
                    if (!(slowMask & (1U << i))) { continue; }
                    const unsigned iBits   = static_cast<unsigned>(iFepBits >> (i * c_clSize)) & 0xFFU;
                    const int      ci      = sci * c_numClPerSupercl + i;
                    const bool     subDiag = central && (ci == cj) && (tidxj <= tidxi);
                    const bool     pert    = (((iBits >> tidxi) | (jFepBits >> tidxj)) & 1U) != 0U;
                    if (pert && !subDiag)
                    {
                        const float4 xi     = xqib[i * c_clSize + tidxi];
                        const int2   tABi   = tABib[i * c_clSize + tidxi];
                        const float3 rv     = make_float3(xi.x - xqj.x, xi.y - xqj.y, xi.z - xqj.z);
                        const float  r2     = rv.x * rv.x + rv.y * rv.y + rv.z * rv.z;
                        const float2 qABi   = qABib[i * c_clSize + tidxi];
                        const float  qq[2]  = { qABi.x * q4j.x, qABi.y * q4j.y };
                        const float2 pA     = USE_TABLE ? nbfpLds[numTypes * tABi.x + t4j.x] : nbfp[numTypes * tABi.x + t4j.x];
                        const float2 pB     = USE_TABLE ? nbfpLds[numTypes * tABi.y + t4j.y] : nbfp[numTypes * tABi.y + t4j.y];
                        const float  c6[2]  = { pA.x, pB.x };
                        const float  c12[2] = { pA.y, pB.y };
                        float        fscal  = 0.0F;
#pragma unroll
                        for (int q = 0; q < c_foreignChunk; q++)
                        {
                            fepPair<FEP_ELEC, VDW == VDK_PSWITCH, false, true>(nbp, Lf[q], r2, ((wexclJ >> i) & 1U) != 0U, false, qq, c6, c12, fscal,
                                                                               fE_lj[q], fE_el[q], fDVDL_lj[q], fDVDL_el[q]);
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < c_foreignChunk; q++)
            {
                const float s0 = waveSum(fE_lj[q]);
                const float s1 = waveSum(fE_el[q]);
                const float s2 = waveSum(fDVDL_lj[q]);
                const float s3 = waveSum(fDVDL_el[q]);
                if (lane < 4U && fbase + q <= numForeignLambda)
                {
                    const float v   = (lane == 0U) ? s0 : ((lane == 1U) ? s1 : ((lane == 2U) ? s2 : s3));
                    float*      out = (lane == 0U) ? atdat.eLJForeign
                                                   : ((lane == 1U) ? atdat.eElecForeign : ((lane == 2U) ? atdat.dvdlLJForeign : atdat.dvdlElecForeign));
                    if (v != 0.0F) { atomicAdd(out + fbase + q, v); }
                }
            }
        }
    }
}

#endif
