/*
 * One perturbed cluster pair of the fused mode, evaluated by one wavefront: the body of nbnxmFepClusterKernel, also called by the
 * trailing workgroups of the force-only cluster kernel (nbnxm_kernel_impl.h), which fill the wave slots its own waves free
 * towards the end.  See nbnxm_fep_cluster_kernel_impl.h for what it replaces.
 */
#ifndef NBNXM_FEP_CLUSTER_BODY_H
#define NBNXM_FEP_CLUSTER_BODY_H

/* item: index into gpu_plist::slowPairs (< numSlowPairs); nbfpLds: the LJ parameter table in LDS (table flavours) */
template<int ELEC, bool TWIN, int VDW, bool ENERGY, bool FOREIGN>
NB_DEVINL void fepClusterPair(const NBAtomDataGpu& atdat,
                              const NBParamGpu&    nbp,
                              const gpu_plist&     plist,
                              const int            bCalcFshift,
                              const nbnxn_cj_packed_t* __restrict__ cjPackedList,
                              const nbnxn_excl_t* __restrict__ exclList,
                              const float4* __restrict__ xq,
                              const float2* __restrict__ ljComb,
                              const unsigned* __restrict__ fepWords,
                              const int     numForeignLambda, /* FOREIGN: lambda indices 0 .. numForeignLambda */
                              const int     item,
                              const float2* nbfpLds,
                              /* FOREIGN: c_fepForeignLdsBytes of LDS of this wave's own (the cluster kernel's trailing workgroups: the wave's
                               * staging buffers; nbnxmFepClusterKernel: behind the LJ table), or nullptr: the foreign lambdas one after the other */
                              float*        waveLds = nullptr,
                              /* FOREIGN: -1: this wave does all of the cluster pair; c in [0, c_fepForeignHeavyChunks): one of the waves a HEAVY
                               * cluster pair is split over (the front of gpu_plist::slowPairs) — lambda indices 1 + c, 1 + c + chunks, ...;
                               * wave 0 also does what is not a foreign-lambda term: forces, energies, index 0 */
                              const int     foreignChunk = -1)
{
    static_assert(!FOREIGN || ENERGY, "the foreign-lambda flavour is an energy flavour");
    constexpr bool LJ_EWALD    = VdwTraits<VDW>::ljEwald;
    constexpr bool EXCL_FORCES = (ELEC != ELK_CUT) || ENERGY || LJ_EWALD;
    constexpr bool USE_TABLE   = VdwTraits<VDW>::useTable;
    constexpr int  FEP_ELEC    = (ELEC == ELK_CUT) ? ELK_RF : ELEC;

    const unsigned lane     = threadIdx.x & (c_waveSize - 1);
    const unsigned tidxi    = lane & 7U;
    const unsigned tidxj    = lane >> 3;
    const unsigned half     = lane >> 5;
    const int      numTypes = atdat.numTypes;

    /* one wave per perturbed cluster pair: the waves are latency-bound chains, so the kernel lasts as long as its
     * longest wave, and one cluster pair is the shortest unit there is */
    /* the pair's record: one scalar load (the item is wave-uniform), and everything below depends on nothing but it — the list word that
     * says whether the pair survives the current pruning travels beside the atoms' data, it is looked at when they are on their way */
    typedef int nb_int8 __attribute__((ext_vector_type(8)));
    const nb_int8     rec      = *reinterpret_cast<const nb_int8*>(&plist.slowPairs[__builtin_amdgcn_readfirstlane(item)]);
    const int         entry    = __builtin_amdgcn_readfirstlane(rec[0]); /* group * 32 + jm * 8 + i */
    const int         group    = entry >> 5;
    const int         jm       = (entry >> 3) & 3;
    const int         i        = entry & 7;
    const int         sciShift = __builtin_amdgcn_readfirstlane(rec[1]); /* sci * 64 + shift index */
    const int         sci      = sciShift >> 6;
    const int         shiftIdx = sciShift & 63;
    const bool        central  = (shiftIdx == c_centralShiftIndex);
    const int         cj       = __builtin_amdgcn_readfirstlane(rec[2]);
    const int         exclIndOfHalf[2] = { __builtin_amdgcn_readfirstlane(rec[3]), __builtin_amdgcn_readfirstlane(rec[4]) };

    const nbnxn_cj_packed_t* __restrict__ grp = &cjPackedList[group];
    const int      ci        = sci * c_numClPerSupercl + i;
    const unsigned imaskNow  = grp->imei[0].imask;
    const bool     diagPair  = central && (ci == cj);

    const unsigned iBits    = (fepWords[ci >> 2] >> ((ci & 3) * 8)) & 0xFFU;
    const unsigned jFepBits = (fepWords[cj >> 2] >> ((cj & 3) * 8)) & 0xFFU;

    float* __restrict__ f = reinterpret_cast<float*>(atdat.f);
    const __amdgpu_buffer_rsrc_t fRsrc =
            __builtin_amdgcn_make_buffer_rsrc(f, 0, atdat.numAtoms * 3 * static_cast<int>(sizeof(float)), 0x00020000);
    const float2* __restrict__ nbfp = nbp.nbfp;

    /* lane (tidxj, tidxi) owns the pair (i-atom tidxi, j-atom tidxj) */
    const int    ai  = ci * c_clSize + static_cast<int>(tidxi);
    const int    aj  = cj * c_clSize + static_cast<int>(tidxj);
    const float3 sh  = atdat.shiftVec[shiftIdx];
    float4       xi  = xq[ai];
    const float4 q4i = atdat.q4[ai];
    const int4   t4i = atdat.atomTypes4[ai];
    const float4 xqj = xq[aj];
    const float4 q4j = atdat.q4[aj];
    const int4   t4j = atdat.atomTypes4[aj];
    xi.x += sh.x;
    xi.y += sh.y;
    xi.z += sh.z;
    const float2   qABi     = make_float2(q4i.x * nbp.epsfac, q4i.y * nbp.epsfac);
    const unsigned wexcl    = exclList[half ? exclIndOfHalf[1] : exclIndOfHalf[0]].pair[lane & 31U];
    const bool     inList   = ((imaskNow >> (jm * c_numClPerSupercl + i)) & 1U) != 0U; /* survives the current pruning */
    /* the atoms' own (i == j) terms belong to the cluster's pair with itself, whether or not that pair is in range */
    if (!inList && !(ENERGY && EXCL_FORCES && diagPair)) { return; }
    const bool     included = ((wexcl >> (jm * c_numClPerSupercl + i)) & 1U) != 0U;
    const float3   rv       = make_float3(xi.x - xqj.x, xi.y - xqj.y, xi.z - xqj.z);
    const float    r2       = rv.x * rv.x + rv.y * rv.y + rv.z * rv.z;
    const bool     subDiag  = diagPair && (tidxj <= tidxi);
    const bool     pert     = (((iBits >> tidxi) | (jFepBits >> tidxj)) & 1U) != 0U;
    const float    qq[2]    = { qABi.x * q4j.x, qABi.y * q4j.y };
    const float2   pA       = USE_TABLE ? nbfpLds[numTypes * t4i.x + t4j.x] : nbfp[numTypes * t4i.x + t4j.x];
    const float2   pB       = USE_TABLE ? nbfpLds[numTypes * t4i.y + t4j.y] : nbfp[numTypes * t4i.y + t4j.y];
    const float    c6AB[2]  = { pA.x, pB.x };
    const float    c12AB[2] = { pA.y, pB.y };
    float          c6gridAB[2];
    ljGridC6AB(nbp, t4i, t4j, c6gridAB); /* zero without LJ-PME */
    /* LJ-PME: what the grid counts for an atom's pair with itself, removed like the Coulomb self term (:1103-1136 with i == j) */
    [[maybe_unused]] const float ljSelfCoef = LJ_EWALD ? 0.5F * c_oneSixth * c_oneSixth * nbp.ewaldcoeff_lj * nbp.ewaldcoeff_lj * nbp.ewaldcoeff_lj
                                                                * nbp.ewaldcoeff_lj * nbp.ewaldcoeff_lj * nbp.ewaldcoeff_lj
                                                       : 0.0F;
    [[maybe_unused]] const float selfCoef = (ELEC == ELK_CUT || ELEC == ELK_RF) ? -0.5F * nbp.c_rf : -nbp.ewald_beta * c_oneOverSqrtPi;
    /* the self term of i-atom tidxi: the lanes tidxj == tidxi of the cluster's pair with itself */
    [[maybe_unused]] const bool selfLane = ENERGY && EXCL_FORCES && diagPair && (tidxj == tidxi) && ((iBits >> tidxi) & 1U);

    /* this pair's lambdas: the object's, or its window's when several windows are batched into the object */
    float  lambdaQ = nbp.lambda_q, lambdaV = nbp.lambda_v;
    float* energySlots  = atdat.energySlots;
    float* foreignSlots = atdat.foreignSlots;
    if (nbp.clustersPerWindow > 0)
    {
        const int    window = ci / nbp.clustersPerWindow;
        const float2 wl     = nbp.windowLambda[window];
        lambdaQ             = wl.x;
        lambdaV             = wl.y;
        energySlots         = atdat.windowSlots + window * atdat.windowSlotStride;
        foreignSlots        = energySlots + atdat.windowForeignOffset;
    }

    const bool primary = !FOREIGN || foreignChunk <= 0;
    float E_lj = 0.0F, E_el = 0.0F, DVDL_lj = 0.0F, DVDL_el = 0.0F;
    float F_invr = 0.0F;
    /* FOREIGN: this lane's own terms at the current lambda — the self term and the perturbed pair —, which are lambda index 0 of the
     * foreign-lambda sums (E_lj .. DVDL_el also collect the plain pairs of the cluster pair, which do not belong there) */
    [[maybe_unused]] float fep0[4] = { 0.0F, 0.0F, 0.0F, 0.0F };
    if constexpr (ENERGY && EXCL_FORCES)
    {
        /* perturbed atoms carry q = 0 in xq; their lambda-dependent self term is what the i == j entry of the
         * atom-pair list contributes (nb_free_energy.cpp:1035-1052,1079-1100) */
        if (primary && selfLane)
        {
            const float sA = qABi.x * qABi.x / nbp.epsfac * selfCoef;
            const float sB = qABi.y * qABi.y / nbp.epsfac * selfCoef;
            E_el += (1.0F - lambdaQ) * sA + lambdaQ * sB;
            DVDL_el += sB - sA;
            if constexpr (LJ_EWALD)
            {
                /* (this lane is the atom's pair with itself: c6gridAB are its own grid C6 in the two states) */
                E_lj += ((1.0F - lambdaV) * c6gridAB[0] + lambdaV * c6gridAB[1]) * ljSelfCoef;
                DVDL_lj += (c6gridAB[1] - c6gridAB[0]) * ljSelfCoef;
            }
        }
    }
    if (primary && inList && pert && !subDiag)
    {
        const FepLambda L     = makeFepLambda(lambdaQ, lambdaV, nbp.lam_power, nbp.alpha_coul, nbp.alpha_vdw);
        float           fscal = 0.0F;
        const bool      done  = fepPair<FEP_ELEC, VDW == VDK_PSWITCH, true, ENERGY>(nbp, L, r2, included, false, qq, c6AB, c12AB, fscal, E_lj, E_el,
                                                                                   DVDL_lj, DVDL_el, c6gridAB);
        F_invr = done ? fscal : 0.0F;
    }
    if constexpr (FOREIGN)
    {
        fep0[0] = E_lj;
        fep0[1] = E_el;
        fep0[2] = DVDL_lj;
        fep0[3] = DVDL_el;
    }
    if (primary && inList && !pert)
    {
        {
            /* a plain pair inside a perturbed cluster pair */
            const int intMask = included ? -1 : 0;
            bool      active;
            if constexpr (EXCL_FORCES) { active = (r2 < nbp.rcoulomb_sq) && !subDiag; }
            else { active = (r2 < nbp.rcoulomb_sq) && included; }
            if (active)
            {
                float c6, c12;
                if constexpr (USE_TABLE)
                {
                    /* non-perturbed atoms: the A-state type is the type */
                    c6  = pA.x;
                    c12 = pA.y;
                }
                else { ljFromComb(VDW, ljComb[ai], ljComb[aj], c6, c12); }
                float E_lj_p = 0.0F, E_el_p = 0.0F, c6grid = 0.0F;
                if constexpr (LJ_EWALD) { c6grid = ljGridC6(VDW, nbp.nbfp_comb[t4i.x], nbp.nbfp_comb[t4j.x]); }
                /* no Ewald table in this kernel's LDS: the rational form of the correction */
                nbPair<ELEC, TWIN, VDW, ENERGY, EXCL_FORCES, true, false>(nbp, nullptr, r2, intMask, xi.w * nbp.epsfac * xqj.w, c6, c12, F_invr,
                                                                         E_lj_p, E_el_p, c6grid);
                if constexpr (ENERGY)
                {
                    E_lj += E_lj_p;
                    E_el += E_el_p;
                }
            }
        }
    }
    const float3 f_ij = make_float3(rv.x * F_invr, rv.y * F_invr, rv.z * F_invr);
    if (primary)
    {
        /* i-forces: sum over tidxj, lanes tidxj 0..2 carry x, y, z; j-forces: sum over tidxi, lanes tidxi 0..2 */
        const float fix = reduceOverTidxj(f_ij.x), fiy = reduceOverTidxj(f_ij.y), fiz = reduceOverTidxj(f_ij.z);
        const float vi  = (tidxj == 0U) ? fix : ((tidxj == 1U) ? fiy : fiz);
        const int   oi  = (tidxj < 3U) ? (3 * ai + static_cast<int>(tidxj)) * static_cast<int>(sizeof(float)) : c_dropLane;
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(vi, fRsrc, oi, 0, 0);
        const float fjx = reduceOver8Lanes(-f_ij.x), fjy = reduceOver8Lanes(-f_ij.y), fjz = reduceOver8Lanes(-f_ij.z);
        const float vj  = (tidxi == 0U) ? fjx : ((tidxi == 1U) ? fjy : fjz);
        const int   oj  = (tidxi < 3U) ? (3 * aj + static_cast<int>(tidxi)) * static_cast<int>(sizeof(float)) : c_dropLane;
        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(vj, fRsrc, oj, 0, 0);
    }

    if (primary && bCalcFshift && !central)
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
        const int   slot = item & (c_numEnergySlots - 1);
        const float v    = waveSum4Transposed(E_lj, E_el, DVDL_lj, DVDL_el, lane); /* lanes 0 .. 3: the four sums */
        if (primary && lane < 4U) { atomicAdd(energySlots + slot * c_energySlotStride + static_cast<int>(lane), v); }
    }

    /* ---- foreign lambdas (dH/dl steps): the same pair's energies at every lambda index ---------------------------------------
     * Index 0 is the current lambda: this lane's own terms of the pass above (fep0).  For the others only what depends on lambda is
     * evaluated again: erf(beta r) / r — a libm call with two branches — is computed once; and a wave none of whose lanes has a
     * perturbed pair within the cut-off or a self term has nothing to add at any lambda. */
    if constexpr (FOREIGN)
    {
        const bool  doPair  = inList && pert && !subDiag;
        const float rcMax2  = fmaxf(nbp.rcoulomb_sq, nbp.rvdw_sq);
        const bool  hasTerm = selfLane || (doPair && (!included || r2 < rcMax2));
        if (__ballot(hasTerm) != 0ULL)
        {
            float vLr = -1.0F;
            if constexpr (FEP_ELEC >= ELK_EWALD_ANA)
            {
                const float r2c  = fmaxf(r2, c_nbnxnMinDistanceSquared);
                const float rInv = __frsqrt_rn(r2c);
                vLr              = fepEwaldPotentialLr(nbp.ewald_beta, r2c * rInv, rInv);
            }
            float* slot = foreignSlots + (item & (c_numForeignSlots - 1)) * atdat.foreignSlotStride;
            const float v0 = waveSum4Transposed(fep0[0], fep0[1], fep0[2], fep0[3], lane); /* lanes 0 .. 3: the four sums at index 0 */
            const int   numIdx = numForeignLambda + 1;
            /* this wave's lambda indices: idxFirst, idxFirst + idxStep, ... (numMine of them) */
            const int idxFirst = (foreignChunk < 0) ? 1 : 1 + foreignChunk;
            const int idxStep  = (foreignChunk < 0) ? 1 : c_fepForeignHeavyChunks;
            const int numMine  = (numForeignLambda >= idxFirst) ? (numForeignLambda - idxFirst) / idxStep + 1 : 0;
            /* The lanes with a term are few — a cluster pair with one perturbed atom has 8 perturbed atom pairs in its 64 lanes —, and
             * walking the lambda indices one after the other evaluates the soft-core pair 11 times with most lanes idle (16 us per wave
             * on the 96k box, the whole tail of a dH/dlambda step).  Instead the (pair, lambda index) combinations are dealt out over
             * the 64 lanes: the lanes with a term leave their pair in LDS (8 values), every lane picks up combination
             * w = pass * 64 + lane -> pair w / numForeignLambda at index 1 + w mod numForeignLambda with ITS lambdas, and adds its four results to per-index
             * accumulators in LDS (ds_add_f32), which lanes 0 .. 4 numIdx - 1 then add to the wave's slot with one atomic each.
             * 8 pairs x 11 indices: 2 passes instead of 11.  Not for LJ-PME (two more values per pair than the staging buffers hold). */
            const unsigned long long termMask = __ballot(hasTerm);
            const int                n        = __builtin_popcountll(termMask);
            /* (from ~40 pairs on the passes are nearly as many as the indices, and the lanes' own bookkeeping costs more than it saves) */
            /* lane k holds the lambdas of index k + 1 (one load for all indices): fetched through the crossbar, or lane by lane */
            const bool  lamLane = static_cast<int>(lane) < numForeignLambda;
            const float lcLane  = lamLane ? nbp.allLambdaCoul[lane] : 0.0F;
            const float lvLane  = lamLane ? nbp.allLambdaVdw[lane] : 0.0F;
            const bool compact = !LJ_EWALD && waveLds != nullptr && 4 * numIdx <= static_cast<int>(c_waveSize) && n <= c_fepForeignCompactMaxPairs;
            if (compact)
            {
                typedef __attribute__((address_space(3))) float LdsF;
                LdsF* rec = (LdsF*)waveLds;       /* [8][64] */
                LdsF* acc = rec + 8 * c_waveSize; /* [4][numIdx] */
                const int rank = static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(termMask >> 32),
                                                                            __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(termMask), 0U)));
                acc[lane] = 0.0F;
                if (lane < 4U) { acc[static_cast<int>(lane) * numIdx] = v0; }
                if (hasTerm)
                {
                    /* (a lane is either an atom's self term or a pair: the diagonal lanes of a cluster's pair with itself are not pairs.
                     * Two flags ride in sign bits: r^2 and 12 C12 are not negative) */
                    const float sA = qABi.x * qABi.x / nbp.epsfac * selfCoef;
                    const float sB = qABi.y * qABi.y / nbp.epsfac * selfCoef;
                    rec[0 * c_waveSize + rank] = included ? r2 : -r2;
                    rec[1 * c_waveSize + rank] = selfLane ? sA : qq[0];
                    rec[2 * c_waveSize + rank] = selfLane ? sB : qq[1];
                    rec[3 * c_waveSize + rank] = c6AB[0];
                    rec[4 * c_waveSize + rank] = c6AB[1];
                    rec[5 * c_waveSize + rank] = selfLane ? -c12AB[0] : c12AB[0];
                    rec[6 * c_waveSize + rank] = c12AB[1];
                    rec[7 * c_waveSize + rank] = vLr;
                }
                const int   numComb = n * numMine;
                const float nfInv   = 1.0F / static_cast<float>(max(numMine, 1));
#pragma clang loop unroll(disable)
                for (int w0 = 0; w0 < numComb; w0 += static_cast<int>(c_waveSize))
                {
                    const int w     = w0 + static_cast<int>(lane);
                    const int wc    = max(min(w, numComb - 1), 0);
                    /* the index runs fastest: the lanes of a pass spread over all accumulators, and the lanes of a pair read one address */
                    const int a     = static_cast<int>((static_cast<float>(wc) + 0.5F) * nfInv); /* exact: (w + 1/2) / numMine is no integer */
                    const int fi    = idxFirst - 1 + (wc - a * numMine) * idxStep;          /* lambda index - 1 */
                    const float lc  = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(fi << 2, __builtin_bit_cast(int, lcLane)));
                    const float lv  = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(fi << 2, __builtin_bit_cast(int, lvLane)));
                    if (w < numComb)
                    {
                        const float sr2   = rec[0 * c_waveSize + a];
                        const float pq[2] = { rec[1 * c_waveSize + a], rec[2 * c_waveSize + a] };
                        const float pc6[2]  = { rec[3 * c_waveSize + a], rec[4 * c_waveSize + a] };
                        const float sc12    = rec[5 * c_waveSize + a];
                        const float pc12[2] = { fabsf(sc12), rec[6 * c_waveSize + a] };
                        const float pvLr    = rec[7 * c_waveSize + a];
                        float fE_lj = 0.0F, fE_el = 0.0F, fDVDL_lj = 0.0F, fDVDL_el = 0.0F, fscal = 0.0F;
                        if (__builtin_bit_cast(int, sc12) < 0)
                        {
                            fE_el    = (1.0F - lc) * pq[0] + lc * pq[1];
                            fDVDL_el = pq[1] - pq[0];
                        }
                        else
                        {
                            const FepLambda Lf        = makeFepLambda(lc, lv, nbp.lam_power, nbp.alpha_coul, nbp.alpha_vdw);
                            const float     noGrid[2] = { 0.0F, 0.0F };
                            fepPair<FEP_ELEC, VDW == VDK_PSWITCH, false, true>(nbp, Lf, fabsf(sr2), __builtin_bit_cast(int, sr2) >= 0, false, pq, pc6, pc12, fscal,
                                                                               fE_lj, fE_el, fDVDL_lj, fDVDL_el, noGrid, pvLr);
                        }
                        LdsF* dst = acc + fi + 1;
                        if (fE_lj != 0.0F) { __builtin_amdgcn_ds_faddf(dst, fE_lj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT, false); }
                        if (fE_el != 0.0F) { __builtin_amdgcn_ds_faddf(dst + numIdx, fE_el, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT, false); }
                        if (fDVDL_lj != 0.0F) { __builtin_amdgcn_ds_faddf(dst + 2 * numIdx, fDVDL_lj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT, false); }
                        if (fDVDL_el != 0.0F) { __builtin_amdgcn_ds_faddf(dst + 3 * numIdx, fDVDL_el, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT, false); }
                    }
                }
                /* (the accumulators have the slot's layout: [E_lj | E_el | dV/dl_lj | dV/dl_el][index]) */
                if (static_cast<int>(lane) < 4 * numIdx)
                {
                    const float v = acc[lane];
                    if (v != 0.0F) { atomicAdd(slot + static_cast<int>(lane), v); }
                }
            }
            else
            {
                if (lane < 4U && v0 != 0.0F) { atomicAdd(slot + static_cast<int>(lane) * numIdx, v0); }
                for (int fidx = idxFirst; fidx <= numForeignLambda; fidx += idxStep)
                {
                    const bool  inLanes = numForeignLambda <= static_cast<int>(c_waveSize);
                    const float lc = inLanes ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lcLane), fidx - 1)) : nbp.allLambdaCoul[fidx - 1];
                    const float lv = inLanes ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lvLane), fidx - 1)) : nbp.allLambdaVdw[fidx - 1];
                    const FepLambda Lf = makeFepLambda(lc, lv, nbp.lam_power, nbp.alpha_coul, nbp.alpha_vdw);
                    float fE_lj = 0.0F, fE_el = 0.0F, fDVDL_lj = 0.0F, fDVDL_el = 0.0F, fscal = 0.0F;
                    if (selfLane)
                    {
                        const float sA = qABi.x * qABi.x / nbp.epsfac * selfCoef;
                        const float sB = qABi.y * qABi.y / nbp.epsfac * selfCoef;
                        fE_el += (1.0F - lc) * sA + lc * sB;
                        fDVDL_el += sB - sA;
                        if constexpr (LJ_EWALD)
                        {
                            fE_lj += ((1.0F - lv) * c6gridAB[0] + lv * c6gridAB[1]) * ljSelfCoef;
                            fDVDL_lj += (c6gridAB[1] - c6gridAB[0]) * ljSelfCoef;
                        }
                    }
                    if (doPair)
                    {
                        fepPair<FEP_ELEC, VDW == VDK_PSWITCH, false, true>(nbp, Lf, r2, included, false, qq, c6AB, c12AB, fscal, fE_lj, fE_el, fDVDL_lj,
                                                                           fDVDL_el, c6gridAB, vLr);
                    }
                    const float v = waveSum4Transposed(fE_lj, fE_el, fDVDL_lj, fDVDL_el, lane); /* lanes 0 .. 3: the four sums */
                    /* into this wave's accumulator slot (NBAtomDataGpu::foreignSlots): thousands of waves adding to the same
                     * 48 addresses serialise in L2 (measured +0.28 ms per dH/dl step) */
                    if (lane < 4U && v != 0.0F) { atomicAdd(slot + static_cast<int>(lane) * numIdx + fidx, v); }
                }
            }
        }
    }
}

#endif
