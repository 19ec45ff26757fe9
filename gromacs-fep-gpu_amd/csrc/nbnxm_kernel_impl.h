/*
 * Cluster-pair non-bonded kernel for gfx950 (wave64), optionally with the perturbed pairs fused in.
 *
 * Replaces  nbnxn_kernel_<Elec>_<Vdw>_{F,VF}_cuda  (nbnxm/cuda/nbnxm_cuda_kernel.cuh:141-702) and, in its
 * FUSED flavour, also  nbnxn_fep_kernel_*  (nbnxm/cuda/nbnxm_fep_cuda_kernel.cuh:87-628) for the force path.
 *
 * Mapping (MI355X-first, not the CUDA one):
 *   - one 64-lane wavefront = one i-super-cluster entry (nbnxn_sci_t), 1-4 wavefronts per workgroup;
 *     lane = tidxj*8 + tidxi covers a complete 8 x (4+4) cluster pair per step, lanes 0-31 read the
 *     exclusion words of imei[0], lanes 32-63 those of imei[1] (the reference's split-2 list, unchanged);
 *   - list words (sci, cjPacked) are wave-uniform and come through scalar loads (restrict kernel arguments,
 *     indices pinned with readfirstlane); the imask tests are scalar branches, so a skipped cluster pair
 *     costs no vector issue, and list reads retire through lgkmcnt, never behind an atomic;
 *   - the 8 i-atoms a lane meets (x, q*epsfac, +shift, type row / LJ parameters) stay in 40 VGPRs for the
 *     whole entry (the 512-entry register file makes the CUDA kernel's per-pair LDS round trip unnecessary);
 *     LDS holds what is indexed at run time: the whole nbfp table (shared by the workgroup's waves, one
 *     ds_read_b64 per pair instead of a 512-byte global gather);
 *   - j-atom data: one 16-byte load per lane of 128 contiguous bytes per j-cluster;
 *   - j-forces: 3 DPP adds per component over the 8 lanes that share a j atom, then one no-return buffer
 *     atomic from 24 lanes (96 contiguous bytes), deferred behind the next j-cluster's loads: gfx950 retires
 *     loads and atomics through one in-order vmcnt counter, so the order load -> atomic keeps every wait a
 *     counted vmcnt(1-2) instead of a full drain behind a ~1-3 us memory-side atomic;
 *   - i-forces: 24 accumulators in registers, reduced over tidxj once per entry and written with
 *     64-lane coalesced atomics (768 contiguous bytes);
 *   - FUSED: (i-cluster, j-cluster) pairs that touch a perturbed atom (Grid::fepBits) are masked out of the list
 *     words with a per-group mask computed once per list (gpu_plist::groupSlowMask, one more staged dword and one
 *     scalar AND per group) and evaluated by nbnxmFepClusterKernel on the FEP stream; the main pass is the plain kernel.
 *   - built with -fno-slp-vectorize: on gfx950 v_pk_*_f32 issues at half rate and costs v_mov shuffles.
 */
#ifndef NBNXM_KERNEL_IMPL_H
#define NBNXM_KERNEL_IMPL_H

#include "nbnxm_device_helpers.h"

template<int VDW>
struct VdwTraits
{
    static constexpr bool ljEwald  = (VDW == VDK_EWALD_GEOM || VDW == VDK_EWALD_LB);
    static constexpr bool useTable = (VDW == VDK_CUT || VDW == VDK_FSWITCH || VDW == VDK_PSWITCH || ljEwald);
};

/* LDS-direct loads (global -> LDS without VGPRs; completion is counted by vmcnt): every active lane moves 16
 * (4) bytes from base + offset to LDS address ldsBase + 16 (4) * lane.  ldsBase is wave-uniform and travels in
 * M0 (semantics checked on MI355X by tools/ubench/lds_dma_test.hip).  Written as asm so that the compiler
 * neither adds its own waits for these loads nor reorders them: the waits are the counted s_waitcnt vmcnt in
 * the group loop. */
NB_DEVINL void ldsDirectLoad16(unsigned ldsBase, unsigned offset, const void* base)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsBase), "v"(offset), "s"(base) : "memory");
}
NB_DEVINL void ldsDirectLoad4(unsigned ldsBase, unsigned offset, const void* base)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(ldsBase), "v"(offset), "s"(base) : "memory");
}

typedef int nb_int4 __attribute__((ext_vector_type(4)));

/* An i-entry record through the scalar unit, whatever the compiler thinks of the memory behind the pointer: inside the piece loop —
 * behind the force atomics — it no longer proves the list read-only and loads the record's words one by one through the VECTOR
 * memory pipeline, each followed by a vmcnt(0), i.e. by a wait for the memory-side atomics the wave has just sent (measured: + 0.65 us
 * per piece of a range, + 2 - 3 us on the waves with three or four pieces, which are the last ones to finish on a 24k-atom box). */
NB_DEVINL nbnxn_sci_t scalarLoadSci(const nbnxn_sci_t* __restrict__ entry)
{
    nb_int4 r;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(entry) : "memory");
    nbnxn_sci_t e;
    e.sci           = r.x;
    e.shift         = r.y;
    e.cjPackedBegin = r.z;
    e.cjPackedEnd   = r.w;
    return e;
}

/* lane index within the wave, recomputed in place (2 VALU ops, no live register): volatile so that it is
 * neither hoisted out of a loop nor merged with an earlier copy */
NB_DEVINL unsigned laneIdNow()
{
    unsigned l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

/* lanes whose byte offset lies beyond the buffer are dropped by the hardware range check */
constexpr int c_dropLane = 0x7FFFFFF0;

NB_DEVINL void ljFromComb(int vdwKind, const float2& a, const float2& b, float& c6, float& c12)
{
    if (vdwKind == VDK_COMB_GEOM)
    {
        c6  = a.x * b.x;
        c12 = a.y * b.y;
    }
    else
    {
        const float sigma  = a.x + b.x;
        const float eps    = a.y * b.y;
        const float sigma2 = sigma * sigma;
        const float sigma6 = sigma2 * sigma2 * sigma2;
        c6                 = eps * sigma6;
        c12                = c6 * sigma6;
    }
}

#include "nbnxm_fep_cluster_body.h"
#include "nbnxm_fep_kernel_impl.h" /* fepAtomPairWave: the atom-pair list (reference shape) in trailing workgroups */

/* First-pass / rolling list pruning of ONE i-entry by one wavefront (nbnxm/cuda/nbnxm_cuda_kernel_pruneonly.cuh:100-316): a cluster
 * pair is kept when any of its 64 atom pairs is within range, and the same mask is written to both halves of the split-2 entry.
 * The i-atoms a lane meets (atom tidxi of the 8 i-clusters) stay in registers: no LDS, no barrier, so the function serves the
 * prune kernel and the trailing workgroups of the force kernel alike. */
template<bool haveFreshList>
NB_DEVINL void pruneEntry(const NBAtomDataGpu& atdat, const NBParamGpu& nbp, const gpu_plist& plist, const int entry,
                          const int firstGroupOfChunk = 0, const int groupsPerChunk = 0x7FFFFFFF /* first pass: a part of the entry */)
{
    const unsigned lane  = threadIdx.x & (c_waveSize - 1);
    const unsigned tidxi = lane & 7U;
    const unsigned tidxj = lane >> 3;

    const nbnxn_sci_t nb_sci   = plist.sci[entry];
    if (haveFreshList && firstGroupOfChunk >= nb_sci.cjPackedEnd - nb_sci.cjPackedBegin) { return; } /* a chunk beyond this entry's end */
    const int         shiftIdx = nb_sci.shift & NBNXM_CI_SHIFT_MASK;
    float4            xi[c_numClPerSupercl];
    {
        const float3 sh = atdat.shiftVec[shiftIdx];
#pragma unroll
        for (int i = 0; i < c_numClPerSupercl; i++)
        {
            float4 v = atdat.xq[nb_sci.sci * c_superClSize + i * c_clSize + static_cast<int>(tidxi)];
            v.x += sh.x;
            v.y += sh.y;
            v.z += sh.z;
            xi[i] = v;
        }
    }

    const float rlistOuter_sq = nbp.rlistOuter_sq;
    const float rlistInner_sq = nbp.rlistInner_sq;

    /* the pair checks of one packed group; returns the new working mask */
    auto checkGroup = [&](const int jPacked, unsigned& imaskFull, const unsigned imaskCheck, unsigned imaskNew) {
#pragma unroll 1
        for (int jm = 0; jm < c_jGroupSize; jm++)
        {
            if (!(imaskCheck & (0xFFU << (jm * c_numClPerSupercl)))) { continue; }
            const int    cj = plist.cjPacked[jPacked].cj[jm];
            const float4 xj = atdat.xq[cj * c_clSize + static_cast<int>(tidxj)];
#pragma unroll
            for (int i = 0; i < c_numClPerSupercl; i++)
            {
                const unsigned mask_ji = 1U << (jm * c_numClPerSupercl + i);
                if (imaskCheck & mask_ji)
                {
                    const float dx = xi[i].x - xj.x, dy = xi[i].y - xj.y, dz = xi[i].z - xj.z;
                    const float r2 = dx * dx + dy * dy + dz * dz;
                    if constexpr (haveFreshList)
                    {
                        if (__ballot(r2 < rlistOuter_sq) == 0ULL) { imaskFull &= ~mask_ji; }
                    }
                    if (__ballot(r2 < rlistInner_sq) != 0ULL) { imaskNew |= mask_ji; }
                }
            }
        }
        return imaskNew;
    };

    if constexpr (haveFreshList)
    {
        const int chunkBegin = nb_sci.cjPackedBegin + firstGroupOfChunk;
        const int chunkEnd   = min(nb_sci.cjPackedEnd, chunkBegin + min(groupsPerChunk, nb_sci.cjPackedEnd - chunkBegin));
        for (int jPacked = chunkBegin; jPacked < chunkEnd; jPacked++)
        {
            /* the whole group at once: its record as two 16-byte loads, then the four j-clusters' coordinates side by side (every cj
             * of a record is a valid cluster, gpu_init_pairlist checks it) — two round trips per group instead of one per j-cluster */
            const nb_int4* rec = reinterpret_cast<const nb_int4*>(&plist.cjPacked[jPacked]);
            const nb_int4  cjs = rec[0];
            unsigned       imaskFull = static_cast<unsigned>(rec[1].x);
            const float4   xj0 = atdat.xq[cjs.x * c_clSize + static_cast<int>(tidxj)];
            const float4   xj1 = atdat.xq[cjs.y * c_clSize + static_cast<int>(tidxj)];
            const float4   xj2 = atdat.xq[cjs.z * c_clSize + static_cast<int>(tidxj)];
            const float4   xj3 = atdat.xq[cjs.w * c_clSize + static_cast<int>(tidxj)];
            unsigned       imaskNew = 0U;
            const unsigned imaskCheck = imaskFull;
#pragma unroll
            for (int jm = 0; jm < c_jGroupSize; jm++)
            {
                if (!(imaskCheck & (0xFFU << (jm * c_numClPerSupercl)))) { continue; }
                const float4 xj = (jm == 0) ? xj0 : ((jm == 1) ? xj1 : ((jm == 2) ? xj2 : xj3));
#pragma unroll
                for (int i = 0; i < c_numClPerSupercl; i++)
                {
                    const unsigned mask_ji = 1U << (jm * c_numClPerSupercl + i);
                    if (imaskCheck & mask_ji)
                    {
                        const float dx = xi[i].x - xj.x, dy = xi[i].y - xj.y, dz = xi[i].z - xj.z;
                        const float r2 = dx * dx + dy * dy + dz * dz;
                        if (__ballot(r2 < rlistOuter_sq) == 0ULL) { imaskFull &= ~mask_ji; }
                        if (__ballot(r2 < rlistInner_sq) != 0ULL) { imaskNew |= mask_ji; }
                    }
                }
            }
            if (lane == 0U)
            {
                plist.imask[jPacked * NBNXM_GPU_CLUSTERPAIR_SPLIT]     = imaskFull;
                plist.imask[jPacked * NBNXM_GPU_CLUSTERPAIR_SPLIT + 1] = imaskFull;
                plist.cjPacked[jPacked].imei[0].imask                  = imaskNew;
                plist.cjPacked[jPacked].imei[1].imask                  = imaskNew;
            }
        }
    }
    else
    {
        /* rolling pass: only cluster pairs of the outer list that are not in the working list are looked at, and in most groups
         * there is none.  64 groups at a time: lane l reads the two masks of group base + l (one round trip for the whole
         * entry instead of one per group), a ballot finds the groups with something to check. */
        for (int base = nb_sci.cjPackedBegin; base < nb_sci.cjPackedEnd; base += c_waveSize)
        {
            const int g    = base + static_cast<int>(lane);
            unsigned  full = 0U, cur = 0U;
            if (g < nb_sci.cjPackedEnd)
            {
                full = plist.imask[g * NBNXM_GPU_CLUSTERPAIR_SPLIT];
                cur  = plist.cjPacked[g].imei[0].imask;
            }
            unsigned long long need = __ballot((full ^ cur) != 0U);
            while (need != 0ULL)
            {
                const int l = __builtin_ctzll(need);
                need &= need - 1ULL;
                unsigned       imaskFull = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(full), l));
                const unsigned imaskCur  = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(cur), l));
                const unsigned imaskNew  = checkGroup(base + l, imaskFull, imaskCur ^ imaskFull, imaskCur);
                if (lane == 0U && imaskNew != imaskCur)
                {
                    plist.cjPacked[base + l].imei[0].imask = imaskNew;
                    plist.cjPacked[base + l].imei[1].imask = imaskNew;
                }
            }
        }
    }
}

/* The unrolled loop over the 8 i-clusters of one j-cluster, as a macro so that both instances index the
 * kernel's register arrays (xqi, trow, fci_buf) directly: a lambda capturing them by reference sends them to
 * scratch memory.  Diagonal rule: on the central image a cluster paired with itself keeps only j > i
 * (wave-uniform, rare: the empty asm keeps it a scalar branch instead of a per-pair select chain). */
/* diagnostics build (-DNBNXM_BLOCK_STATS, tools/block_stats_probe.py): how the lanes of the LISTED pair blocks are used — per block the
 * ballot of the lanes within the cut-off; counters (64-bit, behind the timeline area of gpu_plist::debugTimeline): 0 listed blocks,
 * 1 executed blocks (any lane active), 2 / 3 executed blocks whose lower / upper 32-lane half is empty, 4 sum of active lanes,
 * 5 executed blocks with at least one empty 16-lane quarter, 6 sum of empty quarters over the executed blocks */
#ifdef NBNXM_BLOCK_STATS
#define NBNXM_BLOCK_STATS_HOOK \
    { \
        const unsigned long long bal_ = __ballot(active); \
        if (laneIdNow() == 0U && plist.debugTimeline != nullptr) \
        { \
            unsigned long long* st_ = plist.debugTimeline + 4 * 16384 - 64; \
            const unsigned      lo_ = static_cast<unsigned>(bal_), hi_ = static_cast<unsigned>(bal_ >> 32); \
            int                 eq_ = 0; \
            for (int q_ = 0; q_ < 4; q_++) { eq_ += (((bal_ >> (16 * q_)) & 0xFFFFULL) == 0ULL) ? 1 : 0; } \
            atomicAdd(st_ + 0, 1ULL); \
            if (bal_ != 0ULL) \
            { \
                atomicAdd(st_ + 1, 1ULL); \
                if (lo_ == 0U) { atomicAdd(st_ + 2, 1ULL); } \
                if (hi_ == 0U) { atomicAdd(st_ + 3, 1ULL); } \
                atomicAdd(st_ + 4, static_cast<unsigned long long>(__popcll(bal_))); \
                if (eq_ > 0) { atomicAdd(st_ + 5, 1ULL); } \
                atomicAdd(st_ + 6, static_cast<unsigned long long>(eq_)); \
            } \
        } \
    }
#else
#define NBNXM_BLOCK_STATS_HOOK
#endif
#define NBNXM_PAIR_LOOP(HAS_EXCL)                                                                          \
    _Pragma("unroll") \
                    for (int i = 0; i < c_numClPerSupercl; i++) \
                    { \
                        if (fastMask & (1U << i)) \
                        { \
                            const float3 rv      = make_float3(xqi[i].x - xqj.x, xqi[i].y - xqj.y, xqi[i].z - xqj.z); \
                            /* (the sum starts from c_r2Floor, see nbPair: same instruction count, a v_fmaak instead of a v_mul) */ \
                            const float  r2      = fmaf(rv.z, rv.z, fmaf(rv.y, rv.y, fmaf(rv.x, rv.x, c_r2Floor))); \
                            int          intMask = -1; \
                            bool         active  = (r2 < rcoulomb_sq); \
                            if constexpr (HAS_EXCL) \
                            { \
                                intMask = __builtin_amdgcn_sbfe(static_cast<int>(wexclJ), i, 1); \
                                if constexpr (EXCL_FORCES) \
                                { \
                                    if (diagBits & (1U << i)) \
                                    { \
                                        asm volatile("" ::: "memory"); \
                                        active = active && (tidxj > tidxi); \
                                    } \
                                } \
                                else { active = active && (intMask != 0); } \
                            } \
                            NBNXM_BLOCK_STATS_HOOK \
                            if (active) \
                            { \
                                float c6, c12; \
                                if constexpr (USE_TABLE) \
                                { \
                                    const float2 c6c12 = *reinterpret_cast<const float2*>(nbLds + c_ewaldTabBytes + trow[i] + typejBytes); \
                                    c6                 = c6c12.x; \
                                    c12                = c6c12.y; \
                                } \
                                else { ljFromComb(VDW, ljcpi[i], ljcp_j, c6, c12); } \
                                float F_invr, E_lj_p = 0.0F, E_el_p = 0.0F, c6grid = 0.0F; \
                                if constexpr (LJ_EWALD) { c6grid = ljGridC6(VDW, ljcpi[i], ljcp_j); } \
                                if constexpr (ENERGY_HEADLINE) \
                                { \
                                    nbPairEnergy<ELEC>(nbp, r2, intMask, xqi[i].w * xqj.w, c6, c12, ewaldTabScaleV, F_invr, eSums); \
                                } \
                                else \
                                { \
                                    nbPair<ELEC, TWIN, VDW, ENERGY, EXCL_FORCES, HAS_EXCL>(nbp, ewaldCorrLds, r2, intMask, xqi[i].w * xqj.w, c6, c12, \
                                                                                           F_invr, E_lj_p, E_el_p, c6grid, ewaldTabScaleV); \
                                    if constexpr (ENERGY) \
                                    { \
                                        E_lj += E_lj_p; \
                                        E_el += E_el_p; \
                                    } \
                                } \
                                const float3 f_ij = make_float3(rv.x * F_invr, rv.y * F_invr, rv.z * F_invr); \
                                fcj_buf.x -= f_ij.x; \
                                fcj_buf.y -= f_ij.y; \
                                fcj_buf.z -= f_ij.z; \
                                fci_buf[i].x += f_ij.x; \
                                fci_buf[i].y += f_ij.y; \
                                fci_buf[i].z += f_ij.z; \
                            } \
                        } \
                    }

/* 5 waves per SIMD (<= 96 VGPRs) for the flavours that fit without scratch; the energy, combination-rule and
 * switch flavours carry more live values and run at 4 waves (<= 128 VGPRs) instead of spilling. */
#ifndef NBNXM_FORCE_WAVES_PER_EU
#define NBNXM_FORCE_WAVES_PER_EU 5
#endif
#ifndef NBNXM_COMB_WAVES_PER_EU
#define NBNXM_COMB_WAVES_PER_EU 4
#endif
#ifndef NBNXM_PSWITCH_WAVES_PER_EU
#define NBNXM_PSWITCH_WAVES_PER_EU 5
#endif
template<int VDW, bool ENERGY>
constexpr int c_nbWavesPerEu = ENERGY ? 4
                               : (VDW == VDK_CUT || VDW == VDK_FSWITCH) ? NBNXM_FORCE_WAVES_PER_EU
                               : (VDW == VDK_PSWITCH)                   ? NBNXM_PSWITCH_WAVES_PER_EU
                               : (VDW == VDK_COMB_GEOM || VDW == VDK_COMB_LB) ? NBNXM_COMB_WAVES_PER_EU
                                                                        : 4;

/* FUSED: the cluster pairs that touch a perturbed atom are masked out of the list words (gpu_plist::groupSlowMask) and
 * left to nbnxmFepClusterKernel; otherwise the kernel is the plain one */
template<int ELEC, bool TWIN, int VDW, bool ENERGY, bool FUSED>
__launch_bounds__(c_nbMaxBlockSize) __attribute__((amdgpu_waves_per_eu(c_nbWavesPerEu<VDW, ENERGY>))) __global__
        void nbnxmKernel(/* work partition (gpu_plist::work*): wave w owns the packed j-groups [workDesc[w].rangeBegin, .rangeEnd) and
                          * starts in entry workDesc[w].sciIdx of sciList, which is the list's i-entries ordered by cjPackedBegin;
                          * the record also carries that entry and the first group's indices (NbWorkDesc).  First in the list: the
                          * head of the wave's dependency chain (with kernel-argument preloading they arrive in registers) */
                         const NbWorkDesc* __restrict__ workDesc,
                         const int numWorkRanges,
                         const int wavesPerBlockLog2, /* the launch's workgroup size as log2(waves): with the two arguments above all a wave
                                                       * needs to address its start record, without a load from the dispatch packet */
                         const NBAtomDataGpu atdat,
                         const NBParamGpu    nbp,
                         const gpu_plist     plist,
                         const int           bCalcFshiftIn,
                         /* read-only, non-aliased views of members of the structs above: with noalias the
                          * compiler turns the wave-uniform list reads into scalar (SMEM) loads */
                         const nbnxn_sci_t* __restrict__ sciList,
                         const nbnxn_cj_packed_t* __restrict__ cjPackedList,
                         const nbnxn_excl_t* __restrict__ exclList,
                         const float4* __restrict__ xq,
                         const int* __restrict__ atomTypes,
                         const float2* __restrict__ ljComb,
                         const unsigned* __restrict__ fepWords, /* atdat.fepBits viewed as dwords (scalar loads) */
                         const unsigned* __restrict__ groupSlowMask, /* FUSED: perturbed cluster pairs of each group */
                         const int mergedFepItems, /* FUSED force flavour: perturbed cluster pairs for the trailing workgroups, or 0 */
                         /* force flavour: i-entries idx * pruneNumParts + prunePart, idx < pruneEntries, are rolling-pruned by trailing workgroups */
                         const int pruneNumParts,
                         const int prunePart,
                         const int pruneEntries,
                         /* force flavour: the spare force buffer, zeroed by the last trailing workgroups (or 0 float4) */
                         float4* __restrict__ clearF4,
                         const int clearNumFloat4,
                         /* ... and a second, small area: the spare copy of the scalar outputs and shift forces (or 0 float4) */
                         float4* __restrict__ clearB,
                         const int clearNumFloat4B,
                         /* energy flavour, dH/dlambda step: the perturbed cluster pairs of the trailing workgroups also accumulate their energies
                          * at lambda indices 0 .. mergedFepForeignLambdas (the FOREIGN flavour of fepClusterPair); -1: not such a step */
                         const int mergedFepForeignLambdas,
                         /* not FUSED (the reference's shape: perturbed pairs carved out of the cluster list into an atom-pair list): that list,
                          * evaluated by trailing workgroups — mergedFepItems waves, one per cluster-pair item of the regrouped list
                          * (gpu_feplist::clItem) — instead of a kernel of its own */
                         const gpu_feplist feplist)
{
    constexpr bool LJ_EWALD    = VdwTraits<VDW>::ljEwald;
    constexpr bool EXCL_FORCES = (ELEC != ELK_CUT) || ENERGY || LJ_EWALD; /* nbnxm_cuda_kernel.cuh:69-78 */
    constexpr bool USE_TABLE   = VdwTraits<VDW>::useTable;
    /* energy steps of the headline flavours: the fused energy pair block (nbPairEnergy) */
    constexpr bool ENERGY_HEADLINE = c_energyHeadlineBlock<ELEC, TWIN, VDW, ENERGY>;

#ifdef NBNXM_WAVE_TIMELINE
    const unsigned long long tlTop = wall_clock64(); /* the wave's first instruction */
#endif
    /* wave-uniform values are pinned to SGPRs with readfirstlane so that everything derived from them
     * (list walk, branches, list loads) stays on the scalar unit */
    const unsigned blockSize = static_cast<unsigned>(c_waveSize) << wavesPerBlockLog2;
    const unsigned lane      = threadIdx.x & (c_waveSize - 1);
    const unsigned wave      = __builtin_amdgcn_readfirstlane(threadIdx.x / c_waveSize);
    const unsigned tidxi     = lane & 7U;
    const unsigned tidxj     = lane >> 3;
    /* The start of this wave's range is ONE scalar load (NbWorkDesc: borders, the first i-entry, the first group's cluster and exclusion
     * indices), and it is the first thing the wave does: its address needs the preloaded arguments, the workgroup id and the wave's
     * number only.  (Trailing workgroups load record 0 and ignore it.) */
    const int        workItem = __builtin_amdgcn_readfirstlane(static_cast<int>((blockIdx.x << wavesPerBlockLog2) + wave));
    const bool       inLaunch = (workItem < numWorkRanges);
    /* (as asm: left to the compiler the load sinks to its first use, behind the trailing workgroups' branch and behind the round trip of the
     * other kernel arguments.  Load and wait in ONE statement: with the wait in a statement of its own the compiler takes the registers for
     * written when the load is ISSUED and may copy them before the data is there.  The address needs preloaded arguments only, so the wait
     * costs this one round trip, and the other kernel arguments' loads, issued ahead of it, travel beside it.) */
    typedef int nb_int16 __attribute__((ext_vector_type(16)));
    nb_int16 descWords;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(descWords) : "s"(workDesc + (inLaunch ? workItem : 0)) : "memory");

    /* LDS (all dynamic, sized by nbLdsBytes()): the LJ parameter table shared by the waves of the workgroup,
     * then per wave: two staging buffers for the j-side of a packed group (filled by LDS-direct loads, see the
     * group loop) */
    extern __shared__ __align__(16) unsigned char nbLds[];
    const int numTypes   = atdat.numTypes;
    /* LDS layout: the Ewald correction table (fixed size) first, then the LJ table: both bases are compile-time offsets, so a
     * table read needs no base-address add (the offset sits in the ds_read's immediate field) */
    constexpr bool EWALD_CORR_TABLE = (ELEC == ELK_EWALD_ANA);
    /* tabulated Ewald, energy steps: the potential correction {beta V, step} of the analytical flavours' table in front of the r-indexed
     * force table (the reference calls erff there, two divergent branches per pair) */
    constexpr bool EWALD_V_TABLE    = (ELEC == ELK_EWALD_TAB) && ENERGY;
    constexpr int  c_ewaldTabBytes  = EWALD_CORR_TABLE ? (ENERGY ? c_ewaldCorrTabSizeEnergy * static_cast<int>(sizeof(float4)) : c_ewaldCorrTabSize * static_cast<int>(sizeof(float2)))
                                                       : (EWALD_V_TABLE ? c_ewaldCorrTabSizeEnergy * static_cast<int>(sizeof(float2)) : 0);
    /* tabulated Ewald: the reference's r-indexed force table (run-time size) takes the same place; the LJ table's row offsets
     * (trow) carry its size, so that a table read still needs no base-address add */
    constexpr bool EWALD_R_TABLE = (ELEC == ELK_EWALD_TAB);
    const int      rTabBytes     = EWALD_R_TABLE ? __builtin_amdgcn_readfirstlane(coulombTabLdsBytes(nbp.coulombTabSize)) : 0;
    float2*   nbfpLds    = reinterpret_cast<float2*>(nbLds + c_ewaldTabBytes + rTabBytes);
    /* energy flavours (NBNXM_ENERGY_TAIL, nbnxm_hip_types.h): 0 no trailing workgroups; 1 rolling prune and buffer clear; 2 also the
     * perturbed cluster pairs with their energies and dV/dlambda */
    constexpr bool TAIL     = !ENERGY || (NBNXM_ENERGY_TAIL >= 1);
    constexpr bool TAIL_FEP = !ENERGY || (NBNXM_ENERGY_TAIL >= 2);
    /* the perturbed-pair math has three electrostatics forms (nbnxm_kernels_fep.hip: selectFepKernel) */
    constexpr int FEP_LIST_ELEC = (ELEC == ELK_CUT || ELEC == ELK_RF) ? ELK_RF : ELEC;
    if constexpr (TAIL)
    {
        /* Trailing workgroups, behind the ones of the ranges.  The dispatcher hands workgroups out in order, so their waves start
         * as the ranges' waves retire and run in the wave slots — and issue slots — that the end of the kernel leaves idle (one
         * wave per SIMD finishes alone), at the lowest priority: they must not take issue slots from the ranges.
         *  1. pruneEntries > 0: one i-entry of this step's rolling-prune part per wave, what nbnxmPruneKernel<false> does.  It
         *     rewrites list masks that the ranges may be reading: either value is right for this step (a cluster pair that is
         *     pruned has no atom pair within the inner list radius, so it adds exactly zero);
         *  2. FUSED, mergedFepItems > 0: one perturbed cluster pair per wave, what nbnxmFepClusterKernel does in a kernel of its own. */
        const unsigned wavesPerBlock = 1U << wavesPerBlockLog2;
        const unsigned mainBlocks    = (static_cast<unsigned>(numWorkRanges) + wavesPerBlock - 1U) >> wavesPerBlockLog2;
        if (blockIdx.x >= mainBlocks)
        {
            __builtin_amdgcn_s_setprio(0);
#ifdef NBNXM_WAVE_TIMELINE
            const unsigned long long tlTailStart = wall_clock64();
#endif
            /* the rolling-prune waves first: they are the longer chains (a loop over the entry's j-groups); the perturbed-pair
             * waves are short and fill what is left */
            const unsigned pruneBlocks = (static_cast<unsigned>(pruneEntries) + wavesPerBlock - 1U) >> wavesPerBlockLog2;
            const unsigned fepBlocks   = TAIL_FEP ? (static_cast<unsigned>(mergedFepItems) + wavesPerBlock - 1U) >> wavesPerBlockLog2 : 0U;
            if (blockIdx.x < mainBlocks + pruneBlocks)
            {
                const int idx   = __builtin_amdgcn_readfirstlane(static_cast<int>((blockIdx.x - mainBlocks) * wavesPerBlock + wave));
                const int entry = idx * pruneNumParts + prunePart;
                if (idx < pruneEntries && entry < plist.nsci) { pruneEntry<false>(atdat, nbp, plist, entry); }
            }
            else if (blockIdx.x < mainBlocks + pruneBlocks + fepBlocks)
            {
                if constexpr (TAIL_FEP && !FUSED)
                {
                    const int item = __builtin_amdgcn_readfirstlane(static_cast<int>((blockIdx.x - mainBlocks - pruneBlocks) * wavesPerBlock + wave));
                    if (item < mergedFepItems)
                    {
                        fepListClusterItem<FEP_LIST_ELEC, VDW == VDK_PSWITCH, ENERGY>(atdat, nbp, feplist, bCalcFshiftIn, item);
                    }
                }
                if constexpr (TAIL_FEP && FUSED)
                {
                    if constexpr (VdwTraits<VDW>::useTable)
                    {
                        for (int t = threadIdx.x; t < numTypes * numTypes; t += blockSize) { nbfpLds[t] = nbp.nbfp[t]; }
                        __syncthreads();
                    }
                    /* mergedFepItems is the size of the launch, the number of items is on the device (gpu_plist::slowCount): behind a
                     * new list the host queues this launch without waiting for the count — it sizes the launch from the previous
                     * list's — and the waves stride over whatever the partition pass has found */
                    const int firstItem = __builtin_amdgcn_readfirstlane(static_cast<int>((blockIdx.x - mainBlocks - pruneBlocks) * wavesPerBlock + wave));
                    const int numItems  = __builtin_amdgcn_readfirstlane(min(*plist.slowCount, plist.slowPairs_nalloc));
                    const int stride    = static_cast<int>(fepBlocks * wavesPerBlock);
                    /* this wave's staging area (unused in a trailing workgroup): scratch of the foreign-lambda terms */
                    constexpr bool LJ_EWALD_T = VdwTraits<VDW>::ljEwald;
                    const int      tabBytesT  = c_ewaldTabBytes + rTabBytes
                                          + (VdwTraits<VDW>::useTable ? (((numTypes * numTypes + (LJ_EWALD_T ? numTypes : 0)) * static_cast<int>(sizeof(float2)) + 15) & ~15) : 0);
                    float* waveLds = reinterpret_cast<float*>(nbLds + tabBytesT + wave * (2 * c_jStageBytes + c_jRingBytes));
                    if (ENERGY && mergedFepForeignLambdas >= 0)
                    {
                        /* dH/dlambda step: the heavy cluster pairs at the front of the list (slowCount[1] of them) are split over
                         * c_fepForeignHeavyChunks waves each by lambda index — one wave takes 20 us for such a pair, three times what the
                         * others take, and the kernel would end with it */
                        const int numHeavy   = __builtin_amdgcn_readfirstlane(min(plist.slowCount[1], numItems));
                        const int numVirtual = numItems + numHeavy * (c_fepForeignHeavyChunks - 1);
#pragma clang loop unroll(disable)
                        for (int v = firstItem; v < numVirtual; v += stride)
                        {
                            const bool heavy = v < numHeavy * c_fepForeignHeavyChunks;
                            const int  item  = heavy ? v / c_fepForeignHeavyChunks : v - numHeavy * (c_fepForeignHeavyChunks - 1);
                            const int  chunk = heavy ? v - item * c_fepForeignHeavyChunks : -1;
                            fepClusterPair<ELEC, TWIN, VDW, ENERGY, ENERGY>(atdat, nbp, plist, bCalcFshiftIn, cjPackedList, exclList, xq, ljComb,
                                                                            fepWords, mergedFepForeignLambdas, item, nbfpLds, waveLds, chunk);
                        }
                    }
                    else
                    {
#pragma clang loop unroll(disable)
                        for (int item = firstItem; item < numItems; item += stride)
                        {
                            fepClusterPair<ELEC, TWIN, VDW, ENERGY, false>(atdat, nbp, plist, bCalcFshiftIn, cjPackedList, exclList, xq, ljComb,
                                                                           fepWords, -1, item, nbfpLds);
                        }
                    }
                }
            }
            else
            {
                /* 3. the OTHER force buffer (clearNumFloat4 > 0), zeroed for the next step: nbnxm_gpu_clear_outputs then swaps the
                 *    two buffers instead of launching a kernel */
                const unsigned chunk   = blockSize * c_clearFloat4PerThread;
                const unsigned blocksA = (static_cast<unsigned>(clearNumFloat4) + chunk - 1U) / chunk;
                unsigned       idx     = blockIdx.x - mainBlocks - pruneBlocks - fepBlocks;
                float4*        dst     = clearF4;
                unsigned       num     = static_cast<unsigned>(clearNumFloat4);
                if (idx >= blocksA)
                {
                    idx -= blocksA;
                    dst = clearB;
                    num = static_cast<unsigned>(clearNumFloat4B);
                }
                const unsigned end = min((idx + 1U) * chunk, num);
                /* (streaming stores — `nt` — so that the 1.2 MB leave no dirty lines for the end of the kernel to write back: measured flat,
                 * 50.15 vs 50.27 us at 96k atoms, 18.5 vs 18.5 us at 24k; the 2 - 3 us between two launches are not a write-back) */
                for (unsigned i = idx * chunk + threadIdx.x; i < end; i += blockSize) { dst[i] = make_float4(0.0F, 0.0F, 0.0F, 0.0F); }
            }
#ifdef NBNXM_WAVE_TIMELINE
            {
                /* the trailing waves' records follow the ranges' ones; word 1 = 1 prune, 2 perturbed pairs, 3 clear */
                const unsigned rec = static_cast<unsigned>(numWorkRanges) + (blockIdx.x - mainBlocks) * wavesPerBlock + wave;
                if (lane == 0U && rec < 16384U && plist.debugTimeline != nullptr)
                {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    unsigned hwId, xccId;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwId));
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xccId));
                    plist.debugTimeline[4 * rec + 0] = tlTailStart;
                    plist.debugTimeline[4 * rec + 1] = (blockIdx.x < mainBlocks + pruneBlocks) ? 1U : ((blockIdx.x < mainBlocks + pruneBlocks + fepBlocks) ? 2U : 3U);
                    plist.debugTimeline[4 * rec + 2] = wall_clock64();
                    plist.debugTimeline[4 * rec + 3] = (static_cast<unsigned long long>(xccId) << 32) | hwId;
                }
            }
#endif
            return;
        }
    }
    /* LJ-PME: the per-type grid parameters (nbfp_comb) follow the pair table */
    const int      nbfpEntries = numTypes * numTypes + (LJ_EWALD ? numTypes : 0);
    const int      nbfpBytes   = USE_TABLE ? ((nbfpEntries * static_cast<int>(sizeof(float2)) + 15) & ~15) : 0;
    const int      tableBytes = nbfpBytes + c_ewaldTabBytes + rTabBytes;
    /* (tabulated Ewald on energy steps: the r-indexed table sits behind the potential table) */
    [[maybe_unused]] const float2* ewaldCorrLds = reinterpret_cast<const float2*>(nbLds + (EWALD_V_TABLE ? c_ewaldTabBytes : 0));
    unsigned char* jStage = nbLds + tableBytes + wave * (2 * c_jStageBytes + c_jRingBytes);
    /* Every wave of the launch gets the same amount of pair work: a contiguous range of packed j-groups cut
     * out of the list by weight (nbnxmWorkRangesKernel), regardless of i-entry borders; the launch has one
     * wave per resident wave slot.  Inside its range the wave walks "pieces": the part of an i-entry's j-list
     * that lies in the range.
     * The start of the range is ONE scalar load (NbWorkDesc: borders, the first i-entry, the first group's cluster and exclusion
     * indices), issued before anything else; what depends on it — the list words of the first groups, the first group's j-side and
     * the i-atoms — goes out in one batch, and the staging of the tables into LDS runs beside that batch: two dependent round
     * trips between the start of a wave and its first pair, where there were four (range borders -> i-entry -> list words -> j
     * data) behind the table staging. */
    /* (For the compiler's wait-count pass the trailing workgroups' branch above is a predecessor of this code — the structurizer routes
     * its exits through a common block —, so it believes that loads into the registers used below may be pending and puts a
     * vmcnt(0) in front of every write to them: between the LDS-direct loads of the batch, which it cannot see, that is a full round
     * trip each.  A wait it CAN see, here, where this wave has nothing in flight, clears that state: vmcnt(0), expcnt and lgkmcnt left alone.) */
    __builtin_amdgcn_s_waitcnt(0x0F70);
#ifdef NBNXM_WAVE_TIMELINE
    /* diagnostics build only (tools/timeline_budget.py): per wave {start, first group's data arrived, end, HW_ID} in 100 MHz ticks */
    const unsigned long long tlStart = wall_clock64();
    unsigned long long       tlFirst = 0; /* first group's data has arrived */
    unsigned long long       tlTransReduce = 0, tlTransWait = 0, tlTransCollect = 0, tlTransCount = 0; /* piece transitions of this wave */
#endif
    NbWorkDesc desc;
    desc.rangeBegin = descWords[0];
    desc.rangeEnd   = descWords[1];
    desc.sciIdx     = descWords[2];
    desc.firstGroup = descWords[3];
    desc.entry      = nbnxn_sci_t{ descWords[4], descWords[5], descWords[6], descWords[7] };
    desc.cj[0]      = descWords[8];
    desc.cj[1]      = descWords[9];
    desc.cj[2]      = descWords[10];
    desc.cj[3]      = descWords[11];
    desc.exclInd[0] = descWords[12];
    desc.exclInd[1] = descWords[13];
    const int  rangeBegin = desc.rangeBegin;
    const int  rangeEnd   = desc.rangeEnd;
    const bool hasWork    = inLaunch && (rangeBegin < rangeEnd);
#ifdef NBNXM_WAVE_TIMELINE
    const unsigned long long tlDesc = wall_clock64(); /* the start record has arrived */
#endif
    int        sciIdx     = desc.sciIdx;

    float* __restrict__ f   = reinterpret_cast<float*>(atdat.f);
    const float rcoulomb_sq = nbp.rcoulomb_sq;
    const __amdgpu_buffer_rsrc_t fRsrc =
            __builtin_amdgcn_make_buffer_rsrc(f, 0, atdat.numAtoms * 3 * static_cast<int>(sizeof(float)), 0x00020000);

    /* ---- j-side pipeline -----------------------------------------------------------------------------------
     * Measured on MI355X: with the list words and j-atom data loaded where they are used, the loop skeleton
     * without any pair arithmetic took half of the kernel's time (5 waves per SIMD cannot cover ~1 us loads, and
     * every scalar load in flight turns the next LDS wait into a full lgkmcnt(0)).  So nothing in the group
     * loop is loaded through registers.  With LDS-direct loads (global -> LDS, no VGPRs, counted by vmcnt),
     * while group g is computed:
     *   W(g+3): the list words of group g+3 (32 bytes; FUSED: + its mask of perturbed cluster pairs) go to a ring
     *           of 4 records (three groups ahead: W(g+1) was issued before J(g-1), whose wait in the previous
     *           iteration therefore covers it, and the words of g+1 can be read without a wait of their own), and
     *   J(g+1): the j-side of group g+1 (32 x float4 xq, 32 types or LJ parameters, 64 exclusion words), addressed
     *           with the words of g+1 read back from the ring, goes to the other one of two staging buffers.
     * Loads and atomics retire through ONE in-order vmcnt counter, so every group issues exactly c_vmOpsPerGroup
     * VMEM operations (W + 3 J loads + 4 j-force atomics, dummies for skipped slots): the one wait of an iteration,
     * for J(g), is a counted vmcnt(c_vmOpsPerGroup) and never waits for an atomic.  The pipeline runs across
     * piece borders (the groups of a range are contiguous). */
#ifdef NBNXM_TIMING_NO_J_INSTR /* timing-only diagnostics build (tools/gpu_atomics.sh): results are wrong by construction */
    constexpr int  c_vmOpsPerGroup = (FUSED ? 2 : 1) + 3;
#else
    constexpr int  c_vmOpsPerGroup = (FUSED ? 2 : 1) + 3 + 4;
#endif
    const unsigned jStageLds = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>(jStage)));
    const unsigned ringLds   = jStageLds + 2U * c_jStageBytes;
    const unsigned char* ring = jStage + 2 * c_jStageBytes;
    const int      lastGroup = plist.ncjPacked - 1;
    const void*    ljBase    = USE_TABLE ? static_cast<const void*>(atomTypes) : static_cast<const void*>(ljComb);

/* W(g): list words of group g -> ring record g & 3 (lanes 0-7: the 8 dwords of nbnxn_cj_packed_t; FUSED: lane 8 adds
 * groupSlowMask[g] behind them).  The lane id is recomputed in place on purpose: hoisted out of the loops it would be
 * spilled, and a scratch reload in this loop is a VMEM load that drains the pipeline. */
#define NBNXM_STAGE_WORDS(g) NBNXM_STAGE_WORDS_L(g, laneIdNow())
#define NBNXM_STAGE_WORDS_L(g, laneExpr)                                                                        \
    {                                                                                                          \
        const unsigned lane = (laneExpr);                                                                      \
        const unsigned gw   = static_cast<unsigned>(min((g), lastGroup));                                      \
        const unsigned rec  = ringLds + (static_cast<unsigned>(g) & 3U) * c_ringRecordBytes;                   \
        if (lane < 8U) { ldsDirectLoad4(rec, gw * 32U + lane * 4U, cjPackedList); }                            \
        if constexpr (FUSED)                                                                                   \
        {                                                                                                      \
            if (lane == 8U) { ldsDirectLoad4(rec, gw * 4U, groupSlowMask); }                                   \
        }                                                                                                      \
    }
/* J(g): the three staging loads of group g into buffer buf; each lane reads the list words it needs from the ring */
#define NBNXM_STAGE_GROUP(g, buf) NBNXM_STAGE_GROUP_L(g, buf, laneIdNow())
#define NBNXM_STAGE_GROUP_L(g, buf, laneExpr)                                                                   \
    {                                                                                                          \
        const unsigned lane  = (laneExpr);                                                                     \
        const unsigned half  = lane >> 5;                                                                      \
        const unsigned char* rec = ring + (static_cast<unsigned>(g) & 3U) * c_ringRecordBytes;                 \
        const int      cjl   = *reinterpret_cast<const int*>(rec + ((lane >> 3) & 3U) * 4U);                   \
        const int      exl   = *reinterpret_cast<const int*>(rec + 20U + half * 8U);                           \
        NBNXM_STAGE_GROUP_LOADS(buf)                                                                           \
    }
/* the same for the range's first group, whose indices arrived with the range's start record (scalar registers) */
#define NBNXM_STAGE_GROUP_DESC(buf)                                                                            \
    {                                                                                                          \
        const unsigned lane  = laneIdNow();                                                                    \
        const unsigned half  = lane >> 5;                                                                      \
        const unsigned slot  = (lane >> 3) & 3U;                                                               \
        const int      cjl   = (slot == 0U) ? desc.cj[0] : ((slot == 1U) ? desc.cj[1] : ((slot == 2U) ? desc.cj[2] : desc.cj[3])); \
        const int      exl   = (half == 0U) ? desc.exclInd[0] : desc.exclInd[1];                               \
        NBNXM_STAGE_GROUP_LOADS(buf)                                                                           \
    }
#define NBNXM_STAGE_GROUP_LOADS(buf)                                                                           \
    {                                                                                                          \
        const unsigned ajl   = static_cast<unsigned>(cjl) * c_clSize + (lane & 7U);                            \
        const unsigned base  = jStageLds + static_cast<unsigned>(buf) * c_jStageBytes;                         \
        if constexpr (USE_TABLE)                                                                               \
        {                                                                                                      \
            if (lane < 32U)                                                                                    \
            {                                                                                                  \
                ldsDirectLoad16(base, ajl * 16U, xq);                                                          \
                ldsDirectLoad4(base + c_jStageLjOffset, ajl * 4U, ljBase);                                     \
            }                                                                                                  \
        }                                                                                                      \
        else                                                                                                   \
        {                                                                                                      \
            if (lane < 32U) { ldsDirectLoad16(base, ajl * 16U, xq); }                                          \
            /* lanes 0-31: c.x of the 32 atoms, lanes 32-63: c.y */                                            \
            ldsDirectLoad4(base + c_jStageLjOffset, ajl * 8U + half * 4U, ljBase);                             \
        }                                                                                                      \
        ldsDirectLoad4(base + c_jStageExclOffset, static_cast<unsigned>(exl) * 128U + (lane & 31U) * 4U, exclList); \
    }
#ifdef NBNXM_TIMING_NO_J_INSTR
#define NBNXM_DUMMY_ATOMIC()
#else
#define NBNXM_DUMMY_ATOMIC() __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(0.0F, fRsrc, c_dropLane, 0, 0)
#endif
#define NBNXM_WAIT_VMEM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

    /* progress thresholds of the wave priority in 1/16 of the range (see the group loop): late ones, because only the end
     * of the range decides how long the last wave of a SIMD runs alone (quarters measured 2 % slower) */
#ifndef NBNXM_PRIO_T1
#define NBNXM_PRIO_T1 8
#define NBNXM_PRIO_T2 12
#define NBNXM_PRIO_T3 14
#endif
    const int prioStep1 = rangeBegin + (((rangeEnd - rangeBegin) * NBNXM_PRIO_T1) >> 4);
    const int prioStep2 = rangeBegin + (((rangeEnd - rangeBegin) * NBNXM_PRIO_T2) >> 4);
    const int prioStep3 = rangeBegin + (((rangeEnd - rangeBegin) * NBNXM_PRIO_T3) >> 4);
    __builtin_amdgcn_s_setprio(3);
    int nextPrioStep = prioStep1;

    int curBuf      = 0;  /* staging buffer that holds (or receives) the j-side of group stagedGroup */
    int stagedGroup = -1;

    /* ---- the first piece: everything that depends on the start record, in one batch ------------------------------------ */
    nbnxn_sci_t nb_sci        = desc.entry;
    int         cjPackedBegin = desc.firstGroup;
    int         cjPackedEnd   = min(rangeEnd, nb_sci.cjPackedEnd);
    if (hasWork)
    {
        NBNXM_STAGE_WORDS(cjPackedBegin)
        NBNXM_STAGE_WORDS(cjPackedBegin + 1)
        NBNXM_STAGE_WORDS(cjPackedBegin + 2)
        NBNXM_STAGE_GROUP_DESC(curBuf)
        NBNXM_DUMMY_ATOMIC();
        NBNXM_DUMMY_ATOMIC();
        NBNXM_DUMMY_ATOMIC();
        NBNXM_DUMMY_ATOMIC();
        stagedGroup = cjPackedBegin;
    }

    /* ---- the i-atoms of a piece: the 8 atoms a lane meets stay in registers for the whole piece --------------------------- */
    float4 xqi[c_numClPerSupercl];
    int    trow[c_numClPerSupercl];  /* byte offset of row type_i of the LDS table (table flavours) */
    float2 ljcpi[c_numClPerSupercl]; /* combination-rule flavours */
/* The i-side of a piece.  Every lane needs atom tidxi of all 8 clusters — the same 8 atoms for the 8 lanes that share a tidxi.  Loaded
 * where they are used that is 8 dwordx4 + 8 dword instructions in which 64 lanes fetch 8 distinct addresses: 160 cycles of the CU's
 * address path per wave, and at the start of the kernel all 20 waves of a CU do it at once (measured with the prologue's time stamps:
 * 1.4 - 2.0 us between the start record and the arrival of the batch).  Instead lane l moves atom l, once: the 64 coordinates go to the
 * staging buffer the group loop is not using (LDS-direct, 1 KB), the 64 type rows arrive in one dword per lane, and the lanes pick up
 * their 8 atoms with 8 ds_read_b128 (same address for the lanes of a tidxi: a broadcast) and 8 ds_bpermute: 20 cycles of the address
 * path.  Two parts, so that the prologue can put the table staging between them:
 *   NBNXM_I_ATOMS_REQUEST  issues the loads (the buffer: the one the next J(g + 1) goes to, free until the group loop's first iteration);
 *   NBNXM_I_ATOMS_COLLECT  behind a vmcnt(0): registers xqi / trow / ljcpi; ends with lgkmcnt(0), because the group loop's first
 *                          staging step hands the buffer back to the memory pipeline. */
#define NBNXM_I_ATOMS_REQUEST                                                                                  \
    const unsigned iLane = laneIdNow();                                                                        \
    const int      iAtom = nb_sci.sci * c_superClSize + static_cast<int>(iLane);                               \
    ldsDirectLoad16(jStageLds + static_cast<unsigned>(curBuf ^ 1) * c_jStageBytes, static_cast<unsigned>(iAtom) * 16U, xq); \
    [[maybe_unused]] int    iTypeLane = 0;                                                                     \
    [[maybe_unused]] float2 iCombLane = make_float2(0.0F, 0.0F);                                               \
    if constexpr (USE_TABLE) { iTypeLane = atomTypes[iAtom]; }                                                 \
    else { iCombLane = ljComb[iAtom]; }                                                                        \
    const float3 iShift = atdat.shiftVec[nb_sci.shift & NBNXM_CI_SHIFT_MASK];
/* (LJ-PME: the per-type grid parameters come from the LDS copy, except ahead of the table staging: GRID_FROM_LDS false) */
#define NBNXM_I_ATOMS_COLLECT(GRID_FROM_LDS)                                                                   \
    {                                                                                                          \
        const unsigned char* iBuf  = jStage + (curBuf ^ 1) * c_jStageBytes + (iLane & 7U) * 16U;               \
        const int            iPick = static_cast<int>((iLane & 7U) << 2); /* byte address of lane tidxi for ds_bpermute */ \
        [[maybe_unused]] int iRowLane = 0;                                                                     \
        if constexpr (USE_TABLE && !LJ_EWALD) { iRowLane = numTypes * iTypeLane * static_cast<int>(sizeof(float2)) + rTabBytes; } \
        _Pragma("unroll") for (int i = 0; i < c_numClPerSupercl; i++)                                          \
        {                                                                                                      \
            float4 v = *reinterpret_cast<const float4*>(iBuf + i * (c_clSize * 16));                           \
            v.x += iShift.x;                                                                                   \
            v.y += iShift.y;                                                                                   \
            v.z += iShift.z;                                                                                   \
            v.w *= nbp.epsfac;                                                                                 \
            xqi[i] = v;                                                                                        \
            const int pick = iPick + i * (c_clSize * 4);                                                       \
            if constexpr (USE_TABLE && !LJ_EWALD) { trow[i] = __builtin_amdgcn_ds_bpermute(pick, iRowLane); }  \
            else if constexpr (USE_TABLE)                                                                      \
            {                                                                                                  \
                const int ti = __builtin_amdgcn_ds_bpermute(pick, iTypeLane);                                  \
                trow[i]      = numTypes * ti * static_cast<int>(sizeof(float2)) + rTabBytes;                   \
                ljcpi[i]     = (GRID_FROM_LDS) ? nbfpLds[numTypes * numTypes + ti] : nbp.nbfp_comb[ti];        \
            }                                                                                                  \
            else                                                                                               \
            {                                                                                                  \
                ljcpi[i].x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(pick, __builtin_bit_cast(int, iCombLane.x))); \
                ljcpi[i].y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(pick, __builtin_bit_cast(int, iCombLane.y))); \
            }                                                                                                  \
        }                                                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    }
    /* (waves without work load the i-atoms of the record's zeroed entry — super-cluster 0 —, and leave behind the barrier) */
    NBNXM_I_ATOMS_REQUEST

    /* ---- the tables, staged beside that batch ----------------------------------------------------------------------------- */
    if constexpr (EWALD_CORR_TABLE)
    {
        /* force flavours: {F, step} (16 KB); energy flavours: {F, step, V, step} (32 KB, they run at 4 waves per SIMD).
         * LDS-direct, 1 KB per instruction, the chunks dealt out over the waves of the workgroup: nothing to wait for here — a copy
         * through registers is a round trip per loop iteration (four of them at 16 KB and four waves), which the batch above has
         * made the longest chain of the prologue.  The table sits at LDS address 0 (checked below). */
        const void* src = ENERGY ? static_cast<const void*>(nbp.ewaldCorrTabFV) : static_cast<const void*>(nbp.ewaldCorrTab);
#ifdef NBNXM_TIMING_NO_TABLE_STAGE /* timing-only diagnostics build: what the 16 KB per workgroup cost at the start of the kernel (results wrong) */
        constexpr unsigned c_tabChunks = 1U;
#else
        constexpr unsigned c_tabChunks = static_cast<unsigned>(c_ewaldTabBytes) / 1024U;
#endif
        const unsigned laneT = laneIdNow();
        for (unsigned chunk = wave; chunk < c_tabChunks; chunk += blockSize / c_waveSize)
        {
            ldsDirectLoad16(chunk * 1024U, chunk * 1024U + laneT * 16U, src);
        }
    }
    if constexpr (USE_TABLE)
    {
        for (int t = threadIdx.x; t < numTypes * numTypes; t += blockSize) { nbfpLds[t] = nbp.nbfp[t]; }
        if constexpr (LJ_EWALD)
        {
            for (int t = threadIdx.x; t < numTypes; t += blockSize) { nbfpLds[numTypes * numTypes + t] = nbp.nbfp_comb[t]; }
        }
    }

    if constexpr (EWALD_R_TABLE)
    {
        float* dst = reinterpret_cast<float*>(nbLds + (EWALD_V_TABLE ? c_ewaldTabBytes : 0));
        for (int t = threadIdx.x; t < nbp.coulombTabSize; t += blockSize) { dst[t] = nbp.coulomb_tab[t]; }
    }
    if constexpr (EWALD_V_TABLE)
    {
        float2* dst = reinterpret_cast<float2*>(nbLds);
        for (int t = threadIdx.x; t < c_ewaldCorrTabSizeEnergy; t += blockSize)
        {
            const float4 fv = nbp.ewaldCorrTabFV[t];
            dst[t]          = make_float2(fv.z, fv.w);
        }
    }

    /* the table scale of ewaldTabAddress in a vector register, for the whole kernel (a scalar operand would halve the FMA's issue rate) */
    [[maybe_unused]] float ewaldTabScaleV = 0.0F;
    if constexpr (EWALD_CORR_TABLE)
    {
        asm volatile("v_mov_b32 %0, %1" : "=v"(ewaldTabScaleV) : "s"(ENERGY ? nbp.ewaldCorrTabScale16 : nbp.ewaldCorrTabScale8));
    }
    if constexpr (EWALD_V_TABLE) { asm volatile("v_mov_b32 %0, %1" : "=v"(ewaldTabScaleV) : "s"(nbp.ewaldCorrTabScale16)); }

    /* nbPair addresses the Ewald table with absolute LDS addresses from 0: the kernel has no static LDS, so the dynamic block starts there */
    if ((EWALD_CORR_TABLE || EWALD_R_TABLE) && reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)nbLds) != 0) { __builtin_trap(); }


#ifdef NBNXM_WAVE_TIMELINE
    const unsigned long long tlIssued = wall_clock64(); /* batch and table staging issued */
#endif
    NBNXM_WAIT_VMEM(0); /* the batch has arrived: list words, first group, i-atoms (and the table loads in front of it) */
#ifdef NBNXM_WAVE_TIMELINE
    const unsigned long long tlArrived = wall_clock64();
#endif
    NBNXM_I_ATOMS_COLLECT(false)
    __syncthreads();    /* the tables are in place; from here on the waves of the workgroup are independent */
    if (!hasWork) { return; }
#ifdef NBNXM_WAVE_TIMELINE
    const unsigned long long tlBarrier = wall_clock64();
#endif

    float E_lj = 0.0F, E_el = 0.0F;
    /* the fused energy block (nbPairEnergy) leaves the scale of E_lj and the three potential shifts to the end of the wave (or piece) */
    [[maybe_unused]] NbEnergySums eSums;
    auto finishEnergies = [&]() {
        if constexpr (ENERGY_HEADLINE)
        {
            E_lj += (eSums.ljTimes12 + fmaf(eSums.c12Masked, nbp.repulsion_shift.cpot, -2.0F * eSums.c6Masked * nbp.dispersion_shift.cpot)) * c_oneTwelfth;
            E_el += eSums.el;
            if constexpr (ELEC == ELK_EWALD_ANA) { E_el = fmaf(-nbp.sh_ewald, eSums.qqMasked, E_el); }
            eSums = NbEnergySums();
        }
    };

    /* ---- the pieces of this wave's range: one per i-entry it touches ---------------------------------- */
#pragma unroll 1
    for (;;)
    {
    const int  sci      = nb_sci.sci;
    const int  shiftIdx = nb_sci.shift & NBNXM_CI_SHIFT_MASK;
    const bool central  = (shiftIdx == c_centralShiftIndex);

    if constexpr (ENERGY && EXCL_FORCES)
    {
        /* self terms on the diagonal entry (nbnxm_cuda_kernel.cuh:365-400); lane l owns atom l.  The
         * cluster's own j-cluster is the first one of the entry, so exactly one piece sees it.  (Perturbed atoms
         * carry q = 0 here; their lambda-dependent self term is nbnxmFepClusterKernel's.) */
        if (central && cjPackedList[cjPackedBegin].cj[0] == sci * c_numClPerSupercl)
        {
            const float coef = (ELEC == ELK_CUT || ELEC == ELK_RF) ? -0.5F * nbp.c_rf : -nbp.ewald_beta * c_oneOverSqrtPi;
            const float qi   = xq[sci * c_superClSize + static_cast<int>(lane)].w * nbp.epsfac;
            E_el += qi * qi / nbp.epsfac * coef;
            if constexpr (LJ_EWALD)
            {
                /* nbnxm_cuda_kernel.cuh:374-380: the grid part counts every atom's pair with itself */
                const float lje2 = nbp.ewaldcoeff_lj * nbp.ewaldcoeff_lj;
                const int   ti   = atomTypes[sci * c_superClSize + static_cast<int>(lane)];
                E_lj += nbfpLds[ti * (numTypes + 1)].x * 0.5F * c_oneSixth * (lje2 * lje2 * lje2 * c_oneSixth);
            }
        }
    }

    /* ---- main pass: plain pairs ---------------------------------------------------------------------- */
    float3 fci_buf[c_numClPerSupercl];
#pragma unroll
    for (int i = 0; i < c_numClPerSupercl; i++) { fci_buf[i] = make_float3(0.0F, 0.0F, 0.0F); }

#pragma unroll 1
    for (int jPacked = cjPackedBegin; jPacked < cjPackedEnd; jPacked++)
    {
        /* The SIMD's arbiter serves the oldest wave first: left alone, the 5 waves of a SIMD finish one after the
         * other and the last one runs alone, latency-bound, for the last ~10 % of the kernel (measured with a
         * per-wave timeline).  Each wave therefore lowers its own priority as it advances through its range, so
         * that the waves that are behind get the issue slots and all of them finish together. */
        /* (one compare per group: the next threshold; the three-way chain cost three compares and branches per group) */
        if (__builtin_expect(jPacked == nextPrioStep, 0))
        {
            if (jPacked == prioStep3) { __builtin_amdgcn_s_setprio(0); }
            else if (jPacked == prioStep2)
            {
                __builtin_amdgcn_s_setprio(1);
                nextPrioStep = prioStep3;
            }
            else
            {
                __builtin_amdgcn_s_setprio(2);
                nextPrioStep = (prioStep2 > prioStep1) ? prioStep2 : prioStep3;
            }
        }

        /* pipeline step (see above): W(g+3); J(g+1); wait for J(g).  One lane-id computation per iteration serves the two
         * staging steps and the slot loop (it is recomputed per iteration on purpose, see NBNXM_STAGE_WORDS) */
        const unsigned laneIter = laneIdNow();
        NBNXM_STAGE_WORDS_L(jPacked + 3, laneIter)
        NBNXM_STAGE_GROUP_L(jPacked + 1, curBuf ^ 1, laneIter)
        NBNXM_WAIT_VMEM(c_vmOpsPerGroup);

#ifdef NBNXM_WAVE_TIMELINE
        if (tlFirst == 0) { tlFirst = wall_clock64(); }
#endif
        /* this group's list words, from the ring (the same address for all lanes) to SGPRs */
        const unsigned char* rec  = ring + (static_cast<unsigned>(jPacked) & 3U) * c_ringRecordBytes;
        const nb_int4        recA = *reinterpret_cast<const nb_int4*>(rec);
        const nb_int4 curA = { __builtin_amdgcn_readfirstlane(recA.x), __builtin_amdgcn_readfirstlane(recA.y),
                               __builtin_amdgcn_readfirstlane(recA.z), __builtin_amdgcn_readfirstlane(recA.w) };
        unsigned imask = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const unsigned*>(rec + 16U));
        /* FUSED: the cluster pairs that touch a perturbed atom are nbnxmFepClusterKernel's */
        if constexpr (FUSED) { imask &= ~__builtin_amdgcn_readfirstlane(*reinterpret_cast<const unsigned*>(rec + 32U)); }
        const unsigned char* jData = jStage + curBuf * c_jStageBytes;
        {
            const unsigned laneG = laneIter;
            const unsigned wexcl = *reinterpret_cast<const unsigned*>(jData + c_jStageExclOffset + laneG * 4U);
#ifndef NBNXM_J_FORCE_VGPR_OFFSET
            /* The j-force add of a slot goes to f + cj * 96 bytes + (tidxj * 12 + tidxi * 4): the lane's part is the same for every
             * slot of the group (computed once per group, from the lane id of this iteration), the cluster's part is wave-uniform and
             * travels in the buffer instruction's SCALAR offset — no vector instruction per slot for the address (it took four, two of
             * them half rate).  Lanes tidxi >= 3 carry an offset beyond the buffer: the hardware range check looks at the vector
             * offset alone (raw buffers on gfx9: the scalar offset is not part of it), so they are dropped whatever the scalar offset. */
            /* (tidxj * 12 as shift-adds: a v_mul_lo_u32 is a quarter-rate instruction) */
            const unsigned tj        = laneG >> 3;
            const int      laneFjOff = ((laneG & 7U) < 3U) ? static_cast<int>((((tj << 1) + tj) + (laneG & 7U)) << 2) : c_dropLane;
#endif
#pragma unroll
            for (int jm = 0; jm < c_jGroupSize; jm++)
            {
                /* every slot ends in exactly one VMEM atomic: a skipped slot sends nothing (all lanes out of range) */
                float          fjv    = 0.0F;
                int            fjOff  = c_dropLane;
                [[maybe_unused]] int fjSoff = 0;
                const unsigned imaskJ = (imask >> (jm * c_numClPerSupercl)) & 0xFFU;
                if (imaskJ != 0U)
                {
                const unsigned wexclJ = wexcl >> (jm * c_numClPerSupercl);
                const int      cj     = (jm == 0) ? curA.x : ((jm == 1) ? curA.y : ((jm == 2) ? curA.z : curA.w));
                [[maybe_unused]] const int aj = cj * c_clSize + static_cast<int>(tidxj);
                /* (lane part and slot part of the staging addresses kept apart: the slot part is a constant of the unrolled loop and
                 * belongs in the ds_read's offset field, not in a vector add per slot) */
                const unsigned char* jLane16 = jData + (laneG >> 3) * 16U;
                const unsigned char* jLane4  = jData + c_jStageLjOffset + (laneG >> 3) * 4U;
                const float4   xqj    = *reinterpret_cast<const float4*>(jLane16 + static_cast<unsigned>(jm) * (c_clSize * 16U));
                int            typej  = 0;
                float2         ljcp_j = make_float2(0.0F, 0.0F);
                if constexpr (USE_TABLE)
                {
                    typej = *reinterpret_cast<const int*>(jLane4 + static_cast<unsigned>(jm) * (c_clSize * 4U));
                    if constexpr (LJ_EWALD) { ljcp_j = nbfpLds[numTypes * numTypes + typej]; }
                }
                else
                {
                    ljcp_j.x = *reinterpret_cast<const float*>(jLane4 + static_cast<unsigned>(jm) * (c_clSize * 4U));
                    ljcp_j.y = *reinterpret_cast<const float*>(jLane4 + 128U + static_cast<unsigned>(jm) * (c_clSize * 4U));
                }

                const unsigned fastMask = imaskJ;
                /* which i-cluster (if any) is this j-cluster itself on the central image */
                /* as one bit per i-cluster: a single scalar bit test per pair block */
                [[maybe_unused]] const unsigned diagBits = (central && (cj >> 3) == sci) ? (1U << (cj & 7)) : 0U;

                float3    fcj_buf    = make_float3(0.0F, 0.0F, 0.0F);
                const int typejBytes = typej * static_cast<int>(sizeof(float2));
                /* (A second instance of the loop without exclusion handling for groups with excl_ind 0 was tried:
                 * it saves ~6 of ~55 issue slots per pair step on 80 % of the groups but costs 20 VGPRs, i.e. a
                 * wave per SIMD, and measured 19 % slower on MI355X.) */
                NBNXM_PAIR_LOOP(true)

                /* j-force: sum over the 8 lanes of a j atom; lanes tidxi 0..2 carry x,y,z (96 contiguous bytes) */
                fjv   = reduceXyzOver8Lanes(fcj_buf, laneG);
#ifdef NBNXM_J_FORCE_VGPR_OFFSET /* round 2's form, kept for A/B runs */
                fjOff = (tidxi < 3U) ? (3 * aj + static_cast<int>(tidxi)) * static_cast<int>(sizeof(float)) : c_dropLane;
#else
                fjOff  = laneFjOff;
                fjSoff = cj * (c_clSize * 3 * static_cast<int>(sizeof(float)));
#endif
#if defined(NBNXM_TIMING_NO_J_ATOMIC) /* timing-only: the add keeps its instruction and its operands, every lane is dropped */
                asm volatile("" : "+v"(fjv), "+v"(fjOff));
                fjOff = c_dropLane;
#endif
                }
                /* (the add at the merge point of the two paths costs one register copy per slot — its offset operand is a phi; with an
                 * add in each path instead, the compiler restructures the slot loop and spills 19 values: tried, not kept) */
#ifdef NBNXM_TIMING_NO_J_INSTR
                asm volatile("" ::"v"(fjv), "v"(fjOff));
#else
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(fjv, fRsrc, fjOff, fjSoff, 0);
#endif
            }
        }
        curBuf ^= 1;
    }
    stagedGroup = cjPackedEnd;
#ifdef NBNXM_WAVE_TIMELINE
    const unsigned long long tlLoopEnd = wall_clock64();
#endif

    /* i-forces: reduce over tidxj; lane (tidxj, tidxi) keeps the sum of cluster tidxj, atom tidxi */
    float3 mine;
    mine.x = reduceOverTidxjTransposed(fci_buf[0].x, fci_buf[1].x, fci_buf[2].x, fci_buf[3].x, fci_buf[4].x, fci_buf[5].x, fci_buf[6].x,
                                       fci_buf[7].x, lane);
    mine.y = reduceOverTidxjTransposed(fci_buf[0].y, fci_buf[1].y, fci_buf[2].y, fci_buf[3].y, fci_buf[4].y, fci_buf[5].y, fci_buf[6].y,
                                       fci_buf[7].y, lane);
    mine.z = reduceOverTidxjTransposed(fci_buf[0].z, fci_buf[1].z, fci_buf[2].z, fci_buf[3].z, fci_buf[4].z, fci_buf[5].z, fci_buf[6].z,
                                       fci_buf[7].z, lane);
    {
        [[maybe_unused]] const int ai = sci * c_superClSize + static_cast<int>(lane);
#ifdef NBNXM_TIMING_NO_I_ATOMIC /* timing-only diagnostics build */
        asm volatile("" ::"v"(mine.x), "v"(mine.y), "v"(mine.z), "v"(ai));
#elif defined(NBNXM_I_FORCE_STRIDED) /* round 2's form, kept for A/B runs: three 64-lane adds at a 12-byte stride */
        atomicAdd(&f[3 * ai + 0], mine.x);
        atomicAdd(&f[3 * ai + 1], mine.y);
        atomicAdd(&f[3 * ai + 2], mine.z);
#else
        /* The super-cluster's 192 force floats are contiguous (768 bytes).  A float atomic leaves L2 as one memory-side request
         * per 64-byte line it touches (tools/ubench/atomic_shapes.hip: 13-14 ns per line and CU whatever the shape), and an add
         * at a 12-byte stride touches all twelve lines: 36 requests per piece.  Transposed through the staging buffer that the
         * group loop has just left (the other one may be receiving the next group), lane l adds floats l, l + 64 and l + 128:
         * three adds of 256 contiguous bytes, 12 requests.  DS operations of one wave execute in order: no wait between the
         * stores and the loads, the compiler barrier only keeps their program order. */
        const unsigned laneT = laneIdNow(); /* in place: an address hoisted out of the loops would cost the group loop a register */
        float*         tr    = reinterpret_cast<float*>(jStage + (curBuf ^ 1) * c_jStageBytes);
        tr[3U * laneT + 0U]  = mine.x;
        tr[3U * laneT + 1U]  = mine.y;
        tr[3U * laneT + 2U]  = mine.z;
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        const float o0 = tr[laneT];
        const float o1 = tr[laneT + 64U];
        const float o2 = tr[laneT + 128U];
        float*      fs = f + 3 * (sci * c_superClSize) + static_cast<int>(laneT);
        atomicAdd(fs, o0);
        atomicAdd(fs + 64, o1);
        atomicAdd(fs + 128, o2);
#endif
    }
    float3 fshiftAcc = mine; /* per-lane share of this entry's total i-force */

    if (bCalcFshiftIn && !central)
    {
        const float sx = waveSum(fshiftAcc.x);
        const float sy = waveSum(fshiftAcc.y);
        const float sz = waveSum(fshiftAcc.z);
        if (lane < 3U)
        {
            const float v = (lane == 0U) ? sx : ((lane == 1U) ? sy : sz);
            /* one of c_numFshiftSlots copies (nbnxm_hip_types.h): the same few shift indices from thousands of pieces */
            /* (switch flavours: the lane id in place — the offset hoisted out of the loops is what they spilled to scratch; the
             * other flavours keep the hoisted one: 0.2 us faster on the headline kernel, whose allocation has room for it) */
            const int laneS = (VDW == VDK_FSWITCH || VDW == VDK_PSWITCH) ? static_cast<int>(laneIdNow()) : static_cast<int>(lane);
            atomicAdd(reinterpret_cast<float*>(atdat.fShift) + (1 + (sci & (c_numFshiftSlots - 1))) * c_fshiftSlotStride + 3 * shiftIdx + laneS, v);
        }
    }
    if constexpr (ENERGY)
    {
        if (nbp.clustersPerWindow > 0)
        {
            /* batched lambda windows: a range can run across the border of two windows, so the energies go to the window of the
             * i-entry piece by piece (a wave has one or two pieces) */
            finishEnergies();
            const float sLj = waveSum(E_lj), sEl = waveSum(E_el);
            const float v   = (lane == 0U) ? sLj : sEl;
            if (lane < 2U)
            {
                atomicAdd(atdat.windowSlots + ((sci * c_numClPerSupercl) / nbp.clustersPerWindow) * atdat.windowSlotStride
                                  + (workItem & (c_numEnergySlots - 1)) * c_energySlotStride + static_cast<int>(lane),
                          v);
            }
            E_lj = E_el = 0.0F;
        }
    }

    /* ---- the next piece: the next i-entry that owns a group of the range ------------------------------------------------- */
    bool more = false;
    for (sciIdx++; sciIdx < plist.nsciWork; sciIdx++)
    {
        nb_sci = scalarLoadSci(sciList + sciIdx);
        if (nb_sci.cjPackedBegin >= rangeEnd) { break; }
        cjPackedBegin = max(rangeBegin, nb_sci.cjPackedBegin);
        cjPackedEnd   = min(rangeEnd, nb_sci.cjPackedEnd);
        if (cjPackedBegin < cjPackedEnd)
        {
            more = true;
            break;
        }
    }
    if (!more) { break; }
    if (stagedGroup != cjPackedBegin)
    {
        /* restart the pipeline behind groups that belong to no i-entry */
        NBNXM_STAGE_WORDS(cjPackedBegin)
        NBNXM_STAGE_WORDS(cjPackedBegin + 1)
        NBNXM_STAGE_WORDS(cjPackedBegin + 2)
        NBNXM_WAIT_VMEM(0);
        NBNXM_STAGE_GROUP(cjPackedBegin, curBuf)
        NBNXM_DUMMY_ATOMIC();
        NBNXM_DUMMY_ATOMIC();
        NBNXM_DUMMY_ATOMIC();
        NBNXM_DUMMY_ATOMIC();
    }
    {
#ifdef NBNXM_WAVE_TIMELINE
        const unsigned long long tlReq = wall_clock64();
#endif
        NBNXM_I_ATOMS_REQUEST
        NBNXM_WAIT_VMEM(0);
#ifdef NBNXM_WAVE_TIMELINE
        const unsigned long long tlGot = wall_clock64();
#endif
        NBNXM_I_ATOMS_COLLECT(true)
#ifdef NBNXM_WAVE_TIMELINE
        tlTransReduce += tlReq - tlLoopEnd;   /* i-force reduction, atomics, next entry */
        tlTransWait += tlGot - tlReq;         /* request -> everything older retired */
        tlTransCollect += wall_clock64() - tlGot;
        tlTransCount++;
#endif
    }
    } /* pieces */
#undef NBNXM_STAGE_GROUP
#undef NBNXM_STAGE_GROUP_DESC
#undef NBNXM_STAGE_GROUP_LOADS
#undef NBNXM_I_ATOMS_REQUEST
#undef NBNXM_I_ATOMS_COLLECT
#undef NBNXM_STAGE_WORDS
#undef NBNXM_STAGE_GROUP_L
#undef NBNXM_STAGE_WORDS_L
#undef NBNXM_WAIT_VMEM
#undef NBNXM_DUMMY_ATOMIC

#ifdef NBNXM_WAVE_TIMELINE
    if (lane == 0U && workItem < 16384 && plist.debugTimeline != nullptr)
    {
        unsigned long long* g_nbTimeline = plist.debugTimeline;
        unsigned hwId;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwId));
        unsigned xccId;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xccId));
        g_nbTimeline[4 * workItem + 0] = tlStart;
        g_nbTimeline[4 * workItem + 1] = tlFirst;
        g_nbTimeline[4 * workItem + 2] = wall_clock64();
        g_nbTimeline[4 * workItem + 3] = (static_cast<unsigned long long>(xccId) << 32) | hwId;
        g_nbTimeline[4 * 16384 + 4 * workItem + 0] = tlTop;
        g_nbTimeline[4 * 16384 + 4 * workItem + 1] = tlDesc;
        g_nbTimeline[4 * 16384 + 4 * workItem + 2] = tlIssued;
        g_nbTimeline[4 * 16384 + 4 * workItem + 3] = (tlArrived << 32) | (tlBarrier & 0xFFFFFFFFULL);
        g_nbTimeline[8 * 16384 + 4 * workItem + 0] = tlTransCount;
        g_nbTimeline[8 * 16384 + 4 * workItem + 1] = tlTransReduce;
        g_nbTimeline[8 * 16384 + 4 * workItem + 2] = tlTransWait;
        g_nbTimeline[8 * 16384 + 4 * workItem + 3] = tlTransCollect;
    }
#endif

    if constexpr (ENERGY)
    {
        finishEnergies();
        E_lj = waveSum(E_lj);
        E_el = waveSum(E_el);
        const int   slot = workItem & (c_numEnergySlots - 1);
        const float v    = (lane == 0U) ? E_lj : E_el;
        /* (with batched lambda windows the pieces have delivered everything: the sums are zero) */
        if (lane < 2U && nbp.clustersPerWindow == 0) { atomicAdd(atdat.energySlots + slot * c_energySlotStride + static_cast<int>(lane), v); }
    }
}

/* One wavefront per i-entry `blockIdx.x * numParts + part` (pruneEntry above).  First pass (haveFreshList): the groups of an entry
 * are a dependent chain per group (list word -> j-coordinates -> 32 pair checks), so an entry goes to gridDim.y waves of
 * plist.pruneGroupsPerWave groups each — waves beyond the entry's end leave at once: 63 -> 30 us for the 96k box's 60k groups. */
template<bool haveFreshList>
__launch_bounds__(c_waveSize) __global__
        void nbnxmPruneKernel(const NBAtomDataGpu atdat, const NBParamGpu nbp, const gpu_plist plist, const int numParts, const int part)
{
    const int entry = static_cast<int>(blockIdx.x) * numParts + part;
    if (entry >= plist.nsci) { return; }
    if constexpr (haveFreshList)
    {
        pruneEntry<true>(atdat, nbp, plist, entry, static_cast<int>(blockIdx.y) * plist.pruneGroupsPerWave, plist.pruneGroupsPerWave);
    }
    else { pruneEntry<false>(atdat, nbp, plist, entry); }
}

#endif
