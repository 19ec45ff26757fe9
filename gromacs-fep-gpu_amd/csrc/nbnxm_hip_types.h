/*
 * Device-visible data contract of the MI355X Nbnxm path.
 *
 * NBAtomDataGpu / NBParamGpu / gpu_plist / gpu_feplist keep the members of the reference's structs
 * (nbnxm/gpu_types_common.h:103-356) so that a maintainer can map them one to one; texture objects
 * are dropped (plain global / LDS loads on CDNA4) and a few MI355X-only members are appended at the end
 * of each struct.  Kernels receive the first three by value, as in the reference
 * (nbnxm/cuda/nbnxm_cuda.cu:142-144).
 */
#ifndef NBNXM_HIP_TYPES_H
#define NBNXM_HIP_TYPES_H

#include <hip/hip_runtime.h>

#include "nbnxm_hip.h"

constexpr int c_clSize            = NBNXM_GPU_CLUSTER_SIZE;
constexpr int c_numClPerSupercl   = NBNXM_GPU_NUM_CLUSTER_PER_SUPERCLUSTER;
constexpr int c_jGroupSize        = NBNXM_GPU_JGROUP_SIZE;
constexpr int c_superClSize       = c_clSize * c_numClPerSupercl;
constexpr int c_centralShiftIndex = NBNXM_CENTRAL_SHIFT_INDEX;
constexpr int c_numShiftVectors   = NBNXM_NUM_SHIFT_VECTORS;
constexpr int c_waveSize          = 64;
/* launch geometry of the cluster-pair kernel: one wavefront per i-entry, 4 wavefronts per workgroup */
constexpr int c_nbWavesPerBlock   = 4;
constexpr int c_nbBlockSize       = c_nbWavesPerBlock * c_waveSize;
/* the launch may use workgroups of 8 or 16 waves that share one copy of the LDS tables (many atom types, see nbnxm_gpu_launch_kernel);
 * the bound changes nothing in the generated code (the register budget comes from amdgpu_waves_per_eu) */
constexpr int c_nbMaxBlockSize    = 1024;
constexpr int c_simdsPerCu        = 4;
constexpr int c_ldsBytesPerCu     = 160 * 1024;
constexpr int c_maxTypesAtFullOccupancy = 28; /* 8 ntype^2 bytes + 16 KB Ewald table + 9 KB staging, five times in 160 KB */
constexpr int c_ldsAllocGranularity = 1280; /* LDS allocation granule of gfx950, bytes (measured: five workgroups of 31,872 B are resident per CU, five of 32,336 B are not) */
/* nbnxm/pairlist.h:166: keeps r^-12 finite in fp32 */
constexpr float c_nbnxnMinDistanceSquared = 3.82e-07F;
/* The cluster kernel's sum of squares starts from this (nm^2): invisible to any real pair distance in fp32, finite 1/r and r^-6 at r = 0
 * (filler atoms parked on one point, atomdata.cpp:148-184); see nbPair */
constexpr float c_r2Floor = 1.0e-12F;
/* nb_free_energy.cpp:107: cap on r^-6 in the perturbed-pair math */
constexpr float c_maxRInvSix = 1.0e15F;
constexpr int c_numEnergySlots   = 128;
/* Shift forces of the cluster kernel: most i-entries off the central image use one of six shifts, and thousands of adds to the
 * same few addresses serialise in one L2 channel (measured: +5.5 us on a virial step).  The kernel adds to one of
 * c_numFshiftSlots copies behind the primary array (atdat.fShift + (1 + slot) * c_fshiftSlotStride floats, 1 KB apart);
 * the host sums them with the primary array, which the atom-pair, perturbed-pair and listed kernels keep using. */
constexpr int c_numFshiftSlots   = 32;
constexpr int c_fshiftSlotStride = 256;
constexpr int c_fshiftBlockFloats = (1 + c_numFshiftSlots) * c_fshiftSlotStride;
constexpr int c_numForeignSlots  = 64;
constexpr int c_energySlotStride = 32;

/* nbnxm/gpu_types_common.h:103-155 */
struct NBAtomDataGpu
{
    int numAtoms;
    int numAtomsLocal;
    int numAtomsAlloc;

    float4* xq;  /* x,y,z,q  (q of perturbed atoms is 0) */
    float4* q4;  /* .x = qA, .y = qB  (FEP only) */
    float3* f;   /* force output, accumulated with atomics */

    float* eLJ;
    float* eElec;
    float* dvdlLJ;
    float* dvdlElec;
    float* eLJForeign;      /* n_lambda + 1 */
    float* eElecForeign;    /* n_lambda + 1 */
    float* dvdlLJForeign;   /* n_lambda + 1 */
    float* dvdlElecForeign; /* n_lambda + 1 */
    float3* fShift;         /* 45 */

    int     numTypes;
    int*    atomTypes;  /* per atom (perturbed atoms: numTypes-1) */
    float2* ljComb;     /* per atom, combination-rule kernels */
    int4*   atomTypes4; /* .x = typeA, .y = typeB (FEP only) */
    float4* ljComb4;    /* reference member, kept for the layout; never allocated: A/B LJ parameters come from nbfp[typeA/B] */

    float3* shiftVec; /* 45 */
    bool    shiftVecUploaded;

    /* MI355X extension: Grid::fepBits per 8-atom cluster (fused kernel), 8 bytes per super-cluster */
    unsigned char* fepBits;
    int            numClusters;

    /* MI355X extension: the cluster-pair kernel spreads its energy / dV/dl atomics over c_numEnergySlots
     * accumulators of c_energySlotStride floats ([eLJ, eElec, dvdlLJ, dvdlElec, pad...], one 128-byte line each):
     * thousands of waves adding to ONE address serialise in L2 at ~10 ns per atomic (measured: +0.25 ms per
     * energy step at 96k atoms).  The slots are summed on the host with the staged scalars (gpu_try_finish_task). */
    float* energySlots;
    /* ... and the same for the foreign-lambda terms of nbnxmFepClusterKernel: c_numForeignSlots accumulators of
     * foreignSlotStride floats, [eLJ[n+1], eElec[n+1], dvdlLJ[n+1], dvdlElec[n+1], pad] each */
    float* foreignSlots;
    int    foreignSlotStride;
    /* batched lambda windows (NBParamGpu::clustersPerWindow > 0): every window has its own energy and foreign-lambda
     * accumulators, windowSlots + window * windowSlotStride = [energy slots | foreign slots] in the layouts above */
    float* windowSlots;
    int    windowSlotStride;
    int    windowForeignOffset;
};

/* nbnxm/gpu_types_common.h:160-237 */
struct NBParamGpu
{
    int elecType; /* nbnxm_elec_type */
    int vdwType;  /* nbnxm_vdw_type */

    float epsfac;
    float c_rf;
    float two_k_rf;
    float ewald_beta;
    float sh_ewald;
    float sh_lj_ewald;
    float ewaldcoeff_lj;

    float rcoulomb_sq;
    float rvdw_sq;
    float rvdw_switch;
    float rlistOuter_sq;
    float rlistInner_sq;
    bool  useDynamicPruning;

    nbnxm_shift_consts_t  dispersion_shift;
    nbnxm_shift_consts_t  repulsion_shift;
    nbnxm_switch_consts_t vdw_switch;

    float2* nbfp;      /* numTypes^2: (6*C6, 12*C12) */
    float2* nbfp_comb; /* numTypes: LJ-PME grid parameters */

    float  coulomb_tab_scale;
    float* coulomb_tab;

    bool  bFEP;
    float alpha_coul;
    float alpha_vdw;
    int   lam_power;
    float sc_sigma6;
    float sc_sigma6_min;
    float lambda_q;
    float lambda_v;
    /* MI355X extension: the CPU kernel's second soft-core function (interaction_const_t::SoftCoreParameters,
     * mdtypes/interaction_const.h; nb_softcore.h); the reference's GPU kernels have Beutler only */
    int   softcoreType; /* nbnxm_softcore_type */
    float gapsysLinpointVdw;
    float gapsysLinpointCoul;
    float gapsysSigma6Vdw;
    float* allLambdaCoul; /* n_lambda */
    float* allLambdaVdw;  /* n_lambda */
    /* MI355X extension: several lambda windows of the same system batched into one object (R x N atoms, one list): the window of
     * an i-cluster is cluster / clustersPerWindow and its (lambda_q, lambda_v) come from this table; 0: one window, the scalars */
    const float2* windowLambda;
    int           clustersPerWindow;

    /* MI355X extension: rvdw (not squared) and rcoulomb for the per-interaction soft-core cut-offs
     * of the CPU kernel (nb_free_energy.cpp:804-812,880-890) */
    float rcoulomb;
    float rvdw;

    /* MI355X extension, analytical Ewald flavours: the real-space force correction beta^3 F((beta r)^2) on a
     * uniform grid in x = (beta r)^2 over [0, (beta rc)^2], c_ewaldCorrTabSize intervals, as the interval's line
     * {intercept, slope} in r^2 (value = a + b r^2: the FMA takes the r^2 the pair block holds, no fraction of the table
     * coordinate); the cluster kernel keeps it in LDS (one ds_read_b64 + one FMA per pair instead of a [5/4]
     * rational with a reciprocal, ~4 instead of ~14 issue slots).  Linear interpolation error <= 1.2e-6 relative,
     * the class of the rational fit (pme_corr_coeffs.h).  Entry k is the line over [(k - 1/16) h, (k + 15/16) h], h = r_c^2 / (size - 1): the
     * span the kernel's address arithmetic selects (ewaldTabAddress, nbnxm_device_helpers.h).  ewaldCorrTabScale8 / 16 = 8 or 16 bytes per entry / h. */
    float2* ewaldCorrTab;
    float   ewaldCorrTabScale8;
    float   ewaldCorrTabScale16;
    /* the same grid with the potential correction beside the force correction, for the energy flavours: {intercept, slope} of beta^3 F and of beta V, both in r^2,
     * with V(x) = erf(z)/z, x = z^2 (gmx::pmePotentialCorrection, simd/simd_math.h:1660-1760): one ds_read_b128 and two FMAs per pair
     * instead of two [5/4] rationals with a reciprocal each */
    float4* ewaldCorrTabFV;
    /* MI355X extension: 3 vdw_switch.c3, so that the potential-switch derivative needs no scalar product in the kernel */
    float   vdwSwitch3c3;
    /* MI355X extension, LJ-PME flavours: ewaldcoeff_lj^2 and ewaldcoeff_lj^6 / 6 (the kernel would compute them per launch in VGPRs) */
    float   ljEwaldCoeff2;
    float   ljEwaldCoeff6_6;
    /* MI355X extension, tabulated Ewald flavours: entries of coulomb_tab; the cluster kernel stages the table into LDS */
    int     coulombTabSize;
};

#ifndef NBNXM_EWALD_CORR_TAB_SIZE
#define NBNXM_EWALD_CORR_TAB_SIZE 2048
#endif
constexpr int c_ewaldCorrTabSize = NBNXM_EWALD_CORR_TAB_SIZE; /* a power of two (ewaldTabAddress masks with it) */
/* ... and of the energy flavours' table {F intercept, F slope, V intercept, V slope}: 30 KB instead of 32, so that FOUR workgroups of four
 * waves fit a CU's 160 KB with their staging areas (two of eight waves before): a trailing workgroup then starts when four of a CU's
 * range waves have retired, not eight (round 4).  The interpolation error grows by (2047 / 1919)^2. */
constexpr int c_ewaldCorrTabSizeEnergy = 1920;
constexpr int c_coulombTabMaxLds = 16384; /* entries of the r-indexed table that the tabulated flavours stage into LDS (64 KB) */
/* waves per workgroup of nbnxmFepClusterKernel */
constexpr int c_fepClusterWavesPerBlockDef = 4;
/* What the ENERGY flavours of the cluster kernel carry in trailing workgroups (the force flavours carry everything): 0 nothing,
 * 1 a pending rolling-prune part and the clear of the spare force buffer, 2 also the perturbed cluster pairs (fused mode; on
 * dH/dlambda steps with their energies at every foreign lambda, the FOREIGN flavour of fepClusterPair).  Compile-time ceiling;
 * NBNXM_HIP_ENERGY_TAIL (diagnostics) lowers it at run time. */
#ifndef NBNXM_ENERGY_TAIL
#define NBNXM_ENERGY_TAIL 2
#endif
constexpr int c_energyTailCompiled = NBNXM_ENERGY_TAIL;
constexpr unsigned c_clearFloat4PerThread = 4; /* trailing clear workgroups of the cluster kernel: float4 stores per thread */

/* MI355X extension: the start of one work range of the cluster kernel as ONE 64-byte record, so that a wave's prologue is one scalar
 * load followed directly by the i-atom loads and the first group's j-side loads (before: range borders -> i-entry -> list words ->
 * j data, four dependent round trips).  Written by nbnxmWorkDescKernel behind the range borders, whenever the partition changes.
 * Only what no pruning changes is copied: the i-entry record, the first group's j-cluster indices and exclusion-mask indices (the
 * group's imask — rewritten by every rolling-prune part — is read through the list-word ring like any other group's). */
/* A cluster pair that touches a perturbed atom (fused mode; gpu_plist::slowPairs), as the wave that evaluates it reads it — one 32-byte
 * scalar load: with only the pair's position in the list, the j-cluster and the exclusion indices were a dependent round trip more
 * before the first atom could be loaded, and these waves are nothing but a chain of round trips (round 4) */
struct NbSlowPair
{
    int entry;      /* group * 32 + jm * 8 + i */
    int sciShift;   /* its i-entry: sci * 64 + shift index */
    int cj;         /* cjPacked[group].cj[jm] */
    int exclInd[2]; /* cjPacked[group].imei[0 / 1].excl_ind */
    int pad[3];
};
static_assert(sizeof(NbSlowPair) == 32, "one s_load_dwordx8");

/* Work partitions of a list: 0 the energy flavours' (four waves per SIMD), 1 the force flavours' (five), 2 the energy flavours' on dH/dlambda
 * steps (four; round 4: on short lists fewer ranges than wave slots, so that the foreign-lambda work of the trailing workgroups — three
 * times the perturbed pairs' work of an energy step — runs beside the ranges from the start of the kernel) */
constexpr int c_numWorkPartitions = 3;
constexpr int c_partitionEnergy = 0, c_partitionForce = 1, c_partitionDhdl = 2;
inline int workPartitionWaves(int p) { return p == c_partitionForce ? 5 : 4; }

struct NbWorkDesc
{
    int         rangeBegin, rangeEnd; /* packed j-groups [rangeBegin, rangeEnd); empty when no i-entry owns a group of the range */
    int         sciIdx;               /* index into sciSorted of `entry` */
    int         firstGroup;           /* max(rangeBegin, entry.cjPackedBegin): the group whose indices follow */
    nbnxn_sci_t entry;                /* the first i-entry that owns a group of the range */
    int         cj[4];                /* cjPacked[firstGroup].cj */
    int         exclInd[2];           /* cjPacked[firstGroup].imei[0 / 1].excl_ind */
    int         pad[2];
};
static_assert(sizeof(NbWorkDesc) == 64, "one scalar load of 16 dwords");

/* nbnxm/gpu_types_common.h:297-341 */
struct gpu_plist
{
    int na_c;

    int          nsci;
    int          sci_nalloc;
    /* MI355X extension: entries of sciSorted, the cluster kernel's own i-entry list (see nbnxm_gpu_init_pairlist) */
    int          nsciWork;
    nbnxn_sci_t* sci;

    int                ncjPacked;
    int                cjPacked_nalloc;
    nbnxn_cj_packed_t* cjPacked;
    int                nimask;
    int                imask_nalloc;
    unsigned int*      imask; /* outer-pruned masks, 2 per packed group (rolling pruning source) */
    nbnxn_excl_t*      excl;
    int                nexcl;
    int                excl_nalloc;

    bool haveFreshList;
    /* first-pass prune (nbnxmPruneKernel<true>): every i-entry goes to pruneWavesPerEntry waves of pruneGroupsPerWave packed groups
     * (set by gpu_init_pairlist from the longest entry) */
    int  pruneGroupsPerWave;
    int  pruneWavesPerEntry;
    bool firstPruneDone; /* host bookkeeping: the first-pass prune of this list has run, imask[] holds the outer-pruned masks */
    int  rollingPruningNumParts;
    int  rollingPruningPart;
    /* host bookkeeping: a rolling-prune part that waits for the next force-only launch to run in its trailing workgroups */
    int  pendingPrunePart;    /* -1: none */
    int  pendingPruneEntries;
    int  pruneCallsSinceRebalance; /* rolling-prune calls since the work partition was last recomputed */

    /* MI355X extension: work partition of the cluster-pair kernel.  The launch has one wave per resident wave
     * slot of the device; wave w evaluates the packed j-groups [workRangeStart[p][w], workRangeStart[p][w+1]),
     * cut so that every wave gets the same weight (set imask bits, perturbed cluster pairs counted several
     * times), starting in i-entry workFirstSci[p][w] of sciSorted (the i-entries ordered by cjPackedBegin).
     * Two partitions: p = 0 for the 4-waves-per-SIMD flavours, p = 1 for the 5-waves-per-SIMD ones.
     * Recomputed on the device after every (re)prune of the list. */
    nbnxn_sci_t* sciSorted;
    int          sciSorted_nalloc;
    unsigned*    groupSlowMask;   /* ncjPacked: fused mode, the cluster pairs of each group that touch a perturbed atom */
    NbSlowPair*  slowPairs;       /* numSlowPairs: every listed cluster pair with a perturbed atom, with what its wave needs to start (NbSlowPair) */
    int*         slowCount;       /* [0] the number of entries of slowPairs, counted on the device; [1] how many of them, at the front, are heavy */
    int          numSlowHeavy;    /* ... and for how many of them, at the front, are heavy (split over several waves on dH/dlambda steps) */
    int          numSlowPairs;    /* the host's figure for sizing launches: exact once the count of this list has arrived, the previous
                                   * list's (plus a margin) until then — the kernels stride over *slowCount items either way */
    bool         slowCountPending; /* host bookkeeping: the copy of *slowCount for this list has been queued, not read yet */
    bool         slowCountKnown;   /* ... a count has been read at least once for this object */
    int          slowPairs_nalloc;
    bool         slowListDirty;   /* the list, fepBits or the mode changed since groupSlowMask / slowGroups were built */
    int*         groupWeight;     /* ncjPacked, scratch */
    int          groupWeight_nalloc;
    int*         weightBlockSum;  /* per 256 groups, then its exclusive scan; last entry = total */
    int          weightBlockSum_nalloc;
    int*         workRangeStart[c_numWorkPartitions];
    int*         workFirstSci[c_numWorkPartitions];
    NbWorkDesc*  workDesc[c_numWorkPartitions];       /* numWorkRanges: what a wave needs to start its range, in one 64-byte record (nbnxmWorkDescKernel) */
    int          workDesc_nalloc[c_numWorkPartitions];
    int          numWorkRanges[c_numWorkPartitions];
    int          work_nalloc[c_numWorkPartitions];
    int          workFirstSciAlloc[c_numWorkPartitions];
    float*       workShare[c_numWorkPartitions];      /* numWorkRanges: share of the total weight of each range, mean 1 (see WorkPartitionOut) */
    float*       workShareCum[c_numWorkPartitions];   /* numWorkRanges + 1: its running sum, normalised to 1 */
    int          workShareCount[c_numWorkPartitions]; /* the number of ranges the shares were set up for */
    int          workParts[c_numWorkPartitions];      /* 2: the ranges are two sets of one-per-wave-slot, for a launch in two parts (nbnxm_gpu_launch_kernel_part) */
    float        workPartFraction[c_numWorkPartitions]; /* ... and the first set's share of the weight the shares were set up for */
    int          workShareTaper16[c_numWorkPartitions]; /* ... and how long the ranges were (sixteenths between the short- and long-range class shares) */
    bool         workRangesDirty;
    unsigned long long* debugTimeline; /* diagnostics builds (NBNXM_WAVE_TIMELINE) only, else nullptr */
};

/* nbnxm/gpu_types_common.h:343-356; iinr/jjnr hold GRID-order atom indices on the device */
struct gpu_feplist
{
    int nri, maxnri;
    int nshift, maxnshift;
    int njidx, maxnjidx;
    int nrj, maxnrj;
    int nexcl, maxnexcl;

    int* iinr;
    int* shift;
    int* jindex;
    int* jjnr;
    int* excl_fep;

    /* MI355X extension: i-entry of every j slot, so that the kernel can flatten the ragged list
     * (one lane per pair, 64 consecutive pairs per wave) */
    int* pairEntry; /* nrj */
    int  pairEntry_nalloc;

    /* MI355X extension: the same list regrouped by cluster pair (built on the host by gpu_init_feppairlist): item k is the
     * (i-cluster, j-cluster, shift) combination clItem[k] = {ci, cj, shift index, 0} with the 8 x 8 bit masks — bit tidxj * 8 + tidxi,
     * i.e. the lane of the cluster kernel's layout — of the atom pairs the list holds (clListed) and of those among them that are not
     * topology exclusions (clIncl, excl_fep != 0).  One wavefront per item evaluates it with coalesced loads and the structured 8 x 8
     * reductions of the cluster kernels (fepListClusterItem); the flattened form above needs four levels of dependent loads and
     * segmented reductions. */
    int    numClusterItems;
    int4*  clItem;
    uint2* clListed;
    uint2* clIncl;
    int    clItem_nalloc;
};

/* Per-wave LDS staging of the j-side of one packed group (4 j-clusters):
 *   [0, 512)     float4 xq of the 32 atoms
 *   [512, 768)   table flavours: int type[32]; combination-rule flavours: float c.x[32], c.y[32]
 *   [768, 1024)  unsigned exclusion word of each of the 64 lanes
 * two such buffers per wave; FUSED adds the 64 i-atoms' A/B data (32 bytes per atom). */
constexpr int c_jStageBytes      = 1024;
constexpr int c_jStageLjOffset   = 512;
constexpr int c_jStageExclOffset = 768;
constexpr int c_iStageBytes      = c_superClSize * static_cast<int>(sizeof(float4) + sizeof(float2) + sizeof(int2));
/* ring of 4 list-word records per wave (32 bytes of nbnxn_cj_packed_t + 4 bytes of fepBits, padded) */
constexpr int c_ringRecordBytes  = 64;
constexpr int c_jRingBytes       = 4 * c_ringRecordBytes;
/* LDS of one wave for the foreign-lambda terms of a perturbed cluster pair (fepClusterPair): 8 values of up to 64 pairs and 64
 * accumulators — what a wave of the cluster kernel has as staging buffers and ring */
/* packed groups per work range below / above which the age-class shares of short / long ranges apply (NbnxmGpu::waveClassShare, ..Long) */
constexpr int c_shortRangeGroups = 12;  /* 96k atoms: 11.7 */
constexpr int c_longRangeGroups  = 27;  /* 192k atoms: 23 (measured optimum there: 0.8 of the way); 768k atoms: 109, 1.02 M atoms: 145 */
constexpr int c_fepForeignLdsBytes = (8 + 1) * 64 * static_cast<int>(sizeof(float));
/* a perturbed cluster pair with more perturbed atom pairs than this goes to the front of gpu_plist::slowPairs (nbnxmWorkWeightKernel) */
#ifndef NBNXM_SLOW_PAIR_HEAVY
#define NBNXM_SLOW_PAIR_HEAVY 32
#endif
constexpr int c_slowPairHeavy = NBNXM_SLOW_PAIR_HEAVY;
/* ... and on a dH/dlambda step is split over this many waves by lambda index (fepClusterPair) */
#ifndef NBNXM_FEP_FOREIGN_HEAVY_CHUNKS
#define NBNXM_FEP_FOREIGN_HEAVY_CHUNKS 3
#endif
constexpr int c_fepForeignHeavyChunks = NBNXM_FEP_FOREIGN_HEAVY_CHUNKS;
#ifndef NBNXM_FEP_FOREIGN_COMPACT_MAX_PAIRS
#define NBNXM_FEP_FOREIGN_COMPACT_MAX_PAIRS 40
#endif
constexpr int c_fepForeignCompactMaxPairs = NBNXM_FEP_FOREIGN_COMPACT_MAX_PAIRS; /* more pairs with a term in one cluster pair: index by index */
static_assert(c_fepForeignLdsBytes <= 2 * c_jStageBytes + c_jRingBytes, "the foreign-lambda scratch must fit a wave's staging area");

/* Dynamic LDS bytes of one workgroup of the cluster-pair kernel (must match the carve-up in the kernel). */
/* LDS bytes of the reference's r-indexed Ewald force table (tabulated flavours) */
__host__ __device__ inline int coulombTabLdsBytes(int coulombTabSize)
{
    return (coulombTabSize * static_cast<int>(sizeof(float)) + 15) & ~15;
}
/* ewaldTableBytes: the correction table of the analytical flavours (c_ewaldCorrTabSize float2) or coulombTabLdsBytes() or 0 */
inline int nbLdsBytes(int numTypes, bool useTable, bool ljEwald, int ewaldTableBytes, int wavesPerBlock)
{
    const int tableBytes = (useTable ? (((numTypes * numTypes + (ljEwald ? numTypes : 0)) * static_cast<int>(sizeof(float2)) + 15) & ~15) : 0)
                           + ewaldTableBytes;
    return tableBytes + wavesPerBlock * (2 * c_jStageBytes + c_jRingBytes);
}

#endif
