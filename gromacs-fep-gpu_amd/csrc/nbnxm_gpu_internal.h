/*
 * The NbnxmGpu object behind the C ABI (include/nbnxm_hip.h), shared by the translation units that implement it:
 * nbnxm_gpu.hip (data management, launch, copy-back) and nbnxm_buffer_ops.hip (coordinate / force buffer operations,
 * halo packing).  Mirrors nbnxm/cuda/nbnxm_cuda_types.h:67-143.
 */
#ifndef NBNXM_GPU_INTERNAL_H
#define NBNXM_GPU_INTERNAL_H

#include <hip/hip_runtime.h>

#include "device_utils.h"
#include "nbnxm_hip.h"
#include "nbnxm_hip_types.h"

using namespace nbnxm_hip;

/* nbnxm/gpu_types_common.h:81-98, all pinned */
struct NBStagingData
{
    float* eLJ             = nullptr;
    float* eElec           = nullptr;
    float* dvdlLJ          = nullptr;
    float* dvdlElec        = nullptr;
    float* fShift          = nullptr; /* 45 x 3 */
    float* eLJForeign      = nullptr;
    float* eElecForeign    = nullptr;
    float* dvdlLJForeign   = nullptr;
    float* dvdlElecForeign = nullptr;
    /* MI355X: one pinned mirror of the device's scalar-output block (the pointers above, except fShift, point
     * into it) so that one D2H copy brings everything back; energySlots: see NBAtomDataGpu::energySlots */
    float* scalars     = nullptr;
    float* energySlots  = nullptr;
    float* foreignSlots = nullptr;
};

struct InteractionTimers
{
    GpuRegionTimer nb_k, fep_k, prune_k;
    bool           didPrune = false, didRollingPrune = false;
};

/* nbnxm/cuda/nbnxm_cuda_types.h:67-143 */
struct NbnxmGpu
{
    bool           bUseTwoStreams = false;
    NBAtomDataGpu* atdat          = nullptr;
    NBParamGpu*    nbparam        = nullptr;
    gpu_plist*     plist[2]       = { nullptr, nullptr };
    gpu_feplist*   feplist[2]     = { nullptr, nullptr };
    NBStagingData  nbst;
    DeviceStream   deviceStreams[2];
    /* MI355X extension: the atom-pair FEP kernels of a locality run on their own stream, concurrently with
     * the cluster-pair kernel (forked / joined with events inside gpu_launch_kernel; both only += into f) */
    DeviceStream   fepStreams[2];
    hipEvent_t     fepFork[2]                  = { nullptr, nullptr };
    hipEvent_t     fepJoin[2]                  = { nullptr, nullptr };
    bool           fepConcurrent               = true;  /* split mode: atom-pair kernels on the FEP stream */
    bool           fepMergedFused              = true;  /* fused mode, force-only steps: the perturbed cluster pairs run in trailing
                                                          * workgroups of the cluster kernel (NBNXM_HIP_FEP_MERGED=0: own kernel) */
    /* Double-buffered force array: the force-only cluster kernel zeroes the buffer that is NOT in use in its last trailing
     * workgroups, and nbnxm_gpu_clear_outputs swaps the two instead of launching a clear kernel.  Switched off for good as soon
     * as the caller asks for the device pointer (nbnxm_gpu_get_f: it may keep it), or with NBNXM_HIP_F_DOUBLE_BUFFER=0. */
    bool           fDoubleBuffer               = true;
    float3*        fSpare                      = nullptr;
    int            fSpareAlloc                 = 0;
    bool           fSpareCleared               = false;
    /* ... and the same for the small outputs: two copies of [scalar-output block | shift-force block]; nbnxm_gpu_clear_outputs swaps to the
     * one a kernel's trailing workgroups have zeroed instead of launching a kernel for 80 KB (2.5 us of an energy step of the 96k box) */
    float*         outputsBlock[2]             = { nullptr, nullptr };
    int            outputsActive               = 0;
    bool           outputsDoubleBuffer         = true;  /* false once a caller holds a pointer into the block (nbnxm_gpu_get_fshift) */
    bool           outputsSpareCleared         = false;
    bool           fshiftDirty                 = false; /* a launch with shift forces since the last clear */
    bool           scalarsDirty                = true;  /* energies / dV/dl / foreign / window slots written since the last clear */
    int            energyTail                  = c_energyTailCompiled; /* see NBNXM_ENERGY_TAIL (nbnxm_hip_types.h) */
    bool           fepListMerged               = true;  /* atom-pair list mode: the list's force / energy kernel rides in trailing
                                                           workgroups of the cluster kernel (NBNXM_HIP_FEP_LIST_MERGED=0: own kernel) */
    bool           pruneMerged                 = true;  /* rolling pruning rides in trailing workgroups of the next force-only
                                                          * cluster kernel (NBNXM_HIP_PRUNE_MERGED=0: own kernel, at once) */
    bool           fepConcurrentFused          = false; /* fused mode: perturbed-cluster-pair kernel on the FEP stream */
    bool           fepBehindFused              = false; /* ... and launched behind the cluster kernel (energy / dH/dlambda steps) */
    hipEvent_t     nonlocal_done               = nullptr;
    hipEvent_t     misc_ops_and_local_H2D_done = nullptr;
    hipEvent_t     nonlocalKernelDone          = nullptr; /* after the last non-local cluster kernel (double-buffered forces) */
    bool           nonlocalKernelRecorded      = false;
    bool           haveWork[2]                 = { false, false };

    bool                bDoTime = false;
    InteractionTimers   timers[2];
    nbnxm_gpu_timings_t timings{};

    int  n_lambda  = 0;
    bool fusedFep  = false;
    int  numCUs    = 256;

    int nbWavesPerBlock = c_nbWavesPerBlock; /* tunable: NBNXM_HIP_WAVES_PER_BLOCK = 1..4 */
    /* nbnxm_gpu_set_kernel_routing: the caller's pick of a tabulated-Ewald / combination-rule kernel is kept instead of running the
     * faster equivalent (diagnostics: NBNXM_HIP_KEEP_TAB_KERNELS / NBNXM_HIP_KEEP_COMB_KERNELS set the initial values) */
    bool keepCombinationKernels = false;
    bool keepTabulatedKernels   = false;
    nbnxm_interaction_params_t callerParams{}; /* the interaction parameters as the caller last passed them (the kernel pick is re-derived from them) */
    /* domain decomposition: the local force-only launch in two parts (nbnxm_gpu_set_local_launch_parts), the second one behind the
     * non-local kernel so that it runs beside the force halo */
    int   localLaunchParts    = 1;
    float localPartFraction   = 0.65F;
    int   launchPartNow       = 0; /* 0: whole launch; 1 / 2: the part nbnxm_gpu_launch_kernel_part asked for */
    /* nbnxm_gpu_set_merged_localities: the local and the non-local list of a domain as one device list (local entries first) */
    bool                           mergedLocalities     = false;
    int                            numMergedLocalGroups = 0; /* packed j-groups of the local part */
    bool                           mergeLocalIsFresh    = false; /* the stashed local list is this search's (not yet merged) */
    std::vector<nbnxn_sci_t>       mergeLocalSci;
    std::vector<nbnxn_cj_packed_t> mergeLocalCj;
    std::vector<nbnxn_excl_t>      mergeLocalExcl;
    bool debugLaunchShape       = false;     /* diagnostics: NBNXM_HIP_DEBUG_LAUNCH_SHAPE prints the first launch's workgroup shape */
    /* work partition (gpu_plist::work*): SIMDs of the device, smallest range worth a wave (NBNXM_HIP_MIN_GROUPS_PER_WAVE) */
    int numSimds          = 1024;
    int minGroupsPerWave  = 1; /* (3k-atom box: 0.0156 -> 0.0133 ms per step with 1 instead of 2; larger boxes have more groups than wave slots anyway) */
    /* share of work per age class of the waves of a SIMD, [0]: 4 waves per SIMD, [1]: 5 (see WorkPartitionOut) */
    int waveClassShare[2][5] = { { 1024, 1024, 1024, 1024, 0 }, { 1100, 1060, 1024, 990, 946 } };
    /* ... of LONG ranges (from c_longRangeGroups packed groups per range on; 768k atoms and up): the older waves of a SIMD are favoured at a
     * RATE, so the longer the ranges, the more of the kernel's end is the youngest wave running alone — measured optimum at 1.02 M atoms
     * (round 4: force step 0.4177 -> 0.4078 ms, energy step 0.552 -> 0.533 ms); between c_shortRangeGroups and c_longRangeGroups the
     * shares are interpolated */
    int  waveClassShareLong[2][5] = { { 1225, 1110, 970, 791, 0 }, { 1350, 1185, 1024, 865, 696 } };
    /* ... of a SHORT list's force partition (four ranges per SIMD; 24k atoms: 18.9 -> 18.6 us; steeper ones lose); [0] = 0: equal shares */
    int  waveClassShareShort[4]   = { 1200, 1100, 980, 816 };
    bool waveClassShareFixed[2]   = { false, false }; /* set from the environment (diagnostics): no interpolation */
    int numWorkRangesOverride = 0; /* experiments: NBNXM_HIP_NUM_WORK_RANGES */
    int numWorkRangesEnergy   = 0; /* ranges of the energy flavours' partition (0: one per wave slot) */
    int workWeightsOverride[3] = { -1, -1, -1 }; /* experiments: NBNXM_HIP_WORK_WEIGHTS=slot,group,entry (relative to 8 per cluster pair) */
    PinnedBuffer<nbnxn_sci_t> h_sciSorted;
    PinnedBuffer<int>         h_slowCount;        /* one per locality */
    int*                      h_listError = nullptr; /* mapped host memory: nbnxmValidateListKernel's flag, and its device address */
    int*                      d_listError = nullptr;
    hipEvent_t                listStagingFree = nullptr; /* behind the last copy out of the list staging buffers (uploadPairlist) */
    bool                      listStagingBusy = false;
    hipEvent_t                slowCountReady[2] = { nullptr, nullptr }; /* behind the copy of gpu_plist::slowCount to h_slowCount */

    /* coordinate / force buffer operations (nbnxm_buffer_ops.hip; nbnxm_cuda_types.h:131-142) */
    int* atomIndices        = nullptr; /* grid slot -> atom index, -1 for fillers */
    int  atomIndicesSize    = 0;
    int  atomIndices_nalloc = 0;
    int* cell               = nullptr; /* atom index -> grid slot (force reduction) */
    int  cell_nalloc        = 0;
    int  reductionNumAtoms  = 0, reductionAtomStart = 0;
    bool reductionAccumulate = false;
    PinnedBuffer<int> h_atomIndices, h_cell;
    PinnedBuffer<float2>      h_ewaldCorrTab;
    PinnedBuffer<float4>      h_ewaldCorrTabFV;

    float* scalarOutputs    = nullptr; /* device block behind atdat->eLJ ... dvdlElecForeign, energySlots */
    /* batched lambda windows: device accumulators (atdat->windowSlots), pinned mirror, per-window sums of the last energy step */
    int                 numWindows = 0;
    PinnedBuffer<float> h_windowSlots;
    std::vector<double> windowSums; /* per window: e_lj, e_el, dvdl_lj, dvdl_el, then 4 x (n_lambda + 1) foreign terms */
    int    numHeadScalars   = 0;       /* scalars + foreign arrays */
    int    slotOffset       = 0;       /* first float of the energy slots */
    int    foreignSlotOffset = 0;      /* first float of the foreign-lambda slots */
    int    foreignSlotStride = 0;
    int    numScalarOutputs = 0;

    /* allocation bookkeeping */
    int xq_nalloc = 0, f_nalloc = 0, fep_nalloc = 0, fepBits_nalloc = 0;
    int nbfp_n = 0, nbfp_comb_n = 0, coulomb_tab_n = 0;
    int iinr_nalloc = 0, jindex_nalloc = 0, shiftIdx_nalloc = 0, jjnr_nalloc = 0, exclFep_nalloc = 0;

    /* pinned staging for every asynchronous upload: stays alive until the next upload of the same
     * kind (A.4: the reference frees its temporaries right after queuing the copies) */
    PinnedBuffer<float4> h_q4;
    PinnedBuffer<int4>   h_atomTypes4;
    std::vector<unsigned long long> sciSortKeys;
    std::vector<int>         sciSortRuns;
    std::vector<nbnxn_sci_t> sciWorkHost;      /* uploadPairlist: the entries sorted and joined, before they go to the pinned staging buffer */
    std::vector<int>     fepInverse;           /* gpu_init_feppairlist: topology id -> grid index */
    PinnedBuffer<int4>   h_clItem;             /* gpu_feplist::clItem / clListed / clIncl staging */
    PinnedBuffer<uint2>  h_clListed, h_clIncl;
    PinnedBuffer<int>    h_atomTypes, h_iinr, h_jjnr, h_shift, h_jindex, h_exclFep, h_pairEntry;
    PinnedBuffer<float2> h_ljComb;
    PinnedBuffer<float4> h_xq;
    PinnedBuffer<float>  h_f;
    PinnedBuffer<nbnxn_sci_t>       h_sci;
    PinnedBuffer<nbnxn_cj_packed_t> h_cjPacked;
    PinnedBuffer<nbnxn_excl_t>      h_excl;
    PinnedBuffer<unsigned char>     h_fepBits;
    PinnedBuffer<float>             h_shiftVec;
};

#endif
