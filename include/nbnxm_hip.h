/*
 * nbnxm_hip.h — C ABI of the MI355X (gfx950) non-bonded FEP path.
 *
 * This is the drop-in boundary: every entry point replaces one function of the
 * reference's Nbnxm GPU API (C++, namespace Nbnxm, CUDA backend).  The reference
 * passes C++ objects (nbnxn_atomdata_t, interaction_const_t, NbnxnPairlistGpu,
 * t_nblist, gmx_enerdata_t); across this ABI the same data travels as plain
 * pointers + sizes, so that the mdrun side needs only a thin shim.
 *
 * Reference files are cited relative to /root/reference/src/gromacs/.
 *
 * Conventions kept from the reference (nbnxm/gpu_types_common.h):
 *  - all device work is stream-ordered on the per-locality stream;
 *  - outputs (f, fShift, energies, dV/dlambda, foreign terms) are ACCUMULATED by the
 *    kernels into buffers cleared by nbnxm_gpu_clear_outputs();
 *  - energies / dV/dl / fshift / foreign arrays are only downloaded for the Local locality;
 *  - errors are fatal (message on stderr + abort()), as GMX_RELEASE_ASSERT / gmx_fatal are.
 *
 * Pair-list layout: the reference's CUDA layout, i.e. 8-atom clusters, 8 clusters per
 * super-cluster, 4 j-clusters per packed group and c_nbnxnGpuClusterpairSplit = 2
 * (two {imask, excl_ind} per group, exclusion words for 4+4 j-atoms).  One 64-lane
 * wavefront covers the whole 8x(4+4) cluster pair: lanes 0-31 use imei[0], lanes 32-63 imei[1].
 */
#ifndef NBNXM_HIP_H
#define NBNXM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants (nbnxm/pairlistparams.h:58-98, nbnxm/pairlist.h:125,174-180) -------------- */
#define NBNXM_GPU_CLUSTER_SIZE 8
#define NBNXM_GPU_NUM_CLUSTER_PER_SUPERCLUSTER 8
#define NBNXM_GPU_JGROUP_SIZE 4
#define NBNXM_GPU_CLUSTERPAIR_SPLIT 2
#define NBNXM_GPU_EXCL_SIZE 32 /* c_nbnxnGpuClusterSize^2 / split */
#define NBNXM_CI_SHIFT_MASK 127
#define NBNXM_NUM_SHIFT_VECTORS 45     /* gmx::c_numShiftVectors, pbcutil/ishift.h */
#define NBNXM_CENTRAL_SHIFT_INDEX 22   /* gmx::c_centralShiftIndex */

/* Nbnxm::ElecType / Nbnxm::VdwType, nbnxm/nbnxm.h:181-214 (same numeric values) */
enum nbnxm_elec_type
{
    NBNXM_ELEC_CUT = 0,
    NBNXM_ELEC_RF,
    NBNXM_ELEC_EWALD_TAB,
    NBNXM_ELEC_EWALD_TAB_TWIN,
    NBNXM_ELEC_EWALD_ANA,
    NBNXM_ELEC_EWALD_ANA_TWIN,
    NBNXM_ELEC_COUNT
};
enum nbnxm_vdw_type
{
    NBNXM_VDW_CUT = 0,
    NBNXM_VDW_CUT_COMB_GEOM,
    NBNXM_VDW_CUT_COMB_LB,
    NBNXM_VDW_FSWITCH,
    NBNXM_VDW_PSWITCH,
    NBNXM_VDW_EWALD_GEOM,
    NBNXM_VDW_EWALD_LB,
    NBNXM_VDW_COUNT
};

/* gmx::InteractionLocality / gmx::AtomLocality, mdtypes/locality.h */
enum nbnxm_locality
{
    NBNXM_LOCAL = 0,
    NBNXM_NONLOCAL = 1
};

/* ---- pair-list element types (nbnxm/pairlist.h:198-280) ----------------------------------- */
/* A translation unit that already sees the reference's own definitions of these four structs (nbnxm/pairlist.h — the
 * integration shim) defines NBNXM_HIP_USE_REFERENCE_LIST_TYPES before including this header and passes the reference's
 * objects as they are; integration/nbnxm_hip_shim.cpp static_asserts that their sizes are the ones laid out here. */
#ifndef NBNXM_HIP_USE_REFERENCE_LIST_TYPES
typedef struct
{
    int sci;           /* i-super-cluster */
    int shift;         /* shift vector index (low 7 bits) + flags */
    int cjPackedBegin; /* first packed j-group */
    int cjPackedEnd;   /* one past the last packed j-group */
} nbnxn_sci_t;

typedef struct
{
    unsigned int imask;    /* bit (jm*8 + i): j-cluster jm of the group interacts with i-cluster i */
    int          excl_ind; /* index into the exclusion-mask array, 0 = shared "all interacting" */
} nbnxn_im_ei_t;

typedef struct
{
    int           cj[NBNXM_GPU_JGROUP_SIZE];
    nbnxn_im_ei_t imei[NBNXM_GPU_CLUSTERPAIR_SPLIT];
} nbnxn_cj_packed_t;

typedef struct
{
    /* word ((j & 3)*8 + i) of half (j >> 2); bit (jm*8 + ci): atom pair interacts */
    unsigned int pair[NBNXM_GPU_EXCL_SIZE];
} nbnxn_excl_t;
#endif /* NBNXM_HIP_USE_REFERENCE_LIST_TYPES */

/* ---- parameters ---------------------------------------------------------------------------- */
typedef struct { float c2, c3, cpot; } nbnxm_shift_consts_t;   /* mdtypes/interaction_const.h shift_consts_t */
typedef struct { float c3, c4, c5; } nbnxm_switch_consts_t;    /* switch_consts_t */

/* What gpu_init()/initNbparam()/set_cutoff_parameters() read from interaction_const_t and
 * PairlistParams (nbnxm/nbnxm_gpu_data_mgmt.cpp:201-223,421-489). */
typedef struct
{
    int   elecType; /* nbnxm_elec_type, already resolved as nbnxmGpuPickElectrostaticsKernelType does */
    int   vdwType;  /* nbnxm_vdw_type,  as nbnxmGpuPickVdwKernelType does */
    float epsfac;
    float c_rf;          /* reactionFieldShift */
    float k_rf;          /* reactionFieldCoefficient (two_k_rf = 2 k_rf is derived) */
    float ewaldcoeff_q;  /* ewald_beta */
    float sh_ewald;
    float sh_lj_ewald;
    float ewaldcoeff_lj;
    float rcoulomb;
    float rvdw;
    float rvdw_switch;
    float rlistOuter;
    float rlistInner;
    int   useDynamicPruning;
    nbnxm_shift_consts_t  dispersion_shift;
    nbnxm_shift_consts_t  repulsion_shift;
    nbnxm_switch_consts_t vdw_switch;
    /* Ewald force table (coulombEwaldTables->tableF); may be NULL/0 for non-tabulated types */
    float        coulomb_tab_scale;
    int          coulomb_tab_size;
    const float* coulomb_tab;
} nbnxm_interaction_params_t;

/* gmx::StepWorkload subset the path reads (mdtypes/simulation_workload.h) */
typedef struct
{
    int computeForces; /* always 1 in the reference's GPU path; kept for symmetry */
    int computeEnergy;
    int computeVirial;
    int computeDhdl;
    int useGpuFBufferOps;
} nbnxm_step_workload_t;

/* Caller-owned accumulation targets of gpu_wait_finish_task (gmx_enerdata_t subset,
 * nbnxm/gpu_common.h:405-435, mdtypes/enerdata.h:123-130). */
typedef struct
{
    double  e_lj;            /* grpp LJSR[0]      += */
    double  e_el;            /* grpp CoulombSR[0] += */
    double  dvdl_lin[2];     /* [0]=Coul, [1]=Vdw; used when !haveSoftCore */
    double  dvdl_nonlin[2];  /* used when haveSoftCore */
    int     n_lambda;        /* number of foreign lambdas */
    double* foreign_energies;   /* n_lambda+1: ForeignLambdaTerms::energies_ */
    double* foreign_dhdl_coul;  /* n_lambda+1: dhdl_[idx][Coul] */
    double* foreign_dhdl_vdw;   /* n_lambda+1: dhdl_[idx][Vdw]  */
} nbnxm_enerdata_t;

typedef struct NbnxmGpu NbnxmGpu; /* opaque; nbnxm/cuda/nbnxm_cuda_types.h:67-143 */

/* Kernel-time accounting (gmx_wallclock_gpu_nbnxn_t subset, timing/gpu_timing.h:85-93) */
typedef struct
{
    double nb_k_ms;    int nb_k_count;
    double fep_k_ms;   int fep_k_count;
    double prune_k_ms; int prune_k_count;
} nbnxm_gpu_timings_t;

/* ---- life cycle ---------------------------------------------------------------------------- */

/* Nbnxm::gpu_init — nbnxm/gpu_data_mgmt.h:88-96, nbnxm_gpu_data_mgmt.cpp:538-628.
 * nbfp: 2*numTypes^2 floats (6*C6, 12*C12); nbfp_comb: 2*numTypes floats or NULL.
 * localStream/nonLocalStream: hipStream_t to enqueue into, or NULL to let the module create its own
 * (DeviceStreamManager in the reference). */
NbnxmGpu* nbnxm_gpu_init(const nbnxm_interaction_params_t* ic, int numTypes, const float* nbfp,
                         const float* nbfp_comb, int bLocalAndNonlocal, int bFEP, int n_lambda,
                         void* localStream, void* nonLocalStream);

/* Nbnxm::gpu_free — nbnxm/gpu_data_mgmt.h:115, nbnxm_gpu_data_mgmt.cpp:1540-1654 */
void nbnxm_gpu_free(NbnxmGpu* nb);

/* Nbnxm::cuda_copy_fepparams — nbnxm/gpu_data_mgmt.h:73-85, nbnxm_gpu_data_mgmt.cpp:491-536.
 * all_lambda_coul/vdw: n_lambda doubles each (all_lambda[Coul], all_lambda[Vdw]). */
void nbnxm_gpu_copy_fepparams(NbnxmGpu* nb, int bFEP, float alpha_coul, float alpha_vdw,
                              int lam_power, float sc_sigma6_def, float sc_sigma6_min,
                              float lambda_q, float lambda_v, int n_lambda,
                              const double* all_lambda_coul, const double* all_lambda_vdw);

/* MI355X extension: keep the caller's kernel pick.  By default two picks of the reference run a faster equivalent kernel: a tabulated
 * Ewald pick (the reference's default on AMD devices, nbnxm_gpu_data_mgmt.cpp:120-145) runs the analytical Ewald kernels, and on
 * force-only steps a combination-rule LJ pick (nbnxm.h:196-205) runs the type-table kernel while the table is small — same terms, same
 * results within the parity bar.  keepTabulatedKernels / keepCombinationKernels != 0 run exactly
 * the kernels the caller picked.  Takes effect from the next launch; nbnxm_gpu_is_kernel_ewald_analytical reports what runs. */
void nbnxm_gpu_set_kernel_routing(NbnxmGpu* nb, int keepTabulatedKernels, int keepCombinationKernels);

/* MI355X extension: the soft-core function of the perturbed pairs.  The reference's GPU kernels implement Beutler only (the
 * caller falls back to the CPU for anything else); its CPU kernel also has Gapsys (SoftcoreType, mdtypes/md_enums.h;
 * interaction_const_t::SoftCoreParameters{softcoreType, gapsysScaleLinpointVdW, gapsysScaleLinpointCoul, gapsysSigma6VdW},
 * mdtypes/interaction_const.h; gmxlib/nonbonded/nb_softcore.h).  Beutler (the state after nbnxm_gpu_init) uses the alphas of
 * nbnxm_gpu_copy_fepparams and ignores the three Gapsys parameters; Gapsys ignores the alphas. */
enum nbnxm_softcore_type
{
    NBNXM_SOFTCORE_BEUTLER = 0,
    NBNXM_SOFTCORE_GAPSYS  = 1
};
void nbnxm_gpu_set_softcore(NbnxmGpu* nb, int softcoreType, float gapsysScaleLinpointVdW, float gapsysScaleLinpointCoul,
                            float gapsysSigma6VdW);

/* Nbnxm::gpu_pme_loadbal_update_param — nbnxm_gpu_data_mgmt.cpp (cut-off/Ewald update) */
void nbnxm_gpu_pme_loadbal_update_param(NbnxmGpu* nb, const nbnxm_interaction_params_t* ic);

/* ---- per search step ----------------------------------------------------------------------- */

/* Nbnxm::gpu_init_atomdata — nbnxm/gpu_data_mgmt.h:112-113, nbnxm_gpu_data_mgmt.cpp:873-1045.
 * Arrays are in nbnxm grid order, numAtoms long (nbat->params()):
 *   type / lj_comb(2 per atom)           : "normal" parameters, perturbed atoms zeroed by the caller
 *   qA,qB,typeA,typeB                    : A/B-state parameters (FEP only; may be NULL otherwise)
 *   lj_combA,lj_combB                    : accepted for the reference's call shape and NOT read: the perturbed pairs take the A/B
 *                                          c6 / c12 from the type-pair table nbfp[typeA], nbfp[typeB] in every flavour */
void nbnxm_gpu_init_atomdata(NbnxmGpu* nb, int numAtoms, int numAtomsLocal, const int* type,
                             const float* lj_comb, const float* qA, const float* qB,
                             const int* typeA, const int* typeB, const float* lj_combA,
                             const float* lj_combB);

/* Nbnxm::gpu_init_pairlist — nbnxm/gpu_data_mgmt.h:99-101, nbnxm_gpu_data_mgmt.cpp:667-759.
 * Arrays in page-locked host memory known to the HIP runtime (hipHostMalloc / hipHostRegister: what the reference's pinned HostVectors
 * are) are read in place and asynchronously, as the reference reads them (:706-735): keep them unchanged until the locality's stream
 * has passed the copies — the reference's caller keeps its lists for the whole search interval.  Arrays in any other memory are copied
 * before the call returns. */
void nbnxm_gpu_init_pairlist(NbnxmGpu* nb, int iloc, int na_c, int nsci, const nbnxn_sci_t* sci,
                             int ncjPacked, const nbnxn_cj_packed_t* cjPacked, int nexcl,
                             const nbnxn_excl_t* excl);

/* Nbnxm::gpu_init_feppairlist — nbnxm/gpu_data_mgmt.h:104-109, nbnxm_gpu_data_mgmt.cpp:761-871.
 * The t_nblist (mdtypes/nblist.h:40-54) holds TOPOLOGY atom ids; atomIndices is
 * gridSet.atomIndices() (grid index -> topology id, -1 for fillers), numAtomIndices long.
 * Pass atomIndices = NULL if iinr/jjnr are already in grid order. */
void nbnxm_gpu_init_feppairlist(NbnxmGpu* nb, int iloc, int nri, const int* iinr, const int* shift,
                                const int* jindex, int nrj, const int* jjnr, const int* excl_fep,
                                int numAtomIndices, const int* atomIndices);

/* MI355X extension ("shape B"): per-cluster perturbed-atom bits (Grid::fepBits,
 * nbnxm/grid.h:289-299), one byte per 8-atom cluster in grid order.  When set, the cluster-pair
 * kernel evaluates perturbed pairs in-line with the soft-core A/B math and the atom-pair
 * FEP list is not needed; the pair list must then keep the topology exclusion bits of
 * perturbed pairs (i.e. the host builder does not apply pairlist.cpp:1919). */
void nbnxm_gpu_init_fep_cluster_bits(NbnxmGpu* nb, int numClusters, const unsigned char* fepBits);

/* ---- per step ------------------------------------------------------------------------------ */

/* Nbnxm::gpu_upload_shiftvec — nbnxm/gpu_data_mgmt.h:123, 45 x 3 floats */
void nbnxm_gpu_upload_shiftvec(NbnxmGpu* nb, const float* shift_vec);

/* Nbnxm::gpu_copy_xq_to_gpu — nbnxm/nbnxm_gpu.h:93-96.  xq: 4 floats per atom, grid order, host. */
void nbnxm_gpu_copy_xq_to_gpu(NbnxmGpu* nb, const float* xq, int atomLocality);

/* Nbnxm::gpu_launch_kernel — nbnxm/nbnxm_gpu.h:108-111, nbnxm/cuda/nbnxm_cuda.cu:642-858 */
void nbnxm_gpu_launch_kernel(NbnxmGpu* nb, const nbnxm_step_workload_t* stepWork, int iloc);

/* Nbnxm::gpu_launch_kernel_pruneonly — nbnxm/nbnxm_gpu.h:148-151, nbnxm_cuda.cu:873-994.
 * First pruning of a fresh list: a kernel, at once.  Rolling pruning (numParts parts, one per call): the part is noted and runs in
 * trailing workgroups of the next force-only nbnxm_gpu_launch_kernel of the locality (in its own kernel ahead of any other
 * flavour, before the next prune call, and before nbnxm_gpu_debug_get_cjpacked hands out the masks); results of a step do not
 * depend on which of the two mask values its kernel reads.  NBNXM_HIP_PRUNE_MERGED=0: always at once. */
void nbnxm_gpu_launch_kernel_pruneonly(NbnxmGpu* nb, int iloc, int numParts);

/* Nbnxm::gpu_launch_cpyback — nbnxm/nbnxm_gpu.h:157-161, nbnxm_gpu_data_mgmt.cpp:1117-1303.
 * f_out: 3 floats per atom (nbat->out[0].f), host. */
void nbnxm_gpu_launch_cpyback(NbnxmGpu* nb, float* f_out, const nbnxm_step_workload_t* stepWork,
                              int atomLocality);

/* Nbnxm::gpu_try_finish_task (Check) / gpu_wait_finish_task (Wait) —
 * nbnxm/nbnxm_gpu.h:201-236, nbnxm/gpu_common.h:293-435.
 * shiftForces: 45 x 3 floats accumulated into (may be NULL when !computeVirial).
 * try returns 1 when the task had finished (and the reduction was done), else 0. */
int  nbnxm_gpu_try_finish_task(NbnxmGpu* nb, const nbnxm_step_workload_t* stepWork, int atomLocality,
                               int haveSoftCore, nbnxm_enerdata_t* enerd, float* shiftForces);
void nbnxm_gpu_wait_finish_task(NbnxmGpu* nb, const nbnxm_step_workload_t* stepWork,
                                int atomLocality, int haveSoftCore, nbnxm_enerdata_t* enerd,
                                float* shiftForces);

/* Nbnxm::gpu_clear_outputs — nbnxm/gpu_data_mgmt.h:126, nbnxm_gpu_data_mgmt.cpp:1047-1070.
 * (Unlike the reference, energies and dV/dl are cleared on every call that follows a launch which wrote them)
 * The forces are double-buffered inside the module: after a force-only launch this call swaps to a buffer that launch has already
 * zeroed and starts no kernel.  The device address of the forces therefore changes from step to step, until nbnxm_gpu_get_f is
 * called: from then on it stays fixed (and every call clears with a kernel). */
void nbnxm_gpu_clear_outputs(NbnxmGpu* nb, int computeVirial);

/* ---- queries / plumbing -------------------------------------------------------------------- */

/* Nbnxm::gpu_get_timings / gpu_reset_timings — nbnxm/gpu_data_mgmt.h:129-132 */
void nbnxm_gpu_get_timings(NbnxmGpu* nb, nbnxm_gpu_timings_t* out);
void nbnxm_gpu_reset_timings(NbnxmGpu* nb);
/* enable/disable hipEvent timing of the kernels (GMX_ENABLE_GPU_TIMING, gpu_utils/gpu_utils.cpp:60-75) */
void nbnxm_gpu_set_timing(NbnxmGpu* nb, int enable);

/* Nbnxm::gpu_min_ci_balanced — nbnxm/gpu_data_mgmt.h:135, cuda/nbnxm_cuda_data_mgmt.cu:82-109 */
int nbnxm_gpu_min_ci_balanced(NbnxmGpu* nb);
/* Nbnxm::gpu_is_kernel_ewald_analytical — nbnxm/gpu_data_mgmt.h:138 */
int nbnxm_gpu_is_kernel_ewald_analytical(const NbnxmGpu* nb);
/* Nbnxm::gpu_get_xq / gpu_get_f / gpuGetNBAtomData-style raw device pointers (nbnxm_gpu.h:238-311) */
void* nbnxm_gpu_get_xq(NbnxmGpu* nb);
void* nbnxm_gpu_get_f(NbnxmGpu* nb); /* also pins the force buffer, see nbnxm_gpu_clear_outputs */
/* (get_fshift: the primary array of 45 x 3 floats other kernels — listed forces — add to; the cluster kernel's own share sits in
 * accumulator slots behind it and joins in gpu_try/wait_finish_task) */
void* nbnxm_gpu_get_fshift(NbnxmGpu* nb);
/* gpuGetNBAtomData(nb)->q4: float4 per grid slot, .x = qA, .y = qB — what the perturbed 1-4 pairs of the listed forces read
 * (mdlib/sim_util.cpp:1678-1689) */
void* nbnxm_gpu_get_q4(NbnxmGpu* nb);
/* stream of a locality (hipStream_t), for callers that order their own work after the kernels */
void* nbnxm_gpu_get_stream(NbnxmGpu* nb, int iloc);
/* Nbnxm::haveGpuShortRangeWork — nbnxm_gpu.h:300-311 */
int nbnxm_gpu_have_short_range_work(const NbnxmGpu* nb, int iloc);

/* ---- coordinate / force buffer operations (GPU update, GPU halo exchange schedules) ----------------- */

/* Nbnxm::nbnxn_gpu_init_x_to_nbat_x — nbnxm/nbnxm_gpu.h:255-262, nbnxm_gpu_data_mgmt.cpp:1400-1500.
 * atomIndices = gridSet.atomIndices(): grid slot -> atom index, -1 for filler slots, one entry per grid slot
 * (numAtomIndices == numAtoms of nbnxm_gpu_init_atomdata).  Call after every search. */
void nbnxm_gpu_init_x_to_nbat_x(NbnxmGpu* nb, int numAtomIndices, const int* atomIndices);

/* Nbnxm::nbnxn_gpu_x_to_nbat_x — nbnxm/nbnxm_gpu.h:264-287, nbnxm_gpu_buffer_ops.cpp:62-98 and
 * cuda/nbnxm_gpu_buffer_ops_internal.cu:66-147.  d_x: device float3[] in atom order; [slotBegin, slotEnd): the
 * grid slots to convert (a grid's cellOffset*numAtomsPerCell .. its end); xReadyOnDevice: hipEvent_t the locality's
 * stream waits for first, or NULL.  Filler slots and the charges (xq.w) are left untouched. */
void nbnxm_gpu_x_to_nbat_x(NbnxmGpu* nb, const void* d_x, void* xReadyOnDevice, int atomLocality,
                           int slotBegin, int slotEnd, int mustInsertNonLocalDependency);

/* Nbnxm::nbnxnInsertNonlocalGpuDependency — nbnxm_gpu_data_mgmt.cpp:1305-1327 */
void nbnxm_gpu_insert_nonlocal_dependency(NbnxmGpu* nb, int iloc);

/* Nbnxm::setupGpuShortRangeWork — nbnxm_gpu_data_mgmt.cpp:1093-1104 */
void nbnxm_gpu_setup_short_range_work(NbnxmGpu* nb, int haveListedForcesGpuInteractions, int iloc);

/* gmx::GpuForceReduction::reinit / execute — mdlib/gpuforcereduction.h:96-118, gpuforcereduction_impl.cpp:97-190,
 * kernel mdlib/gpuforcereduction_impl_internal.cu:57-127: for atom i in [0, numAtoms):
 *   f[atomStart + i] = (accumulate ? f[atomStart + i] : 0) + nbnxmForce[cell[i]] (+ rvecForceToAdd[atomStart + i]).
 * cell = atom index -> grid slot (gridSet.cells()).  stream NULL: the local non-bonded stream. */
void nbnxm_gpu_force_reduction_reinit(NbnxmGpu* nb, int numAtoms, const int* cell, int atomStart, int accumulate);
void nbnxm_gpu_force_reduction_execute(NbnxmGpu* nb, void* d_baseForce, const void* d_rvecForceToAdd, void* stream);
/* A part of the atoms only — the reference keeps one GpuForceReduction object per locality (mdrun/runner.cpp: local atoms
 * [0, numHome), non-local atoms behind them) with its own atomStart / accumulate; here both use the one cell map uploaded
 * with nbnxm_gpu_force_reduction_reinit(nb, numAtomsAll, cell, 0, .):  f[i] = (accumulate ? f[i] : 0) + nbnxmForce[cell[i]]
 * for i in [atomBegin, atomEnd), on `stream` (NULL: the local non-bonded stream). */
void nbnxm_gpu_force_reduction_execute_range(NbnxmGpu* nb, void* d_baseForce, int atomBegin, int atomEnd, int accumulate, void* stream);

/* Pack / unpack kernels of gmx::GpuHaloExchange — domdec/gpuhaloexchange_impl_gpu.cu:62-116,118-183:
 *   pack:    sendBuf[i] = x[map[i]] (+ coordinateShift when not NULL)
 *   unpack:  f[map[i]] (+)= recvBuf[i]
 * Device pointers; stream is a hipStream_t.  The exchange itself: include/halo_hip.h (RCCL send / receive groups). */
void nbnxm_gpu_halo_pack_x(void* stream, const void* d_x, const int* d_map, int mapSize, const float* coordinateShift,
                           void* d_sendBuf);
void nbnxm_gpu_halo_unpack_f(void* stream, void* d_f, const int* d_map, int mapSize, const void* d_recvBuf, int accumulate);

/* Selects how perturbed pairs are evaluated (MI355X extension):
 *   0 = reference shape: cluster kernel + separate atom-pair FEP-list kernels (gpu_feplist);
 *   1 = fused: no FEP list; perturbed cluster pairs are found from Grid::fepBits on the device and evaluated by the
 *       cluster-pair FEP kernel from the same packed list (needs nbnxm_gpu_init_fep_cluster_bits). */
void nbnxm_gpu_set_fep_mode(NbnxmGpu* nb, int fused);

/* MI355X extension: lambda windows batched into one object.  With more windows than GPUs (or one GPU) R independent windows of
 * the same system become ONE object over R x N grid slots with one concatenated list (no pair between windows): the cluster
 * kernel does not depend on lambda, and one long list runs at a higher rate than R short ones (DESIGN §4.1 size table).
 * The perturbed-pair kernel of the fused mode takes the lambdas of a pair from this table, window = i-cluster /
 * clustersPerWindow.  Energies, dV/dlambda and foreign-lambda terms are kept per window: gpu_try/wait_finish_task adds the sum
 * over the windows to its accumulators, and nbnxm_gpu_get_window_energies ADDS window w's own share of the last finished energy
 * step to *enerd (returns -1 for a bad window).  Shift forces (virial) are a sum over the windows.  numWindows = 0 switches back
 * to the scalars of nbnxm_gpu_copy_fepparams. */
void nbnxm_gpu_set_window_lambdas(NbnxmGpu* nb, int numWindows, int clustersPerWindow, const float* lambda_q, const float* lambda_v);
int  nbnxm_gpu_get_window_energies(NbnxmGpu* nb, int window, nbnxm_enerdata_t* enerd, int haveSoftCore);

/* Diagnostics for tests: device pointer of the packed j-list of a locality, and a synchronous
 * device-to-host copy on that object's local stream. */
void* nbnxm_gpu_debug_get_cjpacked(NbnxmGpu* nb, int iloc);
/* device pointer of the work partition's range borders (numRanges + 1 ints) for the 4- (p = 0) or 5-waves-per-SIMD
 * (p = 1) kernels; *numRanges receives their count (tools/calibrate_weights.py) */
void* nbnxm_gpu_debug_get_work_ranges(NbnxmGpu* nb, int iloc, int p, int* numRanges);
void  nbnxm_gpu_debug_download(NbnxmGpu* nb, const void* devicePtr, void* hostPtr, size_t numBytes);
/* share of the total weight each range of partition p gets (numRanges = wave slots of the device, any positive numbers,
 * normalised inside; default: by age class of the wave on its SIMD).  For tests and tools/feedback_test.py. */
void  nbnxm_gpu_debug_set_work_shares(NbnxmGpu* nb, int iloc, int p, const float* shares, int numRanges);
/* measurement only (tools/graph_test.py): the clear + kernel launches of one steady-state local force step captured into
 * a hipGraph and replayed numSteps times; on MI355X this was slower than the plain launches */
void  nbnxm_gpu_debug_graph_steps(NbnxmGpu* nb, const nbnxm_step_workload_t* stepWork, int numSteps);

/* MI355X extension for domain decomposition.  The cluster kernel takes every wave slot of the device until its balanced ranges retire
 * together, so a non-local kernel queued beside the local one only starts behind it and the force halo is exposed.  With two parts the
 * local list is partitioned into two sets of ranges (the first holding firstPartFraction of the work):
 *   nbnxm_gpu_launch_kernel_part(nb, stepWork, NBNXM_LOCAL, 1)   beside the coordinate halo
 *   nbnxm_gpu_launch_kernel(nb, stepWork, NBNXM_NONLOCAL)        on the (high-priority) non-local stream
 *   nbnxm_gpu_launch_kernel_part(nb, stepWork, NBNXM_LOCAL, 2)   behind it, beside the force halo
 * Lists too short for two sets, and callers that use nbnxm_gpu_launch_kernel, run as before (part 2 is then empty). */
void nbnxm_gpu_set_local_launch_parts(NbnxmGpu* nb, int numParts, float firstPartFraction);
void nbnxm_gpu_launch_kernel_part(NbnxmGpu* nb, const nbnxm_step_workload_t* stepWork, int iloc, int part);

/* MI355X extension for domain decomposition: the local and the non-local pair list of a domain as ONE device list.  The cluster
 * kernel's launch has one wave per wave slot of the device, each with a balanced share of the list, so a second launch for the
 * non-local list pays the start and the drain of the whole machine again (81 us for the two kernels of a 96k + 46k-atom domain
 * against 62 us for the same pairs as one list).  Set before the lists are uploaded.  The reference's calls stay —
 * gpu_init_pairlist(Local) then (NonLocal), pairlist.cpp:4450-4452; launches / prunes / copy-backs / finish of both localities —
 * with this meaning: the non-local device list is empty, its launch, prune, copy-back and finish are no-ops; the LOCAL launch
 * evaluates both lists and therefore needs the halo coordinates — queue it behind x -> xq of the halo slots, which is NOT where
 * the reference's schedule has the local launch (sim_util.cpp:1783-1899 launches it before the coordinate halo arrives), so the
 * schedule is halo_gpu_domain_force_step's (include/halo_hip.h) or a caller's own; the LOCAL copy-back (nbnxm_gpu_launch_cpyback)
 * returns the forces of ALL atoms, home and halo, and the local finish is the step's only synchronisation point.  Perturbed pairs:
 * fused mode (nbnxm_gpu_set_fep_mode(nb, 1)); a non-local atom-pair list is refused. */
void nbnxm_gpu_set_merged_localities(NbnxmGpu* nb, int merged);
int  nbnxm_gpu_get_merged_localities(const NbnxmGpu* nb);

/* Host arithmetic only (no device call): the workgroup shape nbnxm_gpu_launch_kernel gives the cluster-pair kernel of a flavour —
 * waves per workgroup (4, 8 or 16: larger workgroups share one copy of the LDS tables when the LJ table of numTypes types is large),
 * resident waves per SIMD (5 or 4; < 4: the tables do not fit the 160 KB LDS and the launch aborts) and the LDS bytes per workgroup.
 * coulombTabSize: entries of the tabulated flavours' force table, else ignored. */
void nbnxm_hip_query_launch_shape(int elecType, int vdwType, int computeEnergy, int numTypes, int coulombTabSize, int* wavesPerWorkgroup,
                                  int* wavesPerSimd, int* ldsBytesPerWorkgroup);

/* Host arithmetic only (no device call): which sets of work ranges one call of nbnxm_gpu_launch_kernel (launchPart 0) or
 * nbnxm_gpu_launch_kernel_part (launchPart 1 / 2) launches for a list whose partition has workParts (1 or 2) sets and numRanges
 * ranges in all: sets firstSet .. firstSet + numSets - 1 of setRanges ranges each (numSets 0: the call queues no kernel), and whether
 * the trailing workgroups — perturbed cluster pairs, a pending rolling-prune part, the clear of the spare force buffer — ride with
 * the last of them.  A call that queues nothing consumes no state of the step (withTail 0).  Replaces nothing in the reference
 * (its launch is always whole, nbnxm_cuda.cu:642-760); it is what makes the two-part schedule testable on a CPU. */
void nbnxm_hip_query_launch_plan(int launchPart, int workParts, int numRanges, int* firstSet, int* numSets, int* setRanges, int* withTail);

/* Library/ABI version and a last-error string for diagnostics (never needed on the success path). */
int         nbnxm_hip_abi_version(void);
const char* nbnxm_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* NBNXM_HIP_H */
