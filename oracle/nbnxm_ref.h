/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of what the reference's GPU cluster-pair kernel evaluates on the
 * GPU-layout pair list (sci / cjPacked / excl, 8x8 clusters, cluster-pair split 2):
 *   /root/reference/src/gromacs/nbnxm/cuda/nbnxm_cuda_kernel.cuh:141-702       (semantics)
 *   /root/reference/src/gromacs/nbnxm/cuda/nbnxm_cuda_kernel_utils.cuh:76-518  (switches, LJ-PME)
 *   /root/reference/src/gromacs/nbnxm/kernels_reference/kernel_gpu_ref.cpp:57-354 (its CPU twin;
 *        list walk, exclusion-bit addressing :196-201, diagonal rule :191, self term :120-145)
 *   /root/reference/src/gromacs/nbnxm/cuda/nbnxm_cuda_kernel_pruneonly.cuh      (prune semantics)
 *
 * Pinning: the reference holds no known-answer vectors for this kernel (nbnxm/tests only checks
 * kernel selection).  It is pinned indirectly, see tests/test_pairlist_cpu.py:
 *   (1) against the golden-pinned FEP oracle in the A == B limit on the same pairs, and
 *   (2) against an O(N^2) minimum-image evaluation of the same functional forms.
 * Flavours without such a cross-check are listed as "parity unpinned" in DESIGN.md §3.
 */
#ifndef NBNXM_REF_H
#define NBNXM_REF_H

#include "../include/nbnxm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
    int    elecType; /* nbnxm_elec_type */
    int    vdwType;  /* nbnxm_vdw_type */
    double epsfac, c_rf, k_rf, ewaldcoeff_q, sh_ewald, sh_lj_ewald, ewaldcoeff_lj;
    double rcoulomb, rvdw, rvdw_switch, rlist;
    double disp_c2, disp_c3, disp_cpot; /* dispersion_shift */
    double rep_c2, rep_c3, rep_cpot;    /* repulsion_shift  */
    double sw_c3, sw_c4, sw_c5;         /* vdw_switch       */
    /* tabulated Ewald flavours: the force table the GPU kernel is handed (EwaldCorrectionTables::tableF, spacing 1/scale),
     * interpolated linearly as kernel_gpu_ref.cpp:265-271 / nbnxm_cuda_kernel_utils.cuh:448-459 do; NULL: analytical */
    double       coulomb_tab_scale;
    const float* coulomb_tab;
    int          coulomb_tab_size;
} nbnxm_ref_params_t;

#define NBNXM_REF_DECL(SUFFIX, REAL)                                                                \
    /* xq: 4 REAL per atom (grid order); type: per atom; nbfp: 2*ntype^2; lj_comb: 2 per atom or   \
     * NULL; nbfp_comb: 2 per type or NULL; shiftvec: 45x3; f: 3 per atom (+=); fshift: 45x3 (+=,  \
     * central shift skipped like the GPU kernel when skipCentralFshift != 0). */                   \
    void oracle_nbnxm_ref_##SUFFIX(int nsci, const nbnxn_sci_t* sci,                                \
                                   const nbnxn_cj_packed_t* cjPacked, const nbnxn_excl_t* excl,    \
                                   const REAL* xq, const int* type, int ntype, const REAL* nbfp,   \
                                   const REAL* lj_comb, const REAL* nbfp_comb,                     \
                                   const nbnxm_ref_params_t* p, const REAL* shiftvec,              \
                                   int computeEnergy, int computeFshift, REAL* f, REAL* fshift,    \
                                   double* Vc, double* Vvdw, long long* npairsWithinCutoff);        \
    /* the same on nthreads OpenMP threads (blocks of i-entries, per-thread force buffers summed at the end) */ \
    void oracle_nbnxm_ref_mt_##SUFFIX(int nthreads, int natoms, int nsci, const nbnxn_sci_t* sci,  \
                                      const nbnxn_cj_packed_t* cjPacked, const nbnxn_excl_t* excl, \
                                      const REAL* xq, const int* type, int ntype, const REAL* nbfp, \
                                      const REAL* lj_comb, const REAL* nbfp_comb,                  \
                                      const nbnxm_ref_params_t* p, const REAL* shiftvec,           \
                                      int computeEnergy, int computeFshift, REAL* f, REAL* fshift, \
                                      double* Vc, double* Vvdw, long long* npairsWithinCutoff);        \
    /* test scale: per shift vector and component the sum of |f_i| over the i-atoms booked to that shift force since the last  \
     * reset (out: 45 x 3 doubles or NULL; reset != 0 clears the sums) */                            \
    void oracle_nbnxm_fshift_abs_##SUFFIX(double* out, int reset);

NBNXM_REF_DECL(f64, double)
NBNXM_REF_DECL(f32, float)

/* Prune: clears imask bits of cluster pairs that have no atom pair within rlist
 * (both imei copies are written).  Returns the number of set bits left. */
long long oracle_nbnxm_prune(int nsci, const nbnxn_sci_t* sci, nbnxn_cj_packed_t* cjPacked,
                             const float* xq, const float* shiftvec, double rlist);

#ifdef __cplusplus
}
#endif
#endif
