/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's perturbed-pair non-bonded kernel,
 * the "oracle" that the HIP kernels are checked against.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Follows (semantics, not code):
 *   /root/reference/src/gromacs/gmxlib/nonbonded/nb_free_energy.cpp:274-1187   pair kernel
 *   /root/reference/src/gromacs/gmxlib/nonbonded/nb_softcore.h:44-279          Gapsys soft-core
 *   /root/reference/src/gromacs/nbnxm/freeenergydispatch.cpp:236-307           foreign-lambda loop
 *   /root/reference/src/gromacs/mdtypes/interaction_const.cpp:50-63            soft-core parameters
 *   /root/reference/src/gromacs/ewald/ewald_utils.cpp:43-130                   Ewald coefficients
 *
 * Pinning: reproduces all 72 known answers in tests/golden/nb_fep_refdata.json
 * (transcribed from gmxlib/nonbonded/tests/refdata) — see tests/test_oracle_golden.py.
 * The reference's own CPU build needs cmake-generated headers (config.h, gmxpre-config.h,
 * SIMD dispatch), so it is treated as unbuildable here; there is no oracle/_ref.
 *
 * The file is compiled twice: -DORACLE_REAL=double (suffix _f64) and =float (_f32).
 */
#ifndef FEP_ORACLE_H
#define FEP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* flags, same meaning as GMX_NONBONDED_DO_* (gmxlib/nonbonded/nonbonded.h) */
#define ORACLE_DO_FORCE 1
#define ORACLE_DO_SHIFTFORCE 2
#define ORACLE_DO_POTENTIAL 4

#define ORACLE_SOFTCORE_BEUTLER 0
#define ORACLE_SOFTCORE_GAPSYS 1

/* interaction_const_t + SoftCoreParameters subset used by the kernel; always double. */
typedef struct
{
    int elecIsEwald;  /* usingPmeOrEwald(eeltype); 0 = plain cut-off / reaction-field */
    int vdwIsEwald;   /* usingLJPme(vdwtype) */
    int vdwPotSwitch; /* vdw_modifier == PotSwitch */
    double epsfac;
    double rcoulomb, rvdw, rvdw_switch;
    double k_rf, c_rf; /* reactionFieldCoefficient, reactionFieldShift */
    double ewaldcoeff_q, ewaldcoeff_lj;
    double sh_ewald, sh_lj_ewald;
    double dispersion_shift_cpot, repulsion_shift_cpot;
    /* SoftCoreParameters */
    int    softcoreType; /* ORACLE_SOFTCORE_* */
    double alphaVdw, alphaCoulomb;
    int    lambdaPower;
    double sigma6WithInvalidSigma, sigma6Minimum;
    double gapsysScaleLinpointVdW, gapsysScaleLinpointCoul, gapsysSigma6VdW;
} oracle_fep_params_t;

/* Fills the soft-core members from t_lambda-style inputs (interaction_const.cpp:50-63). */
void oracle_softcore_from_fepvals(oracle_fep_params_t* p,
                                  double sc_alpha, int sc_power, double sc_sigma, double sc_sigma_min,
                                  int bScCoul, int softcoreType, double gapsysLinpointLJ,
                                  double gapsysLinpointQ, double gapsysSigmaLJ);

double oracle_calc_ewaldcoeff_q(double rc, double rtol);
double oracle_calc_ewaldcoeff_lj(double rc, double rtol);

#define ORACLE_DECL(SUFFIX, REAL)                                                                   \
    void oracle_nb_free_energy_kernel_##SUFFIX(int nri, const int* iinr, const int* jindex,          \
                                               const int* jjnr, const int* shift,                    \
                                               const int* excl_fep, const REAL* x, int ntype,        \
                                               const oracle_fep_params_t* p, const REAL* shiftvec,   \
                                               const REAL* nbfp, const REAL* nbfp_grid,              \
                                               const REAL* chargeA, const REAL* chargeB,             \
                                               const int* typeA, const int* typeB, int flags,        \
                                               double lambdaCoul, double lambdaVdw, REAL* f,         \
                                               REAL* fshift, double* Vc, double* Vv, double* dvdl);  \
    /* sums of |V_coul|, |V_vdw|, |dV/dl_coul|, |dV/dl_vdw| over the pairs of the last kernel call (test scale) */ \
    void oracle_fep_last_abs_sums_##SUFFIX(double* out4);                                            \
    /* and per shift vector and component the sum of |f_i| of the i-entries booked to it (45 x 3) */ \
    void oracle_fep_last_fshift_abs_##SUFFIX(double* out135);                                       \
    /* and the four abs sums of every lambda index of the last oracle_fep_foreign call (4 per index) */ \
    void oracle_fep_foreign_abs_sums_##SUFFIX(double* out, int numIndices);                         \
    void oracle_fep_foreign_##SUFFIX(int nri, const int* iinr, const int* jindex, const int* jjnr,   \
                                     const int* shift, const int* excl_fep, const REAL* x,           \
                                     int ntype, const oracle_fep_params_t* p, const REAL* shiftvec,  \
                                     const REAL* nbfp, const REAL* nbfp_grid, const REAL* chargeA,   \
                                     const REAL* chargeB, const int* typeA, const int* typeB,        \
                                     double lambdaCoul, double lambdaVdw, int n_lambda,              \
                                     const double* allLambdaCoul, const double* allLambdaVdw,        \
                                     double* eVdw, double* eCoul, double* dvdlVdw, double* dvdlCoul);

ORACLE_DECL(f64, double)
ORACLE_DECL(f32, float)

#ifdef __cplusplus
}
#endif
#endif
