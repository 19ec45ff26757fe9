/*
 * C ABI of the MI355X listed-forces (bonded) path with free-energy perturbation — SURVEY §8 row f3.
 *
 * Replaces the perturbed part of gmx::ListedForcesGpu (listed_forces/listed_forces_gpu.h:120-190; implementation
 * listed_forces_gpu_impl.cu, kernels listed_forces_gpu_internal.cu:781-1363: bonds_fep_gpu, angles_fep_gpu,
 * urey_bradley_fep_gpu, pdihs_fep_gpu, rbdihs_fep_gpu, idihs_fep_gpu) for the function types below; every type takes
 * A- and B-state parameters and lambda_bonded, so the unperturbed interactions are the A == B case of the same code.
 * The perturbed 1-4 pairs (pairs_fep_gpu :1365-1600: soft-core LJ + plain Coulomb between the A/B charges of the
 * non-bonded module's q4 buffer) are the seventh type.  Not built: the restraint types.
 *
 * Conventions as in the reference: coordinates are the non-bonded module's xq (float4, nbnxm grid order), atom indices
 * in the interaction lists are already translated to that order (nbnxnAtomOrder, listed_forces_gpu_impl.cu
 * convertIlistToNbnxnOrder), forces are accumulated into the non-bonded force buffer with atomics, energies stay on
 * the device until launch_energy_transfer / wait_accumulate_energy_terms.
 */
#ifndef LISTED_HIP_H
#define LISTED_HIP_H

#ifdef __cplusplus
extern "C"
{
#endif

typedef struct ListedGpu ListedGpu;

enum
{
    LISTED_GPU_BONDS = 0,    /* F_BONDS, F_HARMONIC [type, ai, aj]        p: rA krA rB krB */
    LISTED_GPU_ANGLES,       /* F_ANGLES          [type, ai, aj, ak]      p: thA kA thB kB (degrees) */
    LISTED_GPU_UREY_BRADLEY, /* F_UREY_BRADLEY    [type, ai, aj, ak]      p: thetaA kthetaA r13A kUBA thetaB kthetaB r13B kUBB */
    LISTED_GPU_PDIHS,        /* F_PDIHS, F_PIDIHS [type, ai, aj, ak, al]  p: phiA cpA phiB cpB; mult */
    LISTED_GPU_RBDIHS,       /* F_RBDIHS          [type, ai, aj, ak, al]  p: rbcA[6] rbcB[6] */
    LISTED_GPU_IDIHS,        /* F_IDIHS           [type, ai, aj, ak, al]  p: xA kA xB kB (degrees) */
    LISTED_GPU_LJ14,         /* F_LJ14            [type, ai, aj]          p: c6A c12A c6B c12B (plain C6, C12) */
    /* the other types of the fork's GPU list (listed_forces_gpu.h:87-90) */
    LISTED_GPU_LJC14_Q,      /* F_LJC14_Q         [type, ai, aj]          p: qi qj fqq c6 c12; Coulomb with epsfac, unperturbed */
    LISTED_GPU_LJC_PAIRS_NB, /* F_LJC_PAIRS_NB    [type, ai, aj]          p: qi qj c6 c12; Coulomb with epsfac, unperturbed */
    LISTED_GPU_RESTRBONDS,   /* F_RESTRBONDS      [type, ai, aj]          p: lowA up1A up2A kA lowB up1B up2B kB */
    LISTED_GPU_ANGRES,       /* F_ANGRES          [type, ai, aj, ak, al]  p: phiA cpA phiB cpB; mult (angle between i->j and k->l) */
    LISTED_GPU_DIHRES,       /* F_DIHRES          [type, ai, aj, ak, al]  p: phiA dphiA kfacA phiB dphiB kfacB (degrees) */
    LISTED_GPU_NUM_TYPES
};

/* energy terms returned by listed_gpu_wait_accumulate_energy_terms: one per function type (the slots of the pair types hold
 * the Lennard-Jones part) plus the Coulomb part of the 1-4 pairs (F_LJ14, F_LJC14_Q -> F_COUL14) and of F_LJC_PAIRS_NB
 * (-> F_COUL_SR) */
enum
{
    LISTED_GPU_ENERGY_COULOMB14 = LISTED_GPU_NUM_TYPES,
    LISTED_GPU_ENERGY_COULOMB_PAIRS_NB,
    LISTED_GPU_NUM_ENERGY_TERMS
};
/* dV/dlambda components (FreeEnergyPerturbationCouplingType Bonded, Coul, Vdw, Restraint: the reference's CPU path books
 * the restraint types under Restraint, listed_forces.cpp calc_one_bond; its GPU path adds them to Bonded) */
enum
{
    LISTED_GPU_DVDL_BONDED = 0,
    LISTED_GPU_DVDL_COUL,
    LISTED_GPU_DVDL_VDW,
    LISTED_GPU_DVDL_RESTRAINT,
    LISTED_GPU_NUM_DVDL
};

/* gmx::BondedFepParameters (listed_forces_gpu.h, set from interaction_const_t::SoftCoreParameters and the lambdas) */
typedef struct
{
    float alphaCoul, alphaVdw;
    int   lambdaPower;
    float sc_sigma6, sc_sigma6_min;
    float lambdaBonded, lambdaCoul, lambdaVdw;
    float lambdaRestraint;
} listed_gpu_fep_params_t;

/* t_iparams of the types above (topology/idef.h:71-330), float, A and B state side by side */
typedef struct
{
    float p[12];
    int   mult;
} listed_gpu_iparams_t;

/* ListedForcesGpu::ListedForcesGpu — listed_forces_gpu.h:125-131.  stream: hipStream_t to launch on (the non-bonded
 * local stream in the reference), NULL: an own stream. */
ListedGpu* listed_gpu_create(void* stream);
void       listed_gpu_free(ListedGpu* lg);

/* ListedForcesGpu::updateInteractionListsAndDeviceBuffers — listed_forces_gpu.h:146-151: the force parameters
 * (idef.iparams) and, per function type, the interaction list in nbnxm atom order, 1 + nral ints per interaction. */
void listed_gpu_set_force_params(ListedGpu* lg, int numParams, const listed_gpu_iparams_t* params);
void listed_gpu_update_interaction_list(ListedGpu* lg, int ftype, int numInteractions, const int* iatoms, int numAtoms);

/* ListedForcesGpu::haveInteractions — listed_forces_gpu.h:153-158 */
int listed_gpu_have_interactions(const ListedGpu* lg);

/* ListedForcesGpu::launchKernel(stepWork, box) — listed_forces_gpu.h:160-170.  d_xq: float4[], d_q4: float4[] with
 * .x = qA, .y = qB (NBAtomDataGpu::q4; may be NULL without LJ14 pairs), d_f: float3[] (+=), d_fshift: float3[45] (+= when
 * computeVirial); box: 3x3 row-major; pbcType: 0 none, 2 xy, 3 xyz; electrostaticsScaleFactor = epsfac * fudgeQQ (F_LJ14),
 * epsfac alone scales the Coulomb part of F_LJC14_Q and F_LJC_PAIRS_NB (BondedGpuKernelParameters::epsFac). */
void listed_gpu_launch_kernel(ListedGpu* lg, const void* d_xq, const void* d_q4, void* d_f, void* d_fshift, const float* box,
                              int pbcType, const listed_gpu_fep_params_t* fep, float electrostaticsScaleFactor, float epsfac,
                              int computeEnergy, int computeVirial);

/* launchEnergyTransfer / waitAccumulateEnergyTerms / clearEnergies — listed_forces_gpu.h:172-190:
 * epot[LISTED_GPU_NUM_ENERGY_TERMS] += energies, dvdl[LISTED_GPU_NUM_DVDL] += dV/dlambda components. */
void listed_gpu_launch_energy_transfer(ListedGpu* lg);
void listed_gpu_wait_accumulate_energy_terms(ListedGpu* lg, double* epot, double* dvdl);
void listed_gpu_clear_energies(ListedGpu* lg);

#ifdef __cplusplus
}
#endif
#endif
