/*
 * C ABI of the GPU coordinate update on MI355X — SURVEY §8 row f4: the fork's stochastic-dynamics integrator
 * gmx::LangevinGpu (mdlib/langevin_gpu.h:90-160, kernels mdlib/langevin_gpu_internal.cu:107-190) and the objects it is
 * driven together with in gmx::UpdateConstrainGpu (mdlib/update_constrain_gpu.h:70-185,
 * mdlib/update_constrain_gpu_impl.cpp:75-170): leap-frog, LINCS, SETTLE.  The fork's SD path *requires* the constraints
 * on the GPU (update_constrain_gpu_impl.cpp:93-94,215-217), so they are part of the same drop-in.
 *
 * Stochastic dynamics: as in the reference the update of a step is split around the constraints:
 *   LANGEVIN_FORCES_ONLY            v += f / m dt;  xp = x;  x += v dt
 *   LANGEVIN_FRICTION_AND_NOISE     v' = v em + sqrt(1/m) sigmaV xi;  x += 0.5 (v' - v) dt
 * with em = exp(-dt / tau_t), sigmaV = sqrt(kB T (1 - em^2)) per temperature-coupling group and xi drawn per atom and
 * step from the 14-bit tabulated normal distribution driven by ThreeFry2x64<0>(seed, RandomDomain::UpdateCoordinates)
 * restarted at (step, atom index): the same random stream as the CPU integrator, so trajectories are reproducible
 * across implementations.  x, xp, v, f: device float3 arrays in atom order.
 */
#ifndef UPDATE_HIP_H
#define UPDATE_HIP_H

#ifdef __cplusplus
extern "C"
{
#endif

typedef struct LangevinGpu LangevinGpu;

enum
{
    LANGEVIN_FORCES_ONLY = 0,       /* SDUpdate::ForcesOnly */
    LANGEVIN_FRICTION_AND_NOISE = 1 /* SDUpdate::FrictionAndNoiseOnly */
};

/* LangevinGpu::LangevinGpu — langevin_gpu.h:100-112: per-group reference temperatures and coupling times; stream NULL:
 * an own stream */
LangevinGpu* langevin_gpu_create(void* stream, int numTempCouplGroups, float delta_t, const float* ref_t, const float* tau_t);
void         langevin_gpu_free(LangevinGpu* lg);

/* LangevinGpu::set — langevin_gpu.h:139-147: inverse masses and temperature-coupling group of every atom */
void langevin_gpu_set(LangevinGpu* lg, int numAtoms, const float* inverseMasses, const unsigned short* tempCouplGroups);

/* LangevinGpu::integrate — langevin_gpu.h:114-137 */
void langevin_gpu_integrate(LangevinGpu* lg, void* d_x, void* d_xp, void* d_v, const void* d_f, float dt, int seed, int step,
                            int updateType);

/* ---- leap-frog: gmx::LeapFrogGpu (mdlib/leapfrog_gpu.h:95-170, kernel mdlib/leapfrog_gpu_internal.cu:92-160) ------------
 *   xp = x;  v = lambda[group] v - diag(dtPressureCouple M) v_old + f / m dt;  x += v dt */
typedef struct LeapFrogGpu LeapFrogGpu;

/* LeapFrogGpu::LeapFrogGpu — numTempScaleValues: 0 = no temperature coupling, 1 = one factor, > 1 = one per group */
LeapFrogGpu* leapfrog_gpu_create(void* stream, int numTempScaleValues);
void         leapfrog_gpu_free(LeapFrogGpu* lf);
/* LeapFrogGpu::set — leapfrog_gpu.h:150-160; tempScaleGroups is read only when numTempScaleValues > 1 */
void leapfrog_gpu_set(LeapFrogGpu* lf, int numAtoms, const float* inverseMasses, const unsigned short* tempScaleGroups);
/* LeapFrogGpu::integrate — leapfrog_gpu.h:118-148.  tcLambdas: numTempScaleValues factors (t_grp_tcstat::lambda),
 * prVelocityScalingMatrix: row-major 3x3; only a diagonal matrix is supported, as in the reference (leapfrog_gpu.cpp:104-110) */
void leapfrog_gpu_integrate(LeapFrogGpu* lf, void* d_x, void* d_xp, void* d_v, const void* d_f, float dt, int doTemperatureScaling,
                            const float* tcLambdas, int doParrinelloRahman, float dtPressureCouple,
                            const float* prVelocityScalingMatrix);

/* ---- SETTLE: gmx::SettleGpu (mdlib/settle_gpu.h:70-150, kernel mdlib/settle_gpu_internal.cu:92-372) ----------------------
 * Analytical constraint of rigid three-site waters (Miyamoto & Kollman 1992, oxygen as reference point). */
typedef struct SettleGpu SettleGpu;

/* SettleGpu::SettleGpu — the one SETTLE type of the topology: masses and the O-H, H-H target distances
 * (settle_gpu.cpp:118-140, parameters derived in double as mdlib/settle.cpp:111-151) */
SettleGpu* settle_gpu_create(void* stream, float mO, float mH, float dOH, float dHH);
void       settle_gpu_free(SettleGpu* sg);
/* SettleGpu::set — settle_gpu.cpp:163-182: atoms[3 n] = (O, H, H) indices of the local waters */
void settle_gpu_set(SettleGpu* sg, int numSettles, const int* atoms);
/* SettleGpu::apply — settle_gpu.cpp:67-116.  d_x: coordinates before the update, d_xp: after it (constrained in place),
 * d_v may be NULL unless updateVelocities; the 3x3 row-major virialScaled is ADDED to when computeVirial (synchronous, as in
 * the reference); pbcType: 0 none, 2 xy, 3 xyz (setPbcAiuc, pbcutil/pbc_aiuc.h:98-140), box row-major 3x3 */
void settle_gpu_apply(SettleGpu* sg, const void* d_x, void* d_xp, int updateVelocities, void* d_v, float invdt, int computeVirial,
                      float* virialScaled, int pbcType, const float* box);

/* ---- LINCS: gmx::LincsGpu (mdlib/lincs_gpu.h:75-160, kernel mdlib/lincs_gpu_internal.cu:91-377) ------------------------- */
typedef struct LincsGpu LincsGpu;

LincsGpu* lincs_gpu_create(void* stream, int numIterations, int expansionOrder);
void      lincs_gpu_free(LincsGpu* lg);
/* LincsGpu::set — lincs_gpu.cpp:216-495: iatoms[3 n] = (type, i, j) as InteractionDefinitions::il[F_CONSTR], lengths[type]
 * = t_iparams::constr.dA.  Returns 0, or -1 when a group of coupled constraints exceeds the largest work-group (1024;
 * LincsGpu::isNumCoupledConstraintsSupported, lincs_gpu.cpp:211-214) — nothing is changed in that case. */
int lincs_gpu_set(LincsGpu* lg, int numConstraints, const int* iatoms, const float* lengths, int numAtoms, const float* inverseMasses);
/* LincsGpu::apply — lincs_gpu.cpp:67-118; arguments as settle_gpu_apply */
void lincs_gpu_apply(LincsGpu* lg, const void* d_x, void* d_xp, int updateVelocities, void* d_v, float invdt, int computeVirial,
                     float* virialScaled, int pbcType, const float* box);

/* ---- the composite: gmx::UpdateConstrainGpu (mdlib/update_constrain_gpu.h:70-185) ---------------------------------------- */
typedef struct UpdateConstrainGpu UpdateConstrainGpu;

typedef struct
{
    int          useStochasticDynamics; /* t_inputrec::eI == SD1: LangevinGpu, else LeapFrogGpu (update_constrain_gpu_impl.cpp:203-221) */
    int          numTempCouplGroups;
    float        delta_t;
    const float* ref_t; /* [numTempCouplGroups], SD only */
    const float* tau_t; /* [numTempCouplGroups], SD only */
    int          nLincsIter, nProjOrder;
    int          haveSettle; /* the topology holds a SETTLE type */
    float        mO, mH, dOH, dHH;
} update_constrain_params_t;

typedef struct
{
    int                   numAtoms; /* t_mdatoms::homenr */
    const float*          inverseMasses;
    const unsigned short* tempCouplGroups; /* t_mdatoms::cTC */
    int                   numConstraints;
    const int*            constraints; /* (type, i, j) triples */
    const float*          constraintLengths;
    int                   numSettles;
    const int*            settles; /* (O, H, H) triples */
} update_constrain_topology_t;

UpdateConstrainGpu* update_constrain_gpu_create(void* stream, const update_constrain_params_t* params);
void                update_constrain_gpu_free(UpdateConstrainGpu* uc);
/* UpdateConstrainGpu::set — update_constrain_gpu_impl.cpp:231-281; returns lincs_gpu_set's status */
int update_constrain_gpu_set(UpdateConstrainGpu* uc, void* d_x, void* d_v, const void* d_f, const update_constrain_topology_t* topology);
/* UpdateConstrainGpu::setPbc — update_constrain_gpu_impl.cpp:283-287 */
void update_constrain_gpu_set_pbc(UpdateConstrainGpu* uc, int pbcType, const float* box);
/* UpdateConstrainGpu::integrate — update_constrain_gpu_impl.cpp:75-170.  fReadyEvent: hipEvent_t the update stream waits
 * for (NULL: none).  virial (row-major 3x3) is overwritten with 0.5 / dt^2 times the constraint virial when computeVirial,
 * cleared otherwise.  After the call d_x holds the new constrained coordinates. */
void update_constrain_gpu_integrate(UpdateConstrainGpu* uc, void* fReadyEvent, float dt, int updateVelocities, int computeVirial,
                                    float* virial, int doTemperatureScaling, const float* tcLambdas, int doParrinelloRahman,
                                    float dtPressureCouple, const float* prVelocityScalingMatrix, int seed, int step);
/* UpdateConstrainGpu::scaleCoordinates / scaleVelocities — update_constrain_gpu_impl.cpp:172-201 (lower-triangular
 * row-major 3x3 matrix, kernel update_constrain_gpu_internal.cu:60-77) */
void update_constrain_gpu_scale_coordinates(UpdateConstrainGpu* uc, const float* scalingMatrix);
void update_constrain_gpu_scale_velocities(UpdateConstrainGpu* uc, const float* scalingMatrix);
/* UpdateConstrainGpu::xUpdatedOnDeviceEvent — hipEvent_t recorded at the end of every integrate that FOLLOWS this call (a stream
 * whose coordinates nobody else waits for is spared the event) */
void* update_constrain_gpu_x_updated_event(UpdateConstrainGpu* uc);

/* ---- MI355X extension: the update fused with the non-bonded buffers ---------------------------------------------------------
 * Between two searches the reference runs five kernels per step around the non-bonded ones: x -> xq (nbnxn_gpu_x_to_nbat_x),
 * clear, force reduction, integrator, SETTLE, each one pass over all atoms.  With the atom -> grid-slot map of the search the
 * update can read the non-bonded force buffer itself, clear it behind itself and write the new coordinates straight into xq:
 * ONE kernel, coordinates before the update live in registers only.  Atoms with LINCS constraints (leap-frog only) get their old
 * coordinates stored for the LINCS kernel, which follows and also writes its result into xq.  Stochastic dynamics with LINCS
 * constraints needs the friction step between two constraint passes over memory: use update_constrain_gpu_integrate there. */

/* after update_constrain_gpu_set, every search: cell[numAtoms] = grid slot of each atom (inverse of gridSet.atomIndices()),
 * d_xq / d_f_nbat = nbnxm_gpu_get_xq / nbnxm_gpu_get_f of the non-bonded object */
void update_constrain_gpu_set_nbat_coupling(UpdateConstrainGpu* uc, const int* cell, void* d_xq, void* d_f_nbat);
int  update_constrain_gpu_can_fuse(const UpdateConstrainGpu* uc);
/* as update_constrain_gpu_integrate with updateVelocities = true.  Forces: the non-bonded buffer (grid order), plus the d_f of
 * update_constrain_gpu_set (atom order) when addAtomOrderForces.  Leaves the non-bonded force buffer cleared and xq holding the
 * new coordinates: the next step needs neither nbnxm_gpu_x_to_nbat_x nor the force part of nbnxm_gpu_clear_outputs. */
void update_constrain_gpu_integrate_fused(UpdateConstrainGpu* uc, void* fReadyEvent, float dt, int computeVirial, float* virial,
                                          int doTemperatureScaling, const float* tcLambdas, int doParrinelloRahman, float dtPressureCouple,
                                          const float* prVelocityScalingMatrix, int seed, int step, int addAtomOrderForces);

#ifdef __cplusplus
}
#endif
#endif
