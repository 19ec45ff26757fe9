/*
 * C ABI of the stochastic-dynamics (Langevin) coordinate update on MI355X — SURVEY §8 row f4, the fork's
 * gmx::LangevinGpu (mdlib/langevin_gpu.h:90-160, kernels mdlib/langevin_gpu_internal.cu:107-190).
 *
 * As in the reference the update of a step is split around the constraints:
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

#ifdef __cplusplus
}
#endif
#endif
