/*
 * CPU oracle of the stochastic-dynamics (Langevin) update of the fork's GPU integrator: TEST INFRASTRUCTURE ONLY.
 *   random/threefry.h:420-600            ThreeFry2x64General<20, bits>::generateBlock = Threefry-2x64-20 of
 *                                        Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11)
 *   random/threefry.h:160-330,669-720    reserved high bits of key and counter for the internal counter
 *   random/tabulatednormaldistribution.h:171-212,282-304  table of the inverse error function, 14-bit look-ups
 *   mdlib/langevin_gpu_internal.cu:107-190,262-287         the update itself and its constants
 * Pinned by the reference's known answers: random/tests/refdata/KnownAnswersTest_ThreeFry2x64Test_Default_{0,1,2}.xml
 * and TabulatedNormalDistributionTest_Output14.xml (tests/golden/langevin_refdata.json).
 */
#ifndef LANGEVIN_REF_H
#define LANGEVIN_REF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C"
{
#endif

/* one block of Threefry-2x64-20 */
void oracle_threefry2x64(uint64_t key0, uint64_t key1, uint64_t ctr0, uint64_t ctr1, uint64_t out[2]);

/* TabulatedNormalDistribution<float, bits>::makeTable: table[1 << bits] */
void oracle_normal_table(int bits, float* table);

/* n draws of TabulatedNormalDistribution<float, 14> from ThreeFry2x64<internalCounterBits>(key0, domain) restarted at
 * (ctr0, ctr1): mean + stddev * table value */
void oracle_tabulated_normal(uint64_t key0, uint64_t domain, int internalCounterBits, uint64_t ctr0, uint64_t ctr1, float mean,
                             float stddev, int n, float* out);

/* updateType: 0 = forces only (v += f/m dt; xp = x; x += v dt), 1 = friction and noise only
 * (v' = v em + sqrt(1/m) sigmaV xi; x += 0.5 (v' - v) dt), xi from ThreeFry2x64<0>(seed, UpdateCoordinates) restarted
 * at (step, atom).  em = exp(-dt / tau_t) (1 if tau_t <= 0), sigmaV = sqrt(kB T (1 - em^2)) per coupling group. */
void oracle_langevin_update(int updateType, int numAtoms, float* x, float* xp, float* v, const float* f, const float* inverseMasses,
                            const unsigned short* tcGroups, int numGroups, const float* refT, const float* tauT, float dt, int seed,
                            int step);

#ifdef __cplusplus
}
#endif
#endif
