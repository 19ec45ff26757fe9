/*
 * Stochastic-dynamics update on gfx950 — C ABI include/update_hip.h; semantics mdlib/langevin_gpu_internal.cu:107-190.
 * One thread per atom: 48 B read + 36-48 B written per atom, HBM-bound; the 64 KB table of the normal distribution is read
 * at three random 4-byte positions per atom and stays in L2.  The random numbers are one Threefry-2x64-20 block per atom
 * and step (random/threefry.h:420-600, a published counter-based generator), 42 of its first 64 bits.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "device_utils.h"
#include "update_device.h"
#include "update_hip.h"

using namespace nbnxm_hip;

namespace
{

constexpr double c_boltz = 1.380649e-23 * 6.02214076e23 / 1000.0; /* kJ/(mol K) */

template<int updateType>
__launch_bounds__(c_updateBlock) __global__
        void langevinKernel(const int numAtoms, float3* __restrict__ x, float3* __restrict__ xp, float3* __restrict__ v,
                            const float3* __restrict__ f, const float* __restrict__ inverseMasses, const float dt, const int seed,
                            const int step, const unsigned short* __restrict__ tcGroups, const float* __restrict__ sdSigmaV,
                            const float* __restrict__ sdConstEm, const float* __restrict__ table)
{
    const int a = static_cast<int>(blockIdx.x) * c_updateBlock + static_cast<int>(threadIdx.x);
    if (a >= numAtoms) { return; }
    float3      xa = x[a];
    float3      va = v[a];
    const float im = inverseMasses[a];
    if (updateType == LANGEVIN_FORCES_ONLY)
    {
        const float3 fa   = f[a];
        const float  imdt = im * dt;
        xp[a]             = xa;
        va                = make_float3(va.x + fa.x * imdt, va.y + fa.y * imdt, va.z + fa.z * imdt);
        xa                = make_float3(xa.x + va.x * dt, xa.y + va.y * dt, xa.z + va.z * dt);
    }
    else
    {
        const float3 xi  = langevinNoise(table, seed, step, a);
        const int    g   = tcGroups[a];
        const float  em  = sdConstEm[g];
        const float  amp = sqrtf(im) * sdSigmaV[g];
        const float3 vn  = va;
        va               = make_float3(vn.x * em + amp * xi.x, vn.y * em + amp * xi.y, vn.z * em + amp * xi.z);
        xa = make_float3(xa.x + 0.5F * (va.x - vn.x) * dt, xa.y + 0.5F * (va.y - vn.y) * dt, xa.z + 0.5F * (va.z - vn.z) * dt);
    }
    v[a] = va;
    x[a] = xa;
}

/* inverse error function in double (Halley iterations on erf), for the table of the normal distribution */
double erfinvHost(double y)
{
    const double a  = 0.147;
    const double ln = std::log(1.0 - y * y);
    const double t  = 2.0 / (M_PI * a) + 0.5 * ln;
    double       x  = std::copysign(std::sqrt(std::sqrt(t * t - ln / a) - t), y);
    for (int it = 0; it < 60; it++)
    {
        const double err = std::erf(x) - y;
        const double d   = 2.0 / std::sqrt(M_PI) * std::exp(-x * x);
        const double dx  = err / (d - x * err);
        x -= dx;
        if (std::fabs(dx) <= 1e-16 * std::fabs(x)) { break; }
    }
    return x;
}

/* TabulatedNormalDistribution<float, 14>::makeTable (random/tabulatednormaldistribution.h:171-212): quantile midpoints of
 * the unit normal, the two extremal entries chosen so that the table's variance is exactly 1 */
std::vector<float> makeNormalTable()
{
    const int          size = 1 << c_tableBits, half = size / 2;
    std::vector<float> t(size);
    for (int i = 0; i < half - 1; i++)
    {
        const double xv = std::sqrt(2.0) * erfinvHost((i + 0.5) / half);
        t[half - 1 - i] = static_cast<float>(-xv);
        t[half + i]     = static_cast<float>(xv);
    }
    double sumsq = 0;
    for (int i = 1; i < half; i++) { sumsq += static_cast<double>(t[i]) * t[i]; }
    const double extremal = std::sqrt(0.5 * (1.0 - 2.0 * sumsq / size) * size);
    t[0]                  = static_cast<float>(-extremal);
    t[size - 1]           = static_cast<float>(extremal);
    return t;
}

} // namespace


extern "C"
{

LangevinGpu* langevin_gpu_create(void* stream, int numTempCouplGroups, float delta_t, const float* ref_t, const float* tau_t)
{
    NBNXM_ASSERT(numTempCouplGroups > 0, "at least one temperature-coupling group is needed");
    LangevinGpu* lg = new LangevinGpu;
    lg->stream.init(stream);
    lg->numGroups = numTempCouplGroups;
    std::vector<float> em(numTempCouplGroups), sv(numTempCouplGroups);
    for (int g = 0; g < numTempCouplGroups; g++)
    {
        em[g]          = (tau_t[g] > 0) ? static_cast<float>(std::exp(-delta_t / tau_t[g])) : 1.0F;
        const float kT = static_cast<float>(c_boltz * ref_t[g]);
        sv[g]          = std::sqrt(kT * (1 - em[g] * em[g]));
    }
    const std::vector<float> table = makeNormalTable();
    allocateDeviceBuffer(&lg->d_sdConstEm, numTempCouplGroups);
    allocateDeviceBuffer(&lg->d_sdSigmaV, numTempCouplGroups);
    allocateDeviceBuffer(&lg->d_table, table.size());
    NBNXM_HIP_CHECK(hipMemcpy(lg->d_sdConstEm, em.data(), sizeof(float) * numTempCouplGroups, hipMemcpyHostToDevice));
    NBNXM_HIP_CHECK(hipMemcpy(lg->d_sdSigmaV, sv.data(), sizeof(float) * numTempCouplGroups, hipMemcpyHostToDevice));
    NBNXM_HIP_CHECK(hipMemcpy(lg->d_table, table.data(), sizeof(float) * table.size(), hipMemcpyHostToDevice));
    return lg;
}

void langevin_gpu_free(LangevinGpu* lg)
{
    if (lg == nullptr) { return; }
    (void)hipStreamSynchronize(lg->stream.stream);
    freeDeviceBuffer(&lg->d_sdConstEm);
    freeDeviceBuffer(&lg->d_sdSigmaV);
    freeDeviceBuffer(&lg->d_table);
    freeDeviceBuffer(&lg->d_inverseMasses);
    freeDeviceBuffer(&lg->d_tcGroups);
    lg->stream.destroy();
    delete lg;
}

void langevin_gpu_set(LangevinGpu* lg, int numAtoms, const float* inverseMasses, const unsigned short* tempCouplGroups)
{
    for (int i = 0; i < numAtoms; i++) { NBNXM_ASSERT(tempCouplGroups[i] < lg->numGroups, "temperature-coupling group out of range"); }
    if (numAtoms > lg->atomsAlloc)
    {
        freeDeviceBuffer(&lg->d_inverseMasses);
        freeDeviceBuffer(&lg->d_tcGroups);
        lg->atomsAlloc = static_cast<int>(numAtoms * 1.2) + 1024;
        allocateDeviceBuffer(&lg->d_inverseMasses, lg->atomsAlloc);
        allocateDeviceBuffer(&lg->d_tcGroups, lg->atomsAlloc);
    }
    lg->h_im.resize(numAtoms);
    lg->h_tc.resize(numAtoms);
    if (numAtoms)
    {
        std::memcpy(lg->h_im.data, inverseMasses, sizeof(float) * numAtoms);
        std::memcpy(lg->h_tc.data, tempCouplGroups, sizeof(unsigned short) * numAtoms);
    }
    copyToDeviceBuffer(&lg->d_inverseMasses, lg->h_im.data, 0, numAtoms, lg->stream.stream, true);
    copyToDeviceBuffer(&lg->d_tcGroups, lg->h_tc.data, 0, numAtoms, lg->stream.stream, true);
    lg->numAtoms = numAtoms;
}

void langevin_gpu_integrate(LangevinGpu* lg, void* d_x, void* d_xp, void* d_v, const void* d_f, float dt, int seed, int step, int updateType)
{
    NBNXM_ASSERT(updateType == LANGEVIN_FORCES_ONLY || updateType == LANGEVIN_FRICTION_AND_NOISE,
                 "the GPU integrator does the update in two steps, even without constraints (langevin_gpu_internal.cu:201-203)");
    if (lg->numAtoms == 0) { return; }
    NBNXM_ASSERT(d_x != nullptr && d_v != nullptr && (updateType != LANGEVIN_FORCES_ONLY || (d_xp != nullptr && d_f != nullptr)),
                 "coordinate / velocity / force buffer missing");
    const dim3 grid((lg->numAtoms + c_updateBlock - 1) / c_updateBlock);
    auto       k = (updateType == LANGEVIN_FORCES_ONLY) ? langevinKernel<LANGEVIN_FORCES_ONLY> : langevinKernel<LANGEVIN_FRICTION_AND_NOISE>;
    hipLaunchKernelGGL(k, grid, dim3(c_updateBlock), 0, lg->stream.stream, lg->numAtoms, static_cast<float3*>(d_x), static_cast<float3*>(d_xp),
                       static_cast<float3*>(d_v), static_cast<const float3*>(d_f), lg->d_inverseMasses, dt, seed, step, lg->d_tcGroups,
                       lg->d_sdSigmaV, lg->d_sdConstEm, lg->d_table);
    NBNXM_HIP_CHECK(hipGetLastError());
}

} // extern "C"
