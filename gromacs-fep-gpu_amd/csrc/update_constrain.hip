/*
 * Leap-frog, SETTLE, LINCS and their composition on gfx950 — C ABI include/update_hip.h.
 * Semantics: mdlib/leapfrog_gpu_internal.cu:92-160, mdlib/settle_gpu_internal.cu:92-372, mdlib/lincs_gpu_internal.cu:91-377,
 * mdlib/update_constrain_gpu_impl.cpp:75-170.
 *
 * What is MI355X-specific:
 *  - LINCS keeps a group of coupled constraints inside ONE work-group (as the reference does) but also keeps everything the
 *    iterations touch in LDS: the unit vectors, the coupling matrix, the right-hand sides and the DISPLACEMENT of every atom
 *    the work-group's constraints touch.  An atom with a constraint belongs to exactly one coupled group, hence to one
 *    work-group, so the position corrections are LDS float atomics (ds_add_f32) and each atom is written back once; the
 *    reference sends 6 global float atomics per constraint and iteration to xp and re-reads xp from memory.  Velocities
 *    follow from the same displacement (dv = dx / dt), no second set of atomics.
 *  - The work-group is one wavefront (64 constraints) whenever the largest coupled group fits, so the barriers between
 *    the matrix-expansion steps never wait for another wave; 128 … 1024 only for larger groups.
 *  - Virial: wave-wide shuffle reduction, then 6 atomics per wave.
 * All kernels are HBM/latency-bound streaming kernels (36-72 B per atom); their cost at 100k atoms is a few µs each.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "device_utils.h"
#include "pbc_aiuc.h"
#include "update_device.h"
#include "update_hip.h"

using namespace nbnxm_hip;

namespace
{

/* ---------------------------------------------------------------------------------------------------------------------- */
/* leap-frog */

template<int tempScaling /* 0 none, 1 one factor, 2 per group (kernel arguments), 3 per group (device buffer) */, bool parrinelloRahman>
__launch_bounds__(c_updateBlock) __global__
        void leapfrogKernel(const int numAtoms, float3* __restrict__ x, float3* __restrict__ xp, float3* __restrict__ v,
                            const float3* __restrict__ f, const float* __restrict__ inverseMasses, const float dt,
                            const TcLambdas lambdaArgs, const float* __restrict__ lambdas, const unsigned short* __restrict__ groups,
                            const float3 prDiagonal)
{
    const int a = static_cast<int>(blockIdx.x) * c_updateBlock + static_cast<int>(threadIdx.x);
    if (a >= numAtoms) { return; }
    float3       xa   = x[a];
    float3       va   = v[a];
    const float3 fa   = f[a];
    const float  imdt = inverseMasses[a] * dt;
    xp[a]             = xa;
    if (tempScaling != 0 || parrinelloRahman)
    {
        float3 vs = va;
        if (tempScaling != 0)
        {
            const float lambda = (tempScaling == 1) ? lambdaArgs.v[0] : (tempScaling == 2) ? lambdaArgs.v[groups[a]] : lambdas[groups[a]];
            vs                 = lambda * vs;
        }
        if (parrinelloRahman)
        {
            vs.x -= prDiagonal.x * va.x;
            vs.y -= prDiagonal.y * va.y;
            vs.z -= prDiagonal.z * va.z;
        }
        va = vs;
    }
    va   = va + imdt * fa;
    xa   = xa + dt * va;
    v[a] = va;
    x[a] = xa;
}

__launch_bounds__(c_updateBlock) __global__ void scaleKernel(const int numAtoms, float3* __restrict__ x, const float xx, const float yy,
                                                             const float zz, const float yx, const float zx, const float zy)
{
    const int a = static_cast<int>(blockIdx.x) * c_updateBlock + static_cast<int>(threadIdx.x);
    if (a >= numAtoms) { return; }
    float3 p = x[a];
    p.x      = xx * p.x + yx * p.y + zx * p.z;
    p.y      = yy * p.y + zy * p.z;
    p.z      = zz * p.z;
    x[a]     = p;
}

/* ---------------------------------------------------------------------------------------------------------------------- */
/* SETTLE */

template<bool updateVelocities, bool computeVirial>
__launch_bounds__(c_updateBlock) __global__
        void settleKernel(const int numSettles, const int* __restrict__ atoms, const SettlePars pars, const float3* __restrict__ x,
                          float3* __restrict__ xp, const float invdt, float3* __restrict__ v, float* __restrict__ virial, const PbcAiuc pbc)
{
    const int s = static_cast<int>(blockIdx.x) * c_updateBlock + static_cast<int>(threadIdx.x);
    float     vir[6] = { 0, 0, 0, 0, 0, 0 };
    if (s < numSettles)
    {
        const int    io = atoms[3 * s], ih2 = atoms[3 * s + 1], ih3 = atoms[3 * s + 2];
        const float3 xo = x[io], xh2 = x[ih2], xh3 = x[ih3];
        const float3 po = xp[io], ph2 = xp[ih2], ph3 = xp[ih3];

        const float3 dist21 = pbcDxAiuc(pbc, xh2, xo);
        const float3 dist31 = pbcDxAiuc(pbc, xh3, xo);
        const float3 doh2   = pbcDxAiuc(pbc, ph2, po);
        const float3 doh3   = pbcDxAiuc(pbc, ph3, po);

        float3 dxO, dxH2, dxH3;
        settleTriangle(pars, dist21, dist31, doh2, doh3, dxO, dxH2, dxH3);

        xp[io]  = po + dxO;
        xp[ih2] = ph2 + dxH2;
        xp[ih3] = ph3 + dxH3;
        if (updateVelocities)
        {
            v[io]  = v[io] + invdt * dxO;
            v[ih2] = v[ih2] + invdt * dxH2;
            v[ih3] = v[ih3] + invdt * dxH3;
        }
        if (computeVirial) { settleVirial(pars, xo, dist21, dist31, dxO, dxH2, dxH3, vir); }
    }
    if (computeVirial) { addWaveVirial(virial, vir); }
}

/* ---------------------------------------------------------------------------------------------------------------------- */
/* LINCS */

struct LincsKernelArgs
{
    int          numThreads; /* work-groups x work-group size; every per-constraint array is padded to it */
    int          numIterations, expansionOrder, maxCoupled;
    const int2*  constraints; /* global atom indices, (-1,-1): padding */
    const int2*  localSlots;  /* the same atoms as slots of the work-group's atom list */
    const float* lengths;
    const float2* inverseMasses; /* of the two atoms */
    const int*   coupledCounts;
    const int*   coupledIndices; /* [n * numThreads + thread]: thread index inside the work-group */
    const float* massFactors;    /* same layout */
    const int*   blockAtomStart; /* [numBlocks + 1] into blockAtoms */
    const int*   blockAtoms;     /* global indices of the atoms a work-group owns */
    const int*   blockAtomSlots; /* their grid slots, or null: also write the constrained coordinates into xq (fused update) */
    float*       xq;
    float*       virial;
    PbcAiuc      pbc;
};

template<bool updateVelocities, bool computeVirial>
__global__ void lincsKernel(const LincsKernelArgs a, const float3* __restrict__ x, float3* __restrict__ xp, float3* __restrict__ v,
                            const float invdt)
{
    extern __shared__ float sm[];
    const int    B    = static_cast<int>(blockDim.x);
    const int    t    = static_cast<int>(threadIdx.x);
    const int    gt   = static_cast<int>(blockIdx.x) * B + t;
    float* const disp = sm;                /* [3][2B] displacement of the work-group's atoms */
    float* const r    = disp + 6 * B;      /* [3][B] unit vectors of the constraints before the update */
    float* const rhs  = r + 3 * B;         /* [2][B] */
    float* const matA = rhs + 2 * B;       /* [maxCoupled][B] */
    int* const   nbr  = reinterpret_cast<int*>(matA + a.maxCoupled * B); /* [maxCoupled][B] */

    const int atomStart = a.blockAtomStart[blockIdx.x];
    const int numLocal  = a.blockAtomStart[blockIdx.x + 1] - atomStart;
    for (int k = t; k < 6 * B; k += B) { disp[k] = 0.0F; }

    const int2 pair  = a.constraints[gt];
    const int2 slot  = a.localSlots[gt];
    const bool dummy = (pair.x < 0);

    float  len = 0.0F, imi = 0.0F, imj = 0.0F, sqrtMu = 0.0F;
    float3 rc  = make_float3(0.0F, 0.0F, 0.0F);
    float3 xi0 = rc, xj0 = rc;
    if (!dummy)
    {
        len             = a.lengths[gt];
        const float2 im = a.inverseMasses[gt];
        imi             = im.x;
        imj             = im.y;
        sqrtMu          = rsqrtf(imi + imj);
        const float3 dx = pbcDxAiuc(a.pbc, x[pair.x], x[pair.y]);
        rc              = rsqrtf(dot3(dx, dx)) * dx;
        xi0             = xp[pair.x];
        xj0             = xp[pair.y];
    }
    r[t]         = rc.x;
    r[B + t]     = rc.y;
    r[2 * B + t] = rc.z;
    __syncthreads();

    /* coupling matrix: non-zero entries only */
    const int count = a.coupledCounts[gt];
    for (int n = 0; n < count; n++)
    {
        const int c1    = a.coupledIndices[n * a.numThreads + gt];
        nbr[n * B + t]  = c1;
        matA[n * B + t] = a.massFactors[n * a.numThreads + gt] * (rc.x * r[c1] + rc.y * r[B + c1] + rc.z * r[2 * B + c1]);
    }

    /* (1 - A)^-1 rhs ~ (1 + A + A^2 + ...) rhs; the two halves of rhs[] alternate as source and destination */
    auto expand = [&](float rhs0) {
        float sol = rhs0;
        rhs[t]    = rhs0;
        for (int rec = 0; rec < a.expansionOrder; rec++)
        {
            __syncthreads();
            const float* src = rhs + B * (rec & 1);
            float        mvb = 0.0F;
            for (int n = 0; n < count; n++) { mvb += matA[n * B + t] * src[nbr[n * B + t]]; }
            rhs[B * ((rec + 1) & 1) + t] = mvb;
            sol += mvb;
        }
        return sol;
    };
    auto displace = [&](float lagrange) {
        if (!dummy)
        {
            const float si = -lagrange * imi, sj = lagrange * imj;
            atomicAdd(&disp[slot.x], si * rc.x);
            atomicAdd(&disp[2 * B + slot.x], si * rc.y);
            atomicAdd(&disp[4 * B + slot.x], si * rc.z);
            atomicAdd(&disp[slot.y], sj * rc.x);
            atomicAdd(&disp[2 * B + slot.y], sj * rc.y);
            atomicAdd(&disp[4 * B + slot.y], sj * rc.z);
        }
    };

    float3 dx             = pbcDxAiuc(a.pbc, xi0, xj0);
    float  lagrangeScaled = sqrtMu * expand(sqrtMu * (dot3(rc, dx) - len));
    displace(lagrangeScaled);

    /* correction for the lengthening of a bond that rotated */
    for (int iter = 0; iter < a.numIterations; iter++)
    {
        __syncthreads();
        float3 xi = xi0, xj = xj0;
        if (!dummy)
        {
            xi = xi0 + make_float3(disp[slot.x], disp[2 * B + slot.x], disp[4 * B + slot.x]);
            xj = xj0 + make_float3(disp[slot.y], disp[2 * B + slot.y], disp[4 * B + slot.y]);
        }
        dx                = pbcDxAiuc(a.pbc, xi, xj);
        const float dlen2 = 2.0F * len * len - dot3(dx, dx);
        const float proj  = sqrtMu * ((dlen2 > 0.0F) ? len - dlen2 * rsqrtf(dlen2) : len);
        const float corr  = sqrtMu * expand(proj);
        lagrangeScaled += corr;
        displace(corr);
    }
    __syncthreads();

    /* every atom of the work-group is written exactly once */
    for (int k = t; k < numLocal; k += B)
    {
        const int    g = a.blockAtoms[atomStart + k];
        const float3 d = make_float3(disp[k], disp[2 * B + k], disp[4 * B + k]);
        const float3 xn = xp[g] + d;
        xp[g]          = xn;
        if (updateVelocities) { v[g] = v[g] + invdt * d; }
        if (a.blockAtomSlots != nullptr)
        {
            const int slot     = a.blockAtomSlots[atomStart + k];
            a.xq[4 * slot + 0] = xn.x;
            a.xq[4 * slot + 1] = xn.y;
            a.xq[4 * slot + 2] = xn.z;
        }
    }
    if (computeVirial)
    {
        const float mult   = len * lagrangeScaled;
        const float vir[6] = { mult * rc.x * rc.x, mult * rc.x * rc.y, mult * rc.x * rc.z, mult * rc.y * rc.y, mult * rc.y * rc.z,
                               mult * rc.z * rc.z };
        addWaveVirial(a.virial, vir);
    }
}

/* ---------------------------------------------------------------------------------------------------------------------- */
/* fused update (MI355X extension): force gather from the non-bonded buffer, integrator, SETTLE, and the coordinates of the next
 * step written straight into the non-bonded xq buffer.  One thread per update unit = one SETTLE water or one other atom. */

struct FusedUpdateArgs
{
    int                   numUnits;
    const int*            units; /* per unit 3 atoms (second -1: a single atom, -2: one with LINCS constraints), then their 3 grid slots */
    float3*               xp;    /* coordinates before the update of the atoms LINCS will constrain */
    float3*               x;
    float3*               v;
    const float3*         fAtom; /* forces in atom order to add (listed, long-range), or null */
    float3*               fNbat; /* non-bonded forces in grid order: read, then cleared for the next step */
    float*                xq;    /* float4 per grid slot: xyz rewritten, charge kept */
    const float*          inverseMasses;
    const unsigned short* groups;
    float                 dt;
    int                   tempScaling; /* as leapfrogKernel */
    TcLambdas             lambdaArgs;
    const float*          lambdas;
    int                   parrinelloRahman;
    float3                prDiagonal;
    SettlePars            pars;
    PbcAiuc               pbc;
    float*                virial;
    const float*          sdSigmaV;
    const float*          sdConstEm;
    const float*          table;
    int                   seed, step;
};

/* one wave per work-group: 32k waters are 500 waves, fewer than the chip has SIMDs, so they are spread over all CUs */
constexpr int c_fusedBlock = 64;

template<bool stochasticDynamics, bool computeVirial>
__launch_bounds__(c_fusedBlock) __global__ void fusedUpdateKernel(const FusedUpdateArgs a)
{
    const int u      = static_cast<int>(blockIdx.x) * c_fusedBlock + static_cast<int>(threadIdx.x);
    float     vir[6] = { 0, 0, 0, 0, 0, 0 };
    if (u < a.numUnits)
    {
        const int2* unit  = reinterpret_cast<const int2*>(a.units) + 3 * u; /* one 24-byte record: no dependent second load */
        const int2  u01   = unit[0], u2s0 = unit[1], s12 = unit[2];
        int         at[3] = { u01.x, u01.y, u2s0.x };
        int         slot[3] = { u2s0.y, s12.x, s12.y };
        const bool  water = (at[1] >= 0);
        const bool  keepOld = (at[1] == -2);
        if (!water)
        {
            at[1] = at[2] = at[0]; /* the extra two copies are computed and dropped */
            slot[1] = slot[2] = slot[0];
        }
        float3 xOld[3], xNew[3], vel[3];
        float  im[3];
#pragma unroll
        for (int k = 0; k < 3; k++)
        {
            float3 f = a.fNbat[slot[k]];
            if (a.fAtom) { f = f + a.fAtom[at[k]]; }
            xOld[k]        = a.x[at[k]];
            if (k == 0 && keepOld) { a.xp[at[0]] = xOld[0]; }
            const float3 v = a.v[at[k]];
            im[k]          = a.inverseMasses[at[k]];
            float3 vs      = v;
            if (!stochasticDynamics)
            {
                if (a.tempScaling != 0)
                {
                    const float lambda = (a.tempScaling == 1)   ? a.lambdaArgs.v[0]
                                         : (a.tempScaling == 2) ? a.lambdaArgs.v[a.groups[at[k]]]
                                                                : a.lambdas[a.groups[at[k]]];
                    vs                 = lambda * vs;
                }
                if (a.parrinelloRahman)
                {
                    vs.x -= a.prDiagonal.x * v.x;
                    vs.y -= a.prDiagonal.y * v.y;
                    vs.z -= a.prDiagonal.z * v.z;
                }
            }
            vel[k]  = vs + (im[k] * a.dt) * f;
            xNew[k] = xOld[k] + a.dt * vel[k];
        }
        /* every grid slot is cleared by the one unit that owns its atom */
        a.fNbat[slot[0]] = make_float3(0.0F, 0.0F, 0.0F);
        if (water)
        {
            a.fNbat[slot[1]] = make_float3(0.0F, 0.0F, 0.0F);
            a.fNbat[slot[2]] = make_float3(0.0F, 0.0F, 0.0F);
        }
        float3 dist21, dist31;
        if (water)
        {
            dist21 = pbcDxAiuc(a.pbc, xOld[1], xOld[0]);
            dist31 = pbcDxAiuc(a.pbc, xOld[2], xOld[0]);
            float3 dxO, dxH2, dxH3;
            settleTriangle(a.pars, dist21, dist31, pbcDxAiuc(a.pbc, xNew[1], xNew[0]), pbcDxAiuc(a.pbc, xNew[2], xNew[0]), dxO, dxH2, dxH3);
            const float invdt = 1.0F / a.dt;
            xNew[0] = xNew[0] + dxO;
            xNew[1] = xNew[1] + dxH2;
            xNew[2] = xNew[2] + dxH3;
            vel[0]  = vel[0] + invdt * dxO;
            vel[1]  = vel[1] + invdt * dxH2;
            vel[2]  = vel[2] + invdt * dxH3;
            if (computeVirial) { settleVirial(a.pars, xOld[0], dist21, dist31, dxO, dxH2, dxH3, vir); }
        }
        if (stochasticDynamics)
        {
#pragma unroll
            for (int k = 0; k < 3; k++)
            {
                const float3 xi  = langevinNoise(a.table, a.seed, a.step, at[k]);
                const int    g   = a.groups[at[k]];
                const float  em  = a.sdConstEm[g];
                const float  amp = sqrtf(im[k]) * a.sdSigmaV[g];
                const float3 vn  = vel[k];
                vel[k]           = em * vn + amp * xi;
                xNew[k]          = xNew[k] + (0.5F * a.dt) * (vel[k] - vn);
            }
            if (water)
            {
                float3 dxO, dxH2, dxH3;
                settleTriangle(a.pars, dist21, dist31, pbcDxAiuc(a.pbc, xNew[1], xNew[0]), pbcDxAiuc(a.pbc, xNew[2], xNew[0]), dxO, dxH2, dxH3);
                xNew[0] = xNew[0] + dxO;
                xNew[1] = xNew[1] + dxH2;
                xNew[2] = xNew[2] + dxH3;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++)
        {
            if (k == 0 || water)
            {
                a.x[at[k]]             = xNew[k];
                a.v[at[k]]             = vel[k];
                a.xq[4 * slot[k] + 0] = xNew[k].x;
                a.xq[4 * slot[k] + 1] = xNew[k].y;
                a.xq[4 * slot[k] + 2] = xNew[k].z;
            }
        }
    }
    if (computeVirial) { addWaveVirial(a.virial, vir); }
}

/* [XX XY XZ YY YZ ZZ] -> += into a row-major symmetric 3x3 */
void addSymmetricVirial(float* tensor, const float* six)
{
    tensor[0] += six[0];
    tensor[1] += six[1];
    tensor[2] += six[2];
    tensor[3] += six[1];
    tensor[4] += six[3];
    tensor[5] += six[4];
    tensor[6] += six[2];
    tensor[7] += six[4];
    tensor[8] += six[5];
}

template<typename T>
void uploadVector(T** d_ptr, size_t* d_alloc, const std::vector<T>& h, hipStream_t stream)
{
    if (h.size() > *d_alloc)
    {
        freeDeviceBuffer(d_ptr);
        *d_alloc = h.size() + h.size() / 5 + 64;
        allocateDeviceBuffer(d_ptr, *d_alloc);
    }
    if (!h.empty())
    {
        NBNXM_HIP_CHECK(hipMemcpyAsync(*d_ptr, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, stream));
        NBNXM_HIP_CHECK(hipStreamSynchronize(stream)); /* h is the caller's temporary */
    }
}

} // namespace

/* ====================================================================================================================== */

struct LeapFrogGpu
{
    DeviceStream        stream;
    int                 numTempScaleValues = 0, numAtoms = 0;
    float*              d_inverseMasses = nullptr;
    unsigned short*     d_groups        = nullptr;
    float*              d_lambdas       = nullptr;
    size_t              imAlloc = 0, groupsAlloc = 0;
    PinnedBuffer<float> h_lambdas;
};

struct SettleGpu
{
    DeviceStream        stream;
    SettlePars          pars{};
    int                 numSettles = 0;
    int*                d_atoms    = nullptr;
    size_t              atomsAlloc = 0;
    float*              d_virial   = nullptr;
    PinnedBuffer<float> h_virial;
};

struct LincsGpu
{
    DeviceStream stream;
    int          numIterations = 1, expansionOrder = 4;
    int          blockSize = 64, numBlocks = 0, maxCoupled = 0;
    int2*        d_constraints = nullptr;
    int2*        d_localSlots  = nullptr;
    float*       d_lengths     = nullptr;
    float2*      d_inverseMasses = nullptr;
    int*         d_coupledCounts = nullptr;
    int*         d_coupledIndices = nullptr;
    float*       d_massFactors = nullptr;
    int*         d_blockAtomStart = nullptr;
    int*         d_blockAtoms = nullptr;
    size_t       alloc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    float*       d_virial = nullptr;
    PinnedBuffer<float> h_virial;
    /* fused update: the constrained coordinates also go to the non-bonded xq buffer */
    std::vector<int> h_blockAtoms;
    int*             d_blockAtomSlots = nullptr;
    size_t           slotsAlloc = 0;
    float*           d_xq = nullptr;
};

struct UpdateConstrainGpu
{
    DeviceStream stream;
    LeapFrogGpu* leapFrog = nullptr;
    LangevinGpu* langevin = nullptr;
    LincsGpu*    lincs    = nullptr;
    SettleGpu*   settle   = nullptr;
    float3*      d_x = nullptr;
    float3*      d_v = nullptr;
    const float3* d_f = nullptr;
    float3*      d_xp = nullptr;
    int          numAtoms = 0, xpAlloc = 0, xpSize = 0;
    int          pbcType = 0;
    float        box[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    hipEvent_t   xUpdated = nullptr;
    bool         recordXUpdated = false; /* set once a consumer has asked for the event */
    /* fused update */
    int          numUnits = 0, numConstraints = 0;
    int*         d_units = nullptr; /* 6 ints per unit, complete after set_nbat_coupling */
    size_t       unitsAlloc = 0;
    std::vector<int> h_unitAtoms; /* 3 per unit */
    float*       d_xq = nullptr;
    float3*      d_fNbat = nullptr;
    float*       d_virial = nullptr;
    PinnedBuffer<float> h_virial;
};

/* temperature-scaling factors into kernel arguments (or the device buffer when there are too many groups) and the diagonal of
 * the Parrinello-Rahman matrix times its period; returns the tempScaling mode of the kernels */
static int prepareCoupling(LeapFrogGpu* lf, int doTemperatureScaling, const float* tcLambdas, int doParrinelloRahman, float dtPressureCouple,
                           const float* prMatrix, TcLambdas* lambdaArgs, float3* prDiagonal)
{
    NBNXM_ASSERT(!doTemperatureScaling || lf->numTempScaleValues > 0, "temperature coupling was requested with no temperature-coupling groups");
    hipStream_t s = lf->stream.stream;
    std::memset(lambdaArgs, 0, sizeof(*lambdaArgs));
    const bool lambdasInArgs = (lf->numTempScaleValues <= c_maxLambdasInArgs);
    if (doTemperatureScaling)
    {
        if (lambdasInArgs) { std::memcpy(lambdaArgs->v, tcLambdas, sizeof(float) * lf->numTempScaleValues); }
        else
        {
            /* the pinned staging copy may still be read by the previous step's transfer */
            NBNXM_HIP_CHECK(hipStreamSynchronize(s));
            std::memcpy(lf->h_lambdas.data, tcLambdas, sizeof(float) * lf->numTempScaleValues);
            NBNXM_HIP_CHECK(hipMemcpyAsync(lf->d_lambdas, lf->h_lambdas.data, sizeof(float) * lf->numTempScaleValues, hipMemcpyHostToDevice, s));
        }
    }
    *prDiagonal = make_float3(0.0F, 0.0F, 0.0F);
    if (doParrinelloRahman)
    {
        NBNXM_ASSERT(prMatrix[1] == 0 && prMatrix[2] == 0 && prMatrix[3] == 0 && prMatrix[5] == 0 && prMatrix[6] == 0 && prMatrix[7] == 0,
                     "fully anisotropic Parrinello-Rahman pressure coupling is not supported by the GPU leap-frog integrator");
        *prDiagonal = make_float3(dtPressureCouple * prMatrix[0], dtPressureCouple * prMatrix[4], dtPressureCouple * prMatrix[8]);
    }
    return !doTemperatureScaling ? 0 : (lf->numTempScaleValues == 1 ? 1 : (lambdasInArgs ? 2 : 3));
}

extern "C"
{

/* ---- leap-frog ------------------------------------------------------------------------------------------------------- */

LeapFrogGpu* leapfrog_gpu_create(void* stream, int numTempScaleValues)
{
    NBNXM_ASSERT(numTempScaleValues >= 0, "negative number of temperature-scaling values");
    LeapFrogGpu* lf = new LeapFrogGpu;
    lf->stream.init(stream);
    lf->numTempScaleValues = numTempScaleValues;
    if (numTempScaleValues > 0)
    {
        lf->h_lambdas.resize(numTempScaleValues);
        allocateDeviceBuffer(&lf->d_lambdas, numTempScaleValues);
    }
    return lf;
}

void leapfrog_gpu_free(LeapFrogGpu* lf)
{
    if (lf == nullptr) { return; }
    (void)hipStreamSynchronize(lf->stream.stream);
    freeDeviceBuffer(&lf->d_inverseMasses);
    freeDeviceBuffer(&lf->d_groups);
    freeDeviceBuffer(&lf->d_lambdas);
    lf->stream.destroy();
    delete lf;
}

void leapfrog_gpu_set(LeapFrogGpu* lf, int numAtoms, const float* inverseMasses, const unsigned short* tempScaleGroups)
{
    NBNXM_ASSERT(numAtoms >= 0, "negative number of atoms");
    lf->numAtoms = numAtoms;
    uploadVector(&lf->d_inverseMasses, &lf->imAlloc, std::vector<float>(inverseMasses, inverseMasses + numAtoms), lf->stream.stream);
    if (lf->numTempScaleValues > 1)
    {
        NBNXM_ASSERT(tempScaleGroups != nullptr, "temperature-scaling groups are needed with more than one value");
        for (int i = 0; i < numAtoms; i++) { NBNXM_ASSERT(tempScaleGroups[i] < lf->numTempScaleValues, "temperature-scaling group out of range"); }
        uploadVector(&lf->d_groups, &lf->groupsAlloc, std::vector<unsigned short>(tempScaleGroups, tempScaleGroups + numAtoms),
                     lf->stream.stream);
    }
}

void leapfrog_gpu_integrate(LeapFrogGpu* lf, void* d_x, void* d_xp, void* d_v, const void* d_f, float dt, int doTemperatureScaling,
                            const float* tcLambdas, int doParrinelloRahman, float dtPressureCouple, const float* prMatrix)
{
    NBNXM_ASSERT(lf->numAtoms > 0, "the number of atoms needs to be > 0");
    hipStream_t s = lf->stream.stream;
    TcLambdas   lambdaArgs;
    float3      prDiagonal;
    const int   mode = prepareCoupling(lf, doTemperatureScaling, tcLambdas, doParrinelloRahman, dtPressureCouple, prMatrix, &lambdaArgs, &prDiagonal);
    auto        k    = leapfrogKernel<0, false>;
    if (doParrinelloRahman)
    {
        k = (mode == 0) ? leapfrogKernel<0, true> : (mode == 1) ? leapfrogKernel<1, true> : (mode == 2) ? leapfrogKernel<2, true> : leapfrogKernel<3, true>;
    }
    else
    {
        k = (mode == 0) ? leapfrogKernel<0, false> : (mode == 1) ? leapfrogKernel<1, false> : (mode == 2) ? leapfrogKernel<2, false> : leapfrogKernel<3, false>;
    }
    const dim3 grid((lf->numAtoms + c_updateBlock - 1) / c_updateBlock);
    hipLaunchKernelGGL(k, grid, dim3(c_updateBlock), 0, s, lf->numAtoms, static_cast<float3*>(d_x), static_cast<float3*>(d_xp),
                       static_cast<float3*>(d_v), static_cast<const float3*>(d_f), lf->d_inverseMasses, dt, lambdaArgs, lf->d_lambdas, lf->d_groups,
                       prDiagonal);
    NBNXM_HIP_CHECK(hipGetLastError());
}

/* ---- SETTLE ---------------------------------------------------------------------------------------------------------- */

SettleGpu* settle_gpu_create(void* stream, float mO, float mH, float dOH, float dHH)
{
    NBNXM_ASSERT(mO > 0 && mH > 0 && dOH > 0 && dHH > 0 && dHH < 2 * dOH, "SETTLE needs positive masses and a triangle");
    SettleGpu* sg = new SettleGpu;
    sg->stream.init(stream);
    /* mdlib/settle.cpp:111-151, in double */
    const double wohh = static_cast<double>(mO) + 2.0 * mH;
    const double rc   = dHH / 2.0;
    const double h    = std::sqrt(static_cast<double>(dOH) * dOH - rc * rc);
    const double ra   = 2.0 * mH * h / wohh;
    sg->pars.mO       = mO;
    sg->pars.mH       = mH;
    sg->pars.wh       = static_cast<float>(mH / wohh);
    sg->pars.ra       = static_cast<float>(ra);
    sg->pars.rb       = static_cast<float>(h - ra);
    sg->pars.rc       = static_cast<float>(rc);
    sg->pars.irc2     = static_cast<float>(1.0 / dHH);
    allocateDeviceBuffer(&sg->d_virial, c_virialFloats);
    sg->h_virial.resize(c_virialFloats);
    return sg;
}

void settle_gpu_free(SettleGpu* sg)
{
    if (sg == nullptr) { return; }
    (void)hipStreamSynchronize(sg->stream.stream);
    freeDeviceBuffer(&sg->d_atoms);
    freeDeviceBuffer(&sg->d_virial);
    sg->stream.destroy();
    delete sg;
}

void settle_gpu_set(SettleGpu* sg, int numSettles, const int* atoms)
{
    NBNXM_ASSERT(numSettles >= 0, "negative number of SETTLEs");
    sg->numSettles = numSettles;
    uploadVector(&sg->d_atoms, &sg->atomsAlloc, std::vector<int>(atoms, atoms + 3 * static_cast<size_t>(numSettles)), sg->stream.stream);
}

void settle_gpu_apply(SettleGpu* sg, const void* d_x, void* d_xp, int updateVelocities, void* d_v, float invdt, int computeVirial,
                      float* virialScaled, int pbcType, const float* box)
{
    if (sg->numSettles == 0) { return; }
    NBNXM_ASSERT(!updateVelocities || d_v != nullptr, "velocities are needed to update them");
    hipStream_t s = sg->stream.stream;
    if (computeVirial) { NBNXM_HIP_CHECK(hipMemsetAsync(sg->d_virial, 0, c_virialFloats * sizeof(float), s)); }
    auto k = updateVelocities ? (computeVirial ? settleKernel<true, true> : settleKernel<true, false>)
                              : (computeVirial ? settleKernel<false, true> : settleKernel<false, false>);
    const dim3 grid((sg->numSettles + c_updateBlock - 1) / c_updateBlock);
    hipLaunchKernelGGL(k, grid, dim3(c_updateBlock), 0, s, sg->numSettles, sg->d_atoms, sg->pars, static_cast<const float3*>(d_x),
                       static_cast<float3*>(d_xp), invdt, static_cast<float3*>(d_v), sg->d_virial, makePbcAiuc(pbcType, box));
    NBNXM_HIP_CHECK(hipGetLastError());
    if (computeVirial)
    {
        NBNXM_HIP_CHECK(hipMemcpyAsync(sg->h_virial.data, sg->d_virial, c_virialFloats * sizeof(float), hipMemcpyDeviceToHost, s));
        NBNXM_HIP_CHECK(hipStreamSynchronize(s));
        float six[6];
        sumVirialSlots(sg->h_virial.data, six);
        addSymmetricVirial(virialScaled, six);
    }
}

/* ---- LINCS ----------------------------------------------------------------------------------------------------------- */

LincsGpu* lincs_gpu_create(void* stream, int numIterations, int expansionOrder)
{
    NBNXM_ASSERT(numIterations >= 0 && expansionOrder >= 0, "negative LINCS iteration count or expansion order");
    LincsGpu* lg = new LincsGpu;
    lg->stream.init(stream);
    lg->numIterations  = numIterations;
    lg->expansionOrder = expansionOrder;
    allocateDeviceBuffer(&lg->d_virial, c_virialFloats);
    lg->h_virial.resize(c_virialFloats);
    return lg;
}

void lincs_gpu_free(LincsGpu* lg)
{
    if (lg == nullptr) { return; }
    (void)hipStreamSynchronize(lg->stream.stream);
    freeDeviceBuffer(&lg->d_constraints);
    freeDeviceBuffer(&lg->d_localSlots);
    freeDeviceBuffer(&lg->d_lengths);
    freeDeviceBuffer(&lg->d_inverseMasses);
    freeDeviceBuffer(&lg->d_coupledCounts);
    freeDeviceBuffer(&lg->d_coupledIndices);
    freeDeviceBuffer(&lg->d_massFactors);
    freeDeviceBuffer(&lg->d_blockAtomStart);
    freeDeviceBuffer(&lg->d_blockAtoms);
    freeDeviceBuffer(&lg->d_blockAtomSlots);
    freeDeviceBuffer(&lg->d_virial);
    lg->stream.destroy();
    delete lg;
}

int lincs_gpu_set(LincsGpu* lg, int numConstraints, const int* iatoms, const float* lengths, int numAtoms, const float* inverseMasses)
{
    NBNXM_ASSERT(numConstraints >= 0, "negative number of constraints");
    if (numConstraints == 0)
    {
        lg->numBlocks = 0;
        lg->h_blockAtoms.clear();
        lg->d_xq = nullptr;
        return 0;
    }
    NBNXM_ASSERT(numAtoms > 0, "the number of atoms needs to be > 0 if there are constraints in the domain");
    /* which constraints end at which atom: CSR over atoms, entry = constraint * 2 + end */
    std::vector<int> atomStart(numAtoms + 1, 0);
    for (int c = 0; c < numConstraints; c++)
    {
        for (int e = 1; e <= 2; e++)
        {
            const int at = iatoms[3 * c + e];
            NBNXM_ASSERT(at >= 0 && at < numAtoms, "constrained atom outside the home atoms");
            atomStart[at + 1]++;
        }
    }
    std::partial_sum(atomStart.begin(), atomStart.end(), atomStart.begin());
    std::vector<int> atomEnds(2 * static_cast<size_t>(numConstraints));
    {
        std::vector<int> fill(atomStart.begin(), atomStart.end() - 1);
        for (int c = 0; c < numConstraints; c++)
        {
            for (int e = 1; e <= 2; e++) { atomEnds[fill[iatoms[3 * c + e]]++] = 2 * c + (e - 1); }
        }
    }
    /* groups of coupled constraints, in order of their first member; members in order of discovery so that neighbours
     * in the molecule are neighbours in the work-group */
    std::vector<int> order;
    order.reserve(numConstraints);
    std::vector<int>  groupStart(1, 0);
    std::vector<char> seen(numConstraints, 0);
    std::vector<int>  stack;
    for (int c0 = 0; c0 < numConstraints; c0++)
    {
        if (seen[c0]) { continue; }
        seen[c0] = 1;
        stack.assign(1, c0);
        while (!stack.empty())
        {
            const int c = stack.back();
            stack.pop_back();
            order.push_back(c);
            for (int e = 1; e <= 2; e++)
            {
                const int at = iatoms[3 * c + e];
                for (int k = atomStart[at]; k < atomStart[at + 1]; k++)
                {
                    const int c2 = atomEnds[k] >> 1;
                    if (!seen[c2])
                    {
                        seen[c2] = 1;
                        stack.push_back(c2);
                    }
                }
            }
        }
        groupStart.push_back(static_cast<int>(order.size()));
    }
    const int numGroups = static_cast<int>(groupStart.size()) - 1;
    int       maxGroup  = 0;
    for (int g = 0; g < numGroups; g++) { maxGroup = std::max(maxGroup, groupStart[g + 1] - groupStart[g]); }
    if (maxGroup > 1024) { return -1; }
    int B = 64;
    while (B < maxGroup) { B *= 2; }

    /* work-groups: a group that does not fit into the rest of one opens the next */
    std::vector<int> threadOf(numConstraints);
    int              next = 0;
    for (int g = 0; g < numGroups; g++)
    {
        const int size = groupStart[g + 1] - groupStart[g];
        if (next / B != (next + size - 1) / B) { next = (next / B + 1) * B; }
        for (int k = groupStart[g]; k < groupStart[g + 1]; k++) { threadOf[order[k]] = next++; }
    }
    const int numBlocks  = (next + B - 1) / B;
    const int numThreads = numBlocks * B;

    int maxCoupled = 0;
    for (int c = 0; c < numConstraints; c++)
    {
        const int a1 = iatoms[3 * c + 1], a2 = iatoms[3 * c + 2];
        maxCoupled   = std::max(maxCoupled, (atomStart[a1 + 1] - atomStart[a1]) + (atomStart[a2 + 1] - atomStart[a2]) - 2);
    }
    const size_t ldsBytes = static_cast<size_t>(B) * (11 + 2 * maxCoupled) * sizeof(float);
    if (ldsBytes > 64 * 1024) { return -1; }

    std::vector<int2>   h_constraints(numThreads, make_int2(-1, -1)), h_slots(numThreads, make_int2(0, 0));
    std::vector<float>  h_lengths(numThreads, 0.0F);
    std::vector<float2> h_im(numThreads, make_float2(0.0F, 0.0F));
    std::vector<int>    h_counts(numThreads, 0), h_indices(static_cast<size_t>(maxCoupled) * numThreads, 0);
    std::vector<float>  h_factors(static_cast<size_t>(maxCoupled) * numThreads, 0.0F);
    std::vector<int>    h_blockAtomStart(numBlocks + 1, 0), h_blockAtoms;
    std::vector<int>    slotOfAtom(numAtoms, -1);
    /* atoms of each work-group, in order of first use */
    std::vector<std::vector<int>> membersOfBlock(numBlocks);
    for (int c = 0; c < numConstraints; c++) { membersOfBlock[threadOf[c] / B].push_back(c); }
    for (int b = 0; b < numBlocks; b++)
    {
        std::sort(membersOfBlock[b].begin(), membersOfBlock[b].end(), [&](int p, int q) { return threadOf[p] < threadOf[q]; });
        h_blockAtomStart[b] = static_cast<int>(h_blockAtoms.size());
        for (int c : membersOfBlock[b])
        {
            int s[2];
            for (int e = 0; e < 2; e++)
            {
                const int at = iatoms[3 * c + 1 + e];
                if (slotOfAtom[at] < 0)
                {
                    slotOfAtom[at] = static_cast<int>(h_blockAtoms.size()) - h_blockAtomStart[b];
                    h_blockAtoms.push_back(at);
                }
                s[e] = slotOfAtom[at];
            }
            const int th       = threadOf[c];
            h_constraints[th]  = make_int2(iatoms[3 * c + 1], iatoms[3 * c + 2]);
            h_slots[th]        = make_int2(s[0], s[1]);
            h_lengths[th]      = lengths[iatoms[3 * c]];
            h_im[th]           = make_float2(inverseMasses[iatoms[3 * c + 1]], inverseMasses[iatoms[3 * c + 2]]);
        }
    }
    h_blockAtomStart[numBlocks] = static_cast<int>(h_blockAtoms.size());
    /* coupling coefficients -+ invmass(shared atom) sqrt(mu_c) sqrt(mu_c2): minus when the shared atom sits at the same end of
     * both constraints (lincs_gpu.cpp:371-428 with the sign factors of constraint_gpu_helpers.cpp) */
    for (int c = 0; c < numConstraints; c++)
    {
        const int    th  = threadOf[c];
        const double muC = 1.0 / std::sqrt(static_cast<double>(inverseMasses[iatoms[3 * c + 1]]) + inverseMasses[iatoms[3 * c + 2]]);
        for (int e = 0; e < 2; e++)
        {
            const int at = iatoms[3 * c + 1 + e];
            for (int k = atomStart[at]; k < atomStart[at + 1]; k++)
            {
                const int c2 = atomEnds[k] >> 1, e2 = atomEnds[k] & 1;
                if (c2 == c) { continue; }
                NBNXM_ASSERT(threadOf[c2] / B == th / B, "coupled constraints ended up in different work-groups");
                const double muC2 = 1.0 / std::sqrt(static_cast<double>(inverseMasses[iatoms[3 * c2 + 1]]) + inverseMasses[iatoms[3 * c2 + 2]]);
                const size_t idx  = static_cast<size_t>(h_counts[th]) * numThreads + th;
                h_indices[idx]    = threadOf[c2] % B;
                h_factors[idx]    = static_cast<float>(((e == e2) ? -1.0 : 1.0) * inverseMasses[at] * muC * muC2);
                h_counts[th]++;
            }
        }
    }

    hipStream_t s = lg->stream.stream;
    uploadVector(&lg->d_constraints, &lg->alloc[0], h_constraints, s);
    uploadVector(&lg->d_localSlots, &lg->alloc[1], h_slots, s);
    uploadVector(&lg->d_lengths, &lg->alloc[2], h_lengths, s);
    uploadVector(&lg->d_inverseMasses, &lg->alloc[3], h_im, s);
    uploadVector(&lg->d_coupledCounts, &lg->alloc[4], h_counts, s);
    uploadVector(&lg->d_coupledIndices, &lg->alloc[5], h_indices, s);
    uploadVector(&lg->d_massFactors, &lg->alloc[6], h_factors, s);
    uploadVector(&lg->d_blockAtomStart, &lg->alloc[7], h_blockAtomStart, s);
    uploadVector(&lg->d_blockAtoms, &lg->alloc[8], h_blockAtoms, s);
    lg->h_blockAtoms = h_blockAtoms;
    lg->d_xq         = nullptr; /* a coupling to the non-bonded buffers belongs to the previous search */
    lg->blockSize  = B;
    lg->numBlocks  = numBlocks;
    lg->maxCoupled = maxCoupled;
    return 0;
}

void lincs_gpu_apply(LincsGpu* lg, const void* d_x, void* d_xp, int updateVelocities, void* d_v, float invdt, int computeVirial,
                     float* virialScaled, int pbcType, const float* box)
{
    if (lg->numBlocks == 0) { return; }
    NBNXM_ASSERT(!updateVelocities || d_v != nullptr, "velocities are needed to update them");
    hipStream_t s = lg->stream.stream;
    if (computeVirial) { NBNXM_HIP_CHECK(hipMemsetAsync(lg->d_virial, 0, c_virialFloats * sizeof(float), s)); }
    LincsKernelArgs a;
    a.numThreads     = lg->numBlocks * lg->blockSize;
    a.numIterations  = lg->numIterations;
    a.expansionOrder = lg->expansionOrder;
    a.maxCoupled     = lg->maxCoupled;
    a.constraints    = lg->d_constraints;
    a.localSlots     = lg->d_localSlots;
    a.lengths        = lg->d_lengths;
    a.inverseMasses  = lg->d_inverseMasses;
    a.coupledCounts  = lg->d_coupledCounts;
    a.coupledIndices = lg->d_coupledIndices;
    a.massFactors    = lg->d_massFactors;
    a.blockAtomStart = lg->d_blockAtomStart;
    a.blockAtoms     = lg->d_blockAtoms;
    a.blockAtomSlots = (lg->d_xq != nullptr) ? lg->d_blockAtomSlots : nullptr;
    a.xq             = lg->d_xq;
    a.virial         = lg->d_virial;
    a.pbc            = makePbcAiuc(pbcType, box);
    auto k = updateVelocities ? (computeVirial ? lincsKernel<true, true> : lincsKernel<true, false>)
                              : (computeVirial ? lincsKernel<false, true> : lincsKernel<false, false>);
    const size_t ldsBytes = static_cast<size_t>(lg->blockSize) * (11 + 2 * lg->maxCoupled) * sizeof(float);
    hipLaunchKernelGGL(k, dim3(lg->numBlocks), dim3(lg->blockSize), ldsBytes, s, a, static_cast<const float3*>(d_x),
                       static_cast<float3*>(d_xp), static_cast<float3*>(d_v), invdt);
    NBNXM_HIP_CHECK(hipGetLastError());
    if (computeVirial)
    {
        NBNXM_HIP_CHECK(hipMemcpyAsync(lg->h_virial.data, lg->d_virial, c_virialFloats * sizeof(float), hipMemcpyDeviceToHost, s));
        NBNXM_HIP_CHECK(hipStreamSynchronize(s));
        float six[6];
        sumVirialSlots(lg->h_virial.data, six);
        addSymmetricVirial(virialScaled, six);
    }
}

/* ---- the composite ---------------------------------------------------------------------------------------------------- */

UpdateConstrainGpu* update_constrain_gpu_create(void* stream, const update_constrain_params_t* p)
{
    UpdateConstrainGpu* uc = new UpdateConstrainGpu;
    uc->stream.init(stream);
    void* s = uc->stream.stream;
    if (p->useStochasticDynamics)
    {
        uc->langevin = langevin_gpu_create(s, p->numTempCouplGroups, p->delta_t, p->ref_t, p->tau_t);
    }
    else { uc->leapFrog = leapfrog_gpu_create(s, p->numTempCouplGroups); }
    uc->lincs = lincs_gpu_create(s, p->nLincsIter, p->nProjOrder);
    if (p->haveSettle) { uc->settle = settle_gpu_create(s, p->mO, p->mH, p->dOH, p->dHH); }
    NBNXM_HIP_CHECK(hipEventCreateWithFlags(&uc->xUpdated, hipEventDisableTiming));
    allocateDeviceBuffer(&uc->d_virial, c_virialFloats);
    uc->h_virial.resize(c_virialFloats);
    return uc;
}

void update_constrain_gpu_free(UpdateConstrainGpu* uc)
{
    if (uc == nullptr) { return; }
    (void)hipStreamSynchronize(uc->stream.stream);
    leapfrog_gpu_free(uc->leapFrog);
    langevin_gpu_free(uc->langevin);
    lincs_gpu_free(uc->lincs);
    settle_gpu_free(uc->settle);
    freeDeviceBuffer(&uc->d_xp);
    freeDeviceBuffer(&uc->d_units);
    freeDeviceBuffer(&uc->d_virial);
    (void)hipEventDestroy(uc->xUpdated);
    uc->stream.destroy();
    delete uc;
}

int update_constrain_gpu_set(UpdateConstrainGpu* uc, void* d_x, void* d_v, const void* d_f, const update_constrain_topology_t* t)
{
    NBNXM_ASSERT(d_x && d_v && d_f, "coordinate, velocity and force device buffers should not be null");
    NBNXM_ASSERT(t->numSettles == 0 || uc->settle != nullptr, "SETTLEs in the domain but no SETTLE type in the topology");
    uc->d_x      = static_cast<float3*>(d_x);
    uc->d_v      = static_cast<float3*>(d_v);
    uc->d_f      = static_cast<const float3*>(d_f);
    uc->numAtoms = t->numAtoms;
    reallocateDeviceBuffer(&uc->d_xp, t->numAtoms, &uc->xpSize, &uc->xpAlloc);
    if (uc->leapFrog) { leapfrog_gpu_set(uc->leapFrog, t->numAtoms, t->inverseMasses, t->tempCouplGroups); }
    else { langevin_gpu_set(uc->langevin, t->numAtoms, t->inverseMasses, t->tempCouplGroups); }
    if (uc->settle) { settle_gpu_set(uc->settle, t->numSettles, t->settles); }
    /* update units of the fused path: the waters, then every atom that is not part of one */
    {
        std::vector<char> inWater(t->numAtoms, 0);
        std::vector<int>  units;
        units.reserve(3 * static_cast<size_t>(t->numAtoms));
        for (int w = 0; w < t->numSettles; w++)
        {
            for (int k = 0; k < 3; k++)
            {
                const int at = t->settles[3 * w + k];
                NBNXM_ASSERT(at >= 0 && at < t->numAtoms && !inWater[at], "SETTLE atom outside the home atoms or in two waters");
                inWater[at] = 1;
                units.push_back(at);
            }
        }
        std::vector<char> inLincs(t->numAtoms, 0);
        for (int c = 0; c < t->numConstraints; c++)
        {
            for (int e = 1; e <= 2; e++)
            {
                const int at = t->constraints[3 * c + e];
                NBNXM_ASSERT(at >= 0 && at < t->numAtoms && !inWater[at], "constrained atom outside the home atoms or part of a SETTLE water");
                inLincs[at] = 1;
            }
        }
        for (int at = 0; at < t->numAtoms; at++)
        {
            if (!inWater[at])
            {
                units.push_back(at);
                units.push_back(inLincs[at] ? -2 : -1);
                units.push_back(-1);
            }
        }
        uc->numUnits = static_cast<int>(units.size() / 3);
        uc->h_unitAtoms.swap(units);
        uc->d_xq = nullptr; /* the coupling belongs to the previous search */
    }
    uc->numConstraints = t->numConstraints;
    return lincs_gpu_set(uc->lincs, t->numConstraints, t->constraints, t->constraintLengths, t->numAtoms, t->inverseMasses);
}

void update_constrain_gpu_set_pbc(UpdateConstrainGpu* uc, int pbcType, const float* box)
{
    uc->pbcType = pbcType;
    std::memcpy(uc->box, box, sizeof(uc->box));
}

void update_constrain_gpu_integrate(UpdateConstrainGpu* uc, void* fReadyEvent, float dt, int updateVelocities, int computeVirial,
                                    float* virial, int doTemperatureScaling, const float* tcLambdas, int doParrinelloRahman,
                                    float dtPressureCouple, const float* prVelocityScalingMatrix, int seed, int step)
{
    hipStream_t s = uc->stream.stream;
    if (virial) { std::fill(virial, virial + 9, 0.0F); }
    NBNXM_ASSERT(!computeVirial || virial != nullptr, "a virial tensor is needed to compute the virial");
    if (fReadyEvent) { NBNXM_HIP_CHECK(hipStreamWaitEvent(s, static_cast<hipEvent_t>(fReadyEvent), 0)); }
    if (uc->numAtoms != 0)
    {
        /* the integrators leave the coordinates before the update in d_xp and the updated ones in d_x: the constraints take
         * them in that ("wrong") order, and no copy back is needed (update_constrain_gpu_impl.cpp:103-121) */
        if (uc->leapFrog)
        {
            leapfrog_gpu_integrate(uc->leapFrog, uc->d_x, uc->d_xp, uc->d_v, uc->d_f, dt, doTemperatureScaling, tcLambdas, doParrinelloRahman,
                                   dtPressureCouple, prVelocityScalingMatrix);
        }
        else { langevin_gpu_integrate(uc->langevin, uc->d_x, uc->d_xp, uc->d_v, uc->d_f, dt, seed, step, LANGEVIN_FORCES_ONLY); }
        lincs_gpu_apply(uc->lincs, uc->d_xp, uc->d_x, updateVelocities, uc->d_v, 1.0F / dt, computeVirial, virial, uc->pbcType, uc->box);
        if (uc->settle)
        {
            settle_gpu_apply(uc->settle, uc->d_xp, uc->d_x, updateVelocities, uc->d_v, 1.0F / dt, computeVirial, virial, uc->pbcType, uc->box);
        }
        if (uc->langevin)
        {
            langevin_gpu_integrate(uc->langevin, uc->d_x, uc->d_xp, uc->d_v, uc->d_f, dt, seed, step, LANGEVIN_FRICTION_AND_NOISE);
            /* constrain the coordinates again for the half step; velocities and virial are left alone */
            lincs_gpu_apply(uc->lincs, uc->d_xp, uc->d_x, 0, nullptr, 1.0F / (0.5F * dt), 0, nullptr, uc->pbcType, uc->box);
            if (uc->settle) { settle_gpu_apply(uc->settle, uc->d_xp, uc->d_x, 0, nullptr, 1.0F / (0.5F * dt), 0, nullptr, uc->pbcType, uc->box); }
        }
        if (computeVirial)
        {
            const float scale = 0.5F / (dt * dt);
            for (int i = 0; i < 9; i++) { virial[i] *= scale; }
        }
    }
    if (uc->recordXUpdated) { NBNXM_HIP_CHECK(hipEventRecord(uc->xUpdated, s)); }
}

/* ---- fused update (MI355X extension) ---------------------------------------------------------------------------------- */

void update_constrain_gpu_set_nbat_coupling(UpdateConstrainGpu* uc, const int* cell, void* d_xq, void* d_f_nbat)
{
    NBNXM_ASSERT(cell && d_xq && d_f_nbat, "the atom -> grid-slot map and the non-bonded xq and f buffers are needed");
    for (int i = 0; i < uc->numAtoms; i++) { NBNXM_ASSERT(cell[i] >= 0, "every home atom needs a grid slot"); }
    std::vector<int> units(6 * static_cast<size_t>(uc->numUnits));
    for (int u = 0; u < uc->numUnits; u++)
    {
        for (int k = 0; k < 3; k++)
        {
            const int at     = uc->h_unitAtoms[3 * u + k];
            units[6 * u + k] = at; /* second entry of a single atom: -1, or -2 when LINCS constrains it */
            units[6 * u + 3 + k] = (at >= 0) ? cell[at] : -1;
        }
    }
    uploadVector(&uc->d_units, &uc->unitsAlloc, units, uc->stream.stream);
    /* LINCS writes the constrained coordinates of its atoms into xq as well */
    {
        LincsGpu*        lg = uc->lincs;
        std::vector<int> slots(lg->h_blockAtoms.size());
        for (size_t i = 0; i < slots.size(); i++) { slots[i] = cell[lg->h_blockAtoms[i]]; }
        uploadVector(&lg->d_blockAtomSlots, &lg->slotsAlloc, slots, uc->stream.stream);
        lg->d_xq = (lg->numBlocks > 0) ? static_cast<float*>(d_xq) : nullptr;
    }
    uc->d_xq    = static_cast<float*>(d_xq);
    uc->d_fNbat = static_cast<float3*>(d_f_nbat);
}

int update_constrain_gpu_can_fuse(const UpdateConstrainGpu* uc)
{
    /* stochastic dynamics with LINCS needs the friction step between two constraint passes over memory: kernel sequence */
    return (uc->d_xq != nullptr && (uc->numConstraints == 0 || uc->leapFrog != nullptr)) ? 1 : 0;
}

void update_constrain_gpu_integrate_fused(UpdateConstrainGpu* uc, void* fReadyEvent, float dt, int computeVirial, float* virial,
                                          int doTemperatureScaling, const float* tcLambdas, int doParrinelloRahman, float dtPressureCouple,
                                          const float* prVelocityScalingMatrix, int seed, int step, int addAtomOrderForces)
{
    NBNXM_ASSERT(update_constrain_gpu_can_fuse(uc), "the fused update needs the non-bonded coupling (and leap-frog when there are LINCS constraints)");
    NBNXM_ASSERT(!computeVirial || virial != nullptr, "a virial tensor is needed to compute the virial");
    hipStream_t s = uc->stream.stream;
    if (virial) { std::fill(virial, virial + 9, 0.0F); }
    if (fReadyEvent) { NBNXM_HIP_CHECK(hipStreamWaitEvent(s, static_cast<hipEvent_t>(fReadyEvent), 0)); }
    if (uc->numUnits != 0)
    {
        FusedUpdateArgs a;
        std::memset(&a, 0, sizeof(a));
        a.numUnits = uc->numUnits;
        a.units    = uc->d_units;
        a.x        = uc->d_x;
        a.xp       = uc->d_xp;
        a.v        = uc->d_v;
        a.fAtom    = addAtomOrderForces ? uc->d_f : nullptr;
        a.fNbat    = uc->d_fNbat;
        a.xq       = uc->d_xq;
        a.dt       = dt;
        a.pbc      = makePbcAiuc(uc->pbcType, uc->box);
        a.virial   = uc->d_virial;
        a.seed     = seed;
        a.step     = step;
        if (uc->settle) { a.pars = uc->settle->pars; }
        if (uc->leapFrog)
        {
            a.inverseMasses    = uc->leapFrog->d_inverseMasses;
            a.groups           = uc->leapFrog->d_groups;
            a.lambdas          = uc->leapFrog->d_lambdas;
            a.parrinelloRahman = doParrinelloRahman;
            a.tempScaling      = prepareCoupling(uc->leapFrog, doTemperatureScaling, tcLambdas, doParrinelloRahman, dtPressureCouple,
                                                 prVelocityScalingMatrix, &a.lambdaArgs, &a.prDiagonal);
        }
        else
        {
            a.inverseMasses = uc->langevin->d_inverseMasses;
            a.groups        = uc->langevin->d_tcGroups;
            a.sdSigmaV      = uc->langevin->d_sdSigmaV;
            a.sdConstEm     = uc->langevin->d_sdConstEm;
            a.table         = uc->langevin->d_table;
        }
        if (computeVirial) { NBNXM_HIP_CHECK(hipMemsetAsync(uc->d_virial, 0, c_virialFloats * sizeof(float), s)); }
        auto k = uc->langevin ? (computeVirial ? fusedUpdateKernel<true, true> : fusedUpdateKernel<true, false>)
                              : (computeVirial ? fusedUpdateKernel<false, true> : fusedUpdateKernel<false, false>);
        const dim3 grid((uc->numUnits + c_fusedBlock - 1) / c_fusedBlock);
        hipLaunchKernelGGL(k, grid, dim3(c_fusedBlock), 0, s, a);
        NBNXM_HIP_CHECK(hipGetLastError());
        /* the atoms LINCS constrains: old coordinates in d_xp, updated ones in d_x, as after the integrator kernel */
        lincs_gpu_apply(uc->lincs, uc->d_xp, uc->d_x, 1, uc->d_v, 1.0F / dt, computeVirial, virial, uc->pbcType, uc->box);
        if (computeVirial)
        {
            NBNXM_HIP_CHECK(hipMemcpyAsync(uc->h_virial.data, uc->d_virial, c_virialFloats * sizeof(float), hipMemcpyDeviceToHost, s));
            NBNXM_HIP_CHECK(hipStreamSynchronize(s));
            float six[6];
            sumVirialSlots(uc->h_virial.data, six);
            addSymmetricVirial(virial, six);
            const float scale = 0.5F / (dt * dt);
            for (int i = 0; i < 9; i++) { virial[i] *= scale; }
        }
    }
    if (uc->recordXUpdated) { NBNXM_HIP_CHECK(hipEventRecord(uc->xUpdated, s)); }
}

static void scaleBuffer(UpdateConstrainGpu* uc, float3* d_buf, const float* m)
{
    if (uc->numAtoms == 0) { return; }
    const dim3 grid((uc->numAtoms + c_updateBlock - 1) / c_updateBlock);
    hipLaunchKernelGGL(scaleKernel, grid, dim3(c_updateBlock), 0, uc->stream.stream, uc->numAtoms, d_buf, m[0], m[4], m[8], m[3], m[6], m[7]);
    NBNXM_HIP_CHECK(hipGetLastError());
    NBNXM_HIP_CHECK(hipStreamSynchronize(uc->stream.stream));
}

void update_constrain_gpu_scale_coordinates(UpdateConstrainGpu* uc, const float* scalingMatrix) { scaleBuffer(uc, uc->d_x, scalingMatrix); }
void update_constrain_gpu_scale_velocities(UpdateConstrainGpu* uc, const float* scalingMatrix) { scaleBuffer(uc, uc->d_v, scalingMatrix); }

void* update_constrain_gpu_x_updated_event(UpdateConstrainGpu* uc)
{
    /* an event record costs a ~5 us bubble in front of the next kernel of the stream (measured): only pay it for a consumer */
    uc->recordXUpdated = true;
    return uc->xUpdated;
}

} // extern "C"
