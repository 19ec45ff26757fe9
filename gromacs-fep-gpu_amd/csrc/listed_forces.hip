/*
 * Perturbed listed (bonded) interactions on gfx950 — C ABI include/listed_hip.h.
 *
 * One launch covers every function type: the interactions of all types form one flat index space (prefix table by
 * value in the kernel arguments), one thread per interaction, 256-thread workgroups.  These are gather / few-flop /
 * scatter kernels: per interaction 2-4 float4 coordinate gathers (64 B), one parameter record, 2-4 x 12 B of force
 * atomics — latency / atomic bound, never VALU bound; the layout choices that matter are the coalesced index loads
 * (iatoms as int2..int5 rows read by consecutive lanes) and LDS staging of the per-workgroup energy and shift-force
 * sums so that a workgroup issues at most 7 + 135 global atomics for them.
 *
 * Semantics: listed_forces_gpu_internal.cu:781-1363 (harmonic_fep_gpu, bonds_fep_gpu, angles_fep_gpu,
 * urey_bradley_fep_gpu, dopdihs_fep_gpu, pdihs_fep_gpu, rbdihs_fep_gpu, idihs_fep_gpu, do_dih_fup_gpu :415-473,
 * dih_angle_gpu :374-402, bond_angle_gpu :166-185) = the CPU kernels of listed_forces/bonded.cpp with lambda.
 */
#include <hip/hip_runtime.h>

#include <cstring>

#include "device_utils.h"
#include "listed_hip.h"

using namespace nbnxm_hip;

namespace
{

constexpr int   c_listedBlock  = 256;
constexpr int   c_centralShift = 22;
constexpr int   c_numShifts    = 45;
constexpr float c_deg2rad      = 0.017453292519943295F;
constexpr float c_pi           = 3.14159265358979323846F;

struct PbcAiuc /* pbcutil/pbc_aiuc.h:67-96 */
{
    float invBoxDiagZ, boxZX, boxZY, boxZZ, invBoxDiagY, boxYX, boxYY, invBoxDiagX, boxXX;
};

struct ListedKernelArgs
{
    int                         start[LISTED_GPU_NUM_TYPES + 1]; /* prefix over the interactions of all types */
    const int*                  iatoms[LISTED_GPU_NUM_TYPES];
    const listed_gpu_iparams_t* params;
    const float4*               xq;
    const float4*               q4;
    float*                      f;
    float*                      fshift;
    float*                      epot; /* c_numOut: the energy terms, then the dV/dlambda components */
    PbcAiuc                     pbc;
    float                       lambda; /* lambda_bonded */
    listed_gpu_fep_params_t     fep;
    float                       elecScale;
    float                       epsfac;
};
constexpr int c_numOut = LISTED_GPU_NUM_ENERGY_TERMS + LISTED_GPU_NUM_DVDL;

__device__ __forceinline__ float3 operator-(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ float3 operator*(float s, float3 a) { return make_float3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float  dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float3 cross3(float3 a, float3 b)
{
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* minimum image (pbcDxAiuc, pbc_aiuc_cuda.cuh:70-128); returns the shift index of the image */
template<bool returnShift>
__device__ __forceinline__ int pbcDx(const PbcAiuc& pbc, float4 a, float4 b, float3& dx)
{
    dx              = make_float3(a.x - b.x, a.y - b.y, a.z - b.z);
    const float shz = rintf(dx.z * pbc.invBoxDiagZ);
    dx.x -= shz * pbc.boxZX;
    dx.y -= shz * pbc.boxZY;
    dx.z -= shz * pbc.boxZZ;
    const float shy = rintf(dx.y * pbc.invBoxDiagY);
    dx.x -= shy * pbc.boxYX;
    dx.y -= shy * pbc.boxYY;
    const float shx = rintf(dx.x * pbc.invBoxDiagX);
    dx.x -= shx * pbc.boxXX;
    if (returnShift) { return (1 - static_cast<int>(shz)) * 15 + (1 - static_cast<int>(shy)) * 5 + (2 - static_cast<int>(shx)); }
    return 0;
}

__device__ __forceinline__ void addForce(float* f, int atom, float3 v)
{
    atomicAdd(&f[3 * atom + 0], v.x);
    atomicAdd(&f[3 * atom + 1], v.y);
    atomicAdd(&f[3 * atom + 2], v.z);
}
__device__ __forceinline__ void addShift(float* sm, int idx, float3 v)
{
    atomicAdd(&sm[3 * idx + 0], v.x);
    atomicAdd(&sm[3 * idx + 1], v.y);
    atomicAdd(&sm[3 * idx + 2], v.z);
}

/* V = 1/2 k (x - x0)^2, k and x0 interpolated between the states; returns dV/dlambda */
__device__ __forceinline__ float harmonicFep(float kA, float kB, float xA, float xB, float x, float lambda, float& V, float& F)
{
    const float L1 = 1.0F - lambda;
    const float kk = L1 * kA + lambda * kB;
    const float x0 = L1 * xA + lambda * xB;
    const float dx = x - x0;
    F              = -kk * dx;
    V              = 0.5F * kk * dx * dx;
    return 0.5F * (kB - kA) * dx * dx + (xA - xB) * kk * dx;
}

template<bool calcVir>
__device__ __forceinline__ void bondPair(const ListedKernelArgs& a, float* smShift, int ai, int aj, float kA, float kB, float rA, float rB,
                                         float& epot, float& dvdl)
{
    float3      dx;
    const int   ki  = pbcDx<calcVir>(a.pbc, a.xq[ai], a.xq[aj], dx);
    const float dr2 = dot3(dx, dx);
    const float dr  = sqrtf(dr2);
    float       vb, fb;
    dvdl += harmonicFep(kA, kB, rA, rB, dr, a.lambda, vb, fb);
    epot += vb;
    if (dr2 != 0.0F)
    {
        const float3 fij = (fb * rsqrtf(dr2)) * dx;
        addForce(a.f, ai, fij);
        addForce(a.f, aj, -1.0F * fij);
        if (calcVir && ki != c_centralShift)
        {
            addShift(smShift, ki, fij);
            addShift(smShift, c_centralShift, -1.0F * fij);
        }
    }
}

template<bool calcVir>
__device__ __forceinline__ void angleTriple(const ListedKernelArgs& a, float* smShift, int ai, int aj, int ak, float kA, float kB, float thA,
                                            float thB, float& epot, float& dvdl)
{
    float3      r_ij, r_kj;
    const int   t1    = pbcDx<calcVir>(a.pbc, a.xq[ai], a.xq[aj], r_ij);
    const int   t2    = pbcDx<calcVir>(a.pbc, a.xq[ak], a.xq[aj], r_kj);
    const float nrij2 = dot3(r_ij, r_ij), nrkj2 = dot3(r_kj, r_kj);
    float       costh = dot3(r_ij, r_kj) * rsqrtf(nrij2 * nrkj2);
    costh             = fminf(1.0F, fmaxf(-1.0F, costh));
    const float theta = acosf(costh);
    float       va, dVdt;
    dvdl += harmonicFep(kA, kB, thA * c_deg2rad, thB * c_deg2rad, theta, a.lambda, va, dVdt);
    epot += va;
    const float c2 = costh * costh;
    if (c2 < 1.0F)
    {
        const float  st  = dVdt * rsqrtf(1.0F - c2);
        const float  sth = st * costh;
        const float  cik = st * rsqrtf(nrij2 * nrkj2), cii = sth / nrij2, ckk = sth / nrkj2;
        const float3 f_i = cii * r_ij - cik * r_kj;
        const float3 f_k = ckk * r_kj - cik * r_ij;
        const float3 f_j = -1.0F * f_i - f_k;
        addForce(a.f, ai, f_i);
        addForce(a.f, aj, f_j);
        addForce(a.f, ak, f_k);
        if (calcVir)
        {
            addShift(smShift, t1, f_i);
            addShift(smShift, c_centralShift, f_j);
            addShift(smShift, t2, f_k);
        }
    }
}

/* IUPAC dihedral angle and the vectors its force needs */
template<bool calcVir>
__device__ __forceinline__ float dihedralAngle(const ListedKernelArgs& a, int ai, int aj, int ak, int al, float3& r_ij, float3& r_kj,
                                               float3& r_kl, float3& m, float3& n, int& t1, int& t2)
{
    t1             = pbcDx<calcVir>(a.pbc, a.xq[ai], a.xq[aj], r_ij);
    t2             = pbcDx<calcVir>(a.pbc, a.xq[ak], a.xq[aj], r_kj);
    (void)pbcDx<false>(a.pbc, a.xq[ak], a.xq[al], r_kl);
    m              = cross3(r_ij, r_kj);
    n              = cross3(r_kj, r_kl);
    const float3 w = cross3(m, n);
    const float  phi = atan2f(sqrtf(dot3(w, w)), dot3(m, n));
    return (dot3(r_ij, n) < 0.0F) ? -phi : phi;
}

template<bool calcVir>
__device__ __forceinline__ void dihedralForces(const ListedKernelArgs& a, float* smShift, int ai, int aj, int ak, int al, float ddphi,
                                               float3 r_ij, float3 r_kj, float3 r_kl, float3 m, float3 n, int t1, int t2)
{
    const float iprm = dot3(m, m), iprn = dot3(n, n), nrkj2 = dot3(r_kj, r_kj);
    const float toler = nrkj2 * 1.1920929e-07F;
    if (iprm > toler && iprn > toler)
    {
        const float  nrkj_1 = rsqrtf(nrkj2);
        const float  nrkj   = nrkj2 * nrkj_1;
        const float3 f_i    = (-ddphi * nrkj / iprm) * m;
        const float3 f_l    = (ddphi * nrkj / iprn) * n;
        const float  p      = dot3(r_ij, r_kj) * nrkj_1 * nrkj_1;
        const float  q      = dot3(r_kl, r_kj) * nrkj_1 * nrkj_1;
        const float3 svec   = p * f_i - q * f_l;
        const float3 f_j    = f_i - svec;
        const float3 f_k    = make_float3(f_l.x + svec.x, f_l.y + svec.y, f_l.z + svec.z);
        addForce(a.f, ai, f_i);
        addForce(a.f, aj, -1.0F * f_j);
        addForce(a.f, ak, -1.0F * f_k);
        addForce(a.f, al, f_l);
        if (calcVir)
        {
            float3    dx_jl;
            const int t3 = pbcDx<true>(a.pbc, a.xq[al], a.xq[aj], dx_jl);
            addShift(smShift, t1, f_i);
            addShift(smShift, c_centralShift, -1.0F * f_j);
            addShift(smShift, t2, -1.0F * f_k);
            addShift(smShift, t3, f_l);
        }
    }
}

/* Perturbed 1-4 pair: Beutler soft-core LJ + Coulomb between the A and B states, no cut-off, no tables
 * (pairs_fep_gpu, listed_forces_gpu_internal.cu:1365-1600 = free_energy_evaluate_single, listed_forces/pairs.cpp:130-330,
 * with the table look-ups replaced by the functions they tabulate). */
template<bool calcVir>
__device__ __forceinline__ void pair14(const ListedKernelArgs& a, float* smShift, int ai, int aj, const float* p, float& eLJ, float& eCoul,
                                       float& dvdlVdw, float& dvdlCoul)
{
    const float4 qi = a.q4[ai], qj = a.q4[aj];
    const float  qq[2]  = { qi.x * qj.x, qi.y * qj.y };
    const float  c6[2]  = { p[0], p[2] };
    const float  c12[2] = { p[1], p[3] };
    float3       dr;
    const int    ki   = pbcDx<calcVir>(a.pbc, a.xq[ai], a.xq[aj], dr);
    const float  r2   = dot3(dr, dr);
    const float  rInv = rsqrtf(r2), rInv2 = rInv * rInv, rInv6 = rInv2 * rInv2 * rInv2;
    float        finvr;
    if (qq[0] == qq[1] && c6[0] == c6[1] && c12[0] == c12[1])
    {
        const float velec = a.elecScale * qq[0] * rInv;
        eCoul += velec;
        eLJ += (c12[0] * rInv6 - c6[0]) * rInv6;
        finvr = ((12.0F * c12[0] * rInv6 - 6.0F * c6[0]) * rInv6 + velec) * rInv2;
    }
    else
    {
        const listed_gpu_fep_params_t& fp = a.fep;
        const float rpm2 = r2 * r2, rp = rpm2 * r2;
        const bool  hard = (c12[0] > 0.0F && c12[1] > 0.0F);
        const float alphaV = hard ? 0.0F : fp.alphaVdw, alphaC = hard ? 0.0F : fp.alphaCoul;
        finvr              = 0.0F;
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            const float LFC = (k == 0) ? 1.0F - fp.lambdaCoul : fp.lambdaCoul;
            const float LFV = (k == 0) ? 1.0F - fp.lambdaVdw : fp.lambdaVdw;
            const float DLF = (k == 0) ? -1.0F : 1.0F;
            /* soft-core lambda factors and their derivative factors (sc-r-power 6) */
            const float scC  = (fp.lambdaPower == 2) ? (1.0F - LFC) * (1.0F - LFC) : (1.0F - LFC);
            const float scV  = (fp.lambdaPower == 2) ? (1.0F - LFV) * (1.0F - LFV) : (1.0F - LFV);
            const float dscC = DLF * fp.lambdaPower * (1.0F / 6.0F) * ((fp.lambdaPower == 2) ? (1.0F - LFC) : 1.0F);
            const float dscV = DLF * fp.lambdaPower * (1.0F / 6.0F) * ((fp.lambdaPower == 2) ? (1.0F - LFV) : 1.0F);
            float sigma6 = (c6[k] > 0.0F && c12[k] > 0.0F) ? fmaxf(c12[k] / c6[k], fp.sc_sigma6_min) : fp.sc_sigma6;
            float FC = 0.0F, FV = 0.0F, VC = 0.0F, VV = 0.0F;
            if (qq[k] != 0.0F || c6[k] != 0.0F || c12[k] != 0.0F)
            {
                const float rPInvC = 1.0F / (alphaC * scC * sigma6 + rp);
                const float rPInvV = 1.0F / (alphaV * scV * sigma6 + rp);
                const float rInvC  = sqrtf(cbrtf(rPInvC)); /* (r_C^-6)^(1/6) */
                const float V6 = c6[k] * rPInvV, V12 = c12[k] * rPInvV * rPInvV;
                VV = V12 - V6;
                FV = (12.0F * V12 - 6.0F * V6) * rPInvV;
                VC = a.elecScale * qq[k] * rInvC;
                FC = VC * rPInvC;
            }
            eCoul += LFC * VC;
            eLJ += LFV * VV;
            dvdlCoul += VC * DLF + LFC * alphaC * dscC * FC * sigma6;
            dvdlVdw += VV * DLF + LFV * alphaV * dscV * FV * sigma6;
            finvr += (LFC * FC + LFV * FV) * rpm2;
        }
    }
    const float3 f = finvr * dr;
    addForce(a.f, ai, f);
    addForce(a.f, aj, -1.0F * f);
    if (calcVir && ki != c_centralShift)
    {
        addShift(smShift, ki, f);
        addShift(smShift, c_centralShift, -1.0F * f);
    }
}

/* Unperturbed pair with its own charges and LJ parameters (pairs_gpu pType 1 and 2, listed_forces_gpu_internal.cu:700-778) */
template<bool calcVir>
__device__ __forceinline__ void simplePair(const ListedKernelArgs& a, float* smShift, int ai, int aj, float qq, float c6, float c12, float& eLJ,
                                           float& eCoul)
{
    float3      dr;
    const int   ki    = pbcDx<calcVir>(a.pbc, a.xq[ai], a.xq[aj], dr);
    const float r2    = dot3(dr, dr);
    const float rinv  = rsqrtf(r2), rinv2 = rinv * rinv, rinv6 = rinv2 * rinv2 * rinv2;
    const float velec = a.epsfac * qq * rinv;
    eCoul += velec;
    eLJ += (c12 * rinv6 - c6) * rinv6;
    const float3 f = (((12.0F * c12 * rinv6 - 6.0F * c6) * rinv6 + velec) * rinv2) * dr;
    addForce(a.f, ai, f);
    addForce(a.f, aj, -1.0F * f);
    if (calcVir && ki != c_centralShift)
    {
        addShift(smShift, ki, f);
        addShift(smShift, c_centralShift, -1.0F * f);
    }
}

/* Flat-bottomed distance restraint, linear beyond up2 (restraint_bonds_gpu, listed_forces_gpu_internal.cu:1605-1697;
 * bonded.cpp:619-712).  p: lowA up1A up2A kA lowB up1B up2B kB */
template<bool calcVir>
__device__ __forceinline__ void restraintBond(const ListedKernelArgs& a, float* smShift, int ai, int aj, const float* p, float lambda,
                                              float& epot, float& dvdl)
{
    float3      dx;
    const int   ki  = pbcDx<calcVir>(a.pbc, a.xq[ai], a.xq[aj], dx);
    const float dr2 = dot3(dx, dx), dr = sqrtf(dr2);
    const float L1  = 1.0F - lambda;
    const float low = L1 * p[0] + lambda * p[4], dlow = p[4] - p[0];
    const float up1 = L1 * p[1] + lambda * p[5], dup1 = p[5] - p[1];
    const float up2 = L1 * p[2] + lambda * p[6], dup2 = p[6] - p[2];
    const float k   = L1 * p[3] + lambda * p[7], dk = p[7] - p[3];
    float       vb = 0.0F, fb = 0.0F, dv = 0.0F;
    if (dr < low)
    {
        const float drh = dr - low;
        vb = 0.5F * k * drh * drh;
        fb = -k * drh;
        dv = 0.5F * dk * drh * drh - k * dlow * drh;
    }
    else if (dr <= up1) {}
    else if (dr <= up2)
    {
        const float drh = dr - up1;
        vb = 0.5F * k * drh * drh;
        fb = -k * drh;
        dv = 0.5F * dk * drh * drh - k * dup1 * drh;
    }
    else
    {
        const float drh = dr - up2;
        vb = k * (up2 - up1) * (0.5F * (up2 - up1) + drh);
        fb = -k * (up2 - up1);
        dv = dk * (up2 - up1) * (0.5F * (up2 - up1) + drh) + k * (dup2 - dup1) * (up2 - up1 + drh) - k * (up2 - up1) * dup2;
    }
    dvdl += dv;
    if (dr2 != 0.0F)
    {
        epot += vb;
        const float3 fij = (fb * rsqrtf(dr2)) * dx;
        addForce(a.f, ai, fij);
        addForce(a.f, aj, -1.0F * fij);
        if (calcVir && ki != c_centralShift)
        {
            addShift(smShift, ki, fij);
            addShift(smShift, c_centralShift, -1.0F * fij);
        }
    }
}

/* Angle restraint between the vectors i->j and k->l, V = cp (1 - cos(mult (phi - phi0))) (angleres_gpu,
 * listed_forces_gpu_internal.cu:1699-1777; bonded.cpp low_angres :2337-2420).  p: phiA cpA phiB cpB */
template<bool calcVir>
__device__ __forceinline__ void angleRestraint(const ListedKernelArgs& a, float* smShift, int ai, int aj, int ak, int al, const float* p, int mult,
                                               float lambda, float& epot, float& dvdl)
{
    float3      r_ij, r_kl;
    const int   t1   = pbcDx<calcVir>(a.pbc, a.xq[aj], a.xq[ai], r_ij);
    const int   t2   = pbcDx<calcVir>(a.pbc, a.xq[al], a.xq[ak], r_kl);
    const float nij2 = dot3(r_ij, r_ij), nkl2 = dot3(r_kl, r_kl);
    const float cosPhi = fminf(1.0F, fmaxf(-1.0F, dot3(r_ij, r_kl) * rsqrtf(nij2 * nkl2)));
    const float phi    = acosf(cosPhi);
    const float L1     = 1.0F - lambda;
    const float phi0   = (L1 * p[0] + lambda * p[2]) * c_deg2rad;
    const float dph0   = (p[2] - p[0]) * c_deg2rad;
    const float cp     = L1 * p[1] + lambda * p[3];
    float       s, c;
    sincosf(mult * (phi - phi0), &s, &c);
    const float dVdphi = cp * mult * s;
    dvdl += (p[3] - p[1]) * (1.0F - c) + cp * dph0 * s;
    epot += cp * (1.0F - c);
    const float cosPhi2 = cosPhi * cosPhi;
    if (cosPhi2 < 1.0F)
    {
        const float  st  = -dVdphi * rsqrtf(1.0F - cosPhi2);
        const float  sth = st * cosPhi;
        const float  cc  = st * rsqrtf(nij2 * nkl2);
        const float3 f_i = cc * r_kl - (sth / nij2) * r_ij;
        const float3 f_k = cc * r_ij - (sth / nkl2) * r_kl;
        addForce(a.f, ai, f_i);
        addForce(a.f, aj, -1.0F * f_i);
        addForce(a.f, ak, f_k);
        addForce(a.f, al, -1.0F * f_k);
        if (calcVir)
        {
            addShift(smShift, t1, f_i);
            addShift(smShift, c_centralShift, -1.0F * f_i);
            addShift(smShift, t2, f_k);
            addShift(smShift, c_centralShift, -1.0F * f_k);
        }
    }
}

template<bool calcVir, bool calcEner>
__launch_bounds__(c_listedBlock) __global__ void listedForcesKernel(const ListedKernelArgs a)
{
    __shared__ float smEner[c_numOut];
    __shared__ float smShift[3 * c_numShifts];
    if (calcEner && threadIdx.x < c_numOut) { smEner[threadIdx.x] = 0.0F; }
    if (calcVir)
    {
        for (int i = threadIdx.x; i < 3 * c_numShifts; i += c_listedBlock) { smShift[i] = 0.0F; }
    }
    if (calcVir || calcEner) { __syncthreads(); }

    const int tid = static_cast<int>(blockIdx.x) * c_listedBlock + static_cast<int>(threadIdx.x);
    if (tid < a.start[LISTED_GPU_NUM_TYPES])
    {
        int ftype = 0;
#pragma unroll
        for (int t = 1; t < LISTED_GPU_NUM_TYPES; t++) { ftype += (tid >= a.start[t]) ? 1 : 0; }
        const int  i    = tid - a.start[ftype];
        const int* ia   = a.iatoms[ftype];
        float      epot = 0.0F, dvdl = 0.0F;
        float      eCoul = 0.0F, dvdlCoul = 0.0F, dvdlVdw = 0.0F;
        if (ftype == LISTED_GPU_LJ14)
        {
            pair14<calcVir>(a, smShift, ia[3 * i + 1], ia[3 * i + 2], a.params[ia[3 * i]].p, epot, eCoul, dvdlVdw, dvdlCoul);
        }
        else if (ftype == LISTED_GPU_LJC14_Q || ftype == LISTED_GPU_LJC_PAIRS_NB)
        {
            const float* p  = a.params[ia[3 * i]].p;
            const bool   q14 = (ftype == LISTED_GPU_LJC14_Q);
            simplePair<calcVir>(a, smShift, ia[3 * i + 1], ia[3 * i + 2], q14 ? p[0] * p[1] * p[2] : p[0] * p[1], q14 ? p[3] : p[2],
                                q14 ? p[4] : p[3], epot, eCoul);
        }
        else if (ftype == LISTED_GPU_RESTRBONDS)
        {
            restraintBond<calcVir>(a, smShift, ia[3 * i + 1], ia[3 * i + 2], a.params[ia[3 * i]].p, a.fep.lambdaRestraint, epot, dvdl);
        }
        else if (ftype == LISTED_GPU_ANGRES)
        {
            const listed_gpu_iparams_t& ip = a.params[ia[5 * i]];
            angleRestraint<calcVir>(a, smShift, ia[5 * i + 1], ia[5 * i + 2], ia[5 * i + 3], ia[5 * i + 4], ip.p, ip.mult, a.fep.lambdaRestraint,
                                    epot, dvdl);
        }
        else if (ftype == LISTED_GPU_BONDS)
        {
            const float* p = a.params[ia[3 * i]].p;
            bondPair<calcVir>(a, smShift, ia[3 * i + 1], ia[3 * i + 2], p[1], p[3], p[0], p[2], epot, dvdl);
        }
        else if (ftype == LISTED_GPU_ANGLES)
        {
            const float* p = a.params[ia[4 * i]].p;
            angleTriple<calcVir>(a, smShift, ia[4 * i + 1], ia[4 * i + 2], ia[4 * i + 3], p[1], p[3], p[0], p[2], epot, dvdl);
        }
        else if (ftype == LISTED_GPU_UREY_BRADLEY)
        {
            const float* p = a.params[ia[4 * i]].p;
            angleTriple<calcVir>(a, smShift, ia[4 * i + 1], ia[4 * i + 2], ia[4 * i + 3], p[1], p[5], p[0], p[4], epot, dvdl);
            bondPair<calcVir>(a, smShift, ia[4 * i + 1], ia[4 * i + 3], p[3], p[7], p[2], p[6], epot, dvdl);
        }
        else
        {
            const listed_gpu_iparams_t& ip = a.params[ia[5 * i]];
            const float*                p  = ip.p;
            const int   ai = ia[5 * i + 1], aj = ia[5 * i + 2], ak = ia[5 * i + 3], al = ia[5 * i + 4];
            float3      r_ij, r_kj, r_kl, m, n;
            int         t1, t2;
            float       phi = dihedralAngle<calcVir>(a, ai, aj, ak, al, r_ij, r_kj, r_kl, m, n, t1, t2);
            const float L1  = 1.0F - a.lambda;
            float       ddphi;
            if (ftype == LISTED_GPU_DIHRES)
            {
                /* flat-bottomed dihedral restraint (dihres_gpu, listed_forces_gpu_internal.cu:1779-1872; bonded.cpp:2472-2560) */
                const float lr = a.fep.lambdaRestraint, l1 = 1.0F - lr;
                const float phi0A = p[0] * c_deg2rad, dphiA = p[1] * c_deg2rad, phi0B = p[3] * c_deg2rad, dphiB = p[4] * c_deg2rad;
                const float phi0 = l1 * phi0A + lr * phi0B, dphi = l1 * dphiA + lr * dphiB, kfac = l1 * p[2] + lr * p[5];
                float       dp   = phi - phi0;
                if (dp >= c_pi) { dp -= 2.0F * c_pi; }
                else if (dp < -c_pi) { dp += 2.0F * c_pi; }
                const float ddp = (dp > dphi) ? dp - dphi : ((dp < -dphi) ? dp + dphi : 0.0F);
                epot += 0.5F * kfac * ddp * ddp;
                dvdl += 0.5F * (p[5] - p[2]) * ddp * ddp;
                if (ddp > 0.0F) { dvdl -= kfac * ddp * ((dphiB - dphiA) + (phi0B - phi0A)); }
                else if (ddp < 0.0F) { dvdl += kfac * ddp * ((dphiB - dphiA) - (phi0B - phi0A)); }
                ddphi = kfac * ddp;
            }
            else if (ftype == LISTED_GPU_PDIHS)
            {
                const float phi0  = (L1 * p[0] + a.lambda * p[2]) * c_deg2rad;
                const float dph0  = (p[2] - p[0]) * c_deg2rad;
                const float cp    = L1 * p[1] + a.lambda * p[3];
                const float mdphi = ip.mult * phi - phi0;
                float       s, c;
                sincosf(mdphi, &s, &c);
                ddphi = -cp * ip.mult * s;
                dvdl += (p[3] - p[1]) * (1.0F + c) + cp * dph0 * s;
                epot += cp * (1.0F + c);
            }
            else if (ftype == LISTED_GPU_IDIHS)
            {
                const float kk   = L1 * p[1] + a.lambda * p[3];
                const float phi0 = (L1 * p[0] + a.lambda * p[2]) * c_deg2rad;
                const float dph0 = (p[2] - p[0]) * c_deg2rad;
                float       dp   = phi - phi0;
                if (dp >= c_pi) { dp -= 2.0F * c_pi; }
                else if (dp < -c_pi) { dp += 2.0F * c_pi; }
                dvdl += 0.5F * (p[3] - p[1]) * dp * dp - kk * dph0 * dp;
                epot += 0.5F * kk * dp * dp;
                ddphi = kk * dp;
            }
            else
            {
                /* Ryckaert-Bellemans, polymer convention psi = phi - pi */
                phi += (phi >= c_pi) ? -c_pi : c_pi;
                float s, c;
                sincosf(phi, &s, &c);
                float v = 0.0F, dd = 0.0F, cosfac = 1.0F;
#pragma unroll
                for (int j = 0; j < 6; j++)
                {
                    const float rbp = L1 * p[j] + a.lambda * p[6 + j];
                    if (j > 0)
                    {
                        dd += j * rbp * cosfac;
                        cosfac *= c;
                    }
                    v += cosfac * rbp;
                    dvdl += cosfac * (p[6 + j] - p[j]);
                }
                ddphi = -dd * s;
                epot += v;
            }
            dihedralForces<calcVir>(a, smShift, ai, aj, ak, al, ddphi, r_ij, r_kj, r_kl, m, n, t1, t2);
        }
        if (calcEner)
        {
            atomicAdd(&smEner[ftype], epot);
            if (ftype == LISTED_GPU_LJ14)
            {
                atomicAdd(&smEner[LISTED_GPU_ENERGY_COULOMB14], eCoul);
                atomicAdd(&smEner[LISTED_GPU_NUM_ENERGY_TERMS + LISTED_GPU_DVDL_COUL], dvdlCoul);
                atomicAdd(&smEner[LISTED_GPU_NUM_ENERGY_TERMS + LISTED_GPU_DVDL_VDW], dvdlVdw);
            }
            else if (ftype == LISTED_GPU_LJC14_Q) { atomicAdd(&smEner[LISTED_GPU_ENERGY_COULOMB14], eCoul); }
            else if (ftype == LISTED_GPU_LJC_PAIRS_NB) { atomicAdd(&smEner[LISTED_GPU_ENERGY_COULOMB_PAIRS_NB], eCoul); }
            else if (ftype == LISTED_GPU_RESTRBONDS || ftype == LISTED_GPU_ANGRES || ftype == LISTED_GPU_DIHRES)
            {
                atomicAdd(&smEner[LISTED_GPU_NUM_ENERGY_TERMS + LISTED_GPU_DVDL_RESTRAINT], dvdl);
            }
            else { atomicAdd(&smEner[LISTED_GPU_NUM_ENERGY_TERMS + LISTED_GPU_DVDL_BONDED], dvdl); }
        }
    }
    if (calcVir || calcEner) { __syncthreads(); }
    if (calcEner && threadIdx.x < c_numOut && smEner[threadIdx.x] != 0.0F) { atomicAdd(&a.epot[threadIdx.x], smEner[threadIdx.x]); }
    if (calcVir)
    {
        for (int i = threadIdx.x; i < 3 * c_numShifts; i += c_listedBlock)
        {
            if (smShift[i] != 0.0F) { atomicAdd(&a.fshift[i], smShift[i]); }
        }
    }
}

} // namespace

struct ListedGpu
{
    DeviceStream          stream;
    int                   numInteractions[LISTED_GPU_NUM_TYPES] = {};
    int*                  d_iatoms[LISTED_GPU_NUM_TYPES]        = {};
    int                   iatomsAlloc[LISTED_GPU_NUM_TYPES]     = {};
    listed_gpu_iparams_t* d_params                              = nullptr;
    int                   numParams = 0, paramsAlloc = 0;
    float*                d_epot = nullptr; /* c_numOut */
    PinnedBuffer<float>   h_epot;
    PinnedBuffer<int>     h_iatoms[LISTED_GPU_NUM_TYPES];
    PinnedBuffer<listed_gpu_iparams_t> h_params;
};

static int listedNral(int ftype)
{
    switch (ftype)
    {
        case LISTED_GPU_BONDS:
        case LISTED_GPU_LJ14:
        case LISTED_GPU_LJC14_Q:
        case LISTED_GPU_LJC_PAIRS_NB:
        case LISTED_GPU_RESTRBONDS: return 2;
        case LISTED_GPU_ANGLES:
        case LISTED_GPU_UREY_BRADLEY: return 3;
        default: return 4;
    }
}

extern "C"
{

ListedGpu* listed_gpu_create(void* stream)
{
    ListedGpu* lg = new ListedGpu;
    lg->stream.init(stream);
    allocateDeviceBuffer(&lg->d_epot, c_numOut);
    clearDeviceBufferAsync(&lg->d_epot, 0, c_numOut, lg->stream.stream);
    lg->h_epot.resize(c_numOut);
    NBNXM_HIP_CHECK(hipStreamSynchronize(lg->stream.stream));
    return lg;
}

void listed_gpu_free(ListedGpu* lg)
{
    if (lg == nullptr) { return; }
    (void)hipStreamSynchronize(lg->stream.stream);
    for (auto& p : lg->d_iatoms) { freeDeviceBuffer(&p); }
    freeDeviceBuffer(&lg->d_params);
    freeDeviceBuffer(&lg->d_epot);
    lg->stream.destroy();
    delete lg;
}

void listed_gpu_set_force_params(ListedGpu* lg, int numParams, const listed_gpu_iparams_t* params)
{
    if (numParams > lg->paramsAlloc)
    {
        freeDeviceBuffer(&lg->d_params);
        lg->paramsAlloc = numParams + 64;
        allocateDeviceBuffer(&lg->d_params, lg->paramsAlloc);
    }
    lg->h_params.resize(numParams);
    if (numParams) { std::memcpy(lg->h_params.data, params, sizeof(listed_gpu_iparams_t) * numParams); }
    copyToDeviceBuffer(&lg->d_params, lg->h_params.data, 0, numParams, lg->stream.stream, true);
    lg->numParams = numParams;
}

void listed_gpu_update_interaction_list(ListedGpu* lg, int ftype, int numInteractions, const int* iatoms, int numAtoms)
{
    NBNXM_ASSERT(ftype >= 0 && ftype < LISTED_GPU_NUM_TYPES, "unknown listed function type");
    const int stride = 1 + listedNral(ftype);
    /* shape checks on the host: an out-of-range index would fault on the device */
    for (int i = 0; i < numInteractions; i++)
    {
        NBNXM_ASSERT(iatoms[stride * i] >= 0 && iatoms[stride * i] < lg->numParams, "parameter index out of range (set the force parameters first)");
        for (int k = 1; k < stride; k++) { NBNXM_ASSERT(iatoms[stride * i + k] >= 0 && iatoms[stride * i + k] < numAtoms, "atom index out of range"); }
    }
    const int n = stride * numInteractions;
    if (n > lg->iatomsAlloc[ftype])
    {
        freeDeviceBuffer(&lg->d_iatoms[ftype]);
        lg->iatomsAlloc[ftype] = static_cast<int>(n * 1.2) + 64;
        allocateDeviceBuffer(&lg->d_iatoms[ftype], lg->iatomsAlloc[ftype]);
    }
    lg->h_iatoms[ftype].resize(n);
    if (n) { std::memcpy(lg->h_iatoms[ftype].data, iatoms, sizeof(int) * n); }
    copyToDeviceBuffer(&lg->d_iatoms[ftype], lg->h_iatoms[ftype].data, 0, n, lg->stream.stream, true);
    lg->numInteractions[ftype] = numInteractions;
}

int listed_gpu_have_interactions(const ListedGpu* lg)
{
    for (int n : lg->numInteractions)
    {
        if (n > 0) { return 1; }
    }
    return 0;
}

void listed_gpu_launch_kernel(ListedGpu* lg, const void* d_xq, const void* d_q4, void* d_f, void* d_fshift, const float* box,
                              int pbcType, const listed_gpu_fep_params_t* fep, float electrostaticsScaleFactor, float epsfac,
                              int computeEnergy, int computeVirial)
{
    if (!listed_gpu_have_interactions(lg)) { return; }
    NBNXM_ASSERT(d_xq != nullptr && d_f != nullptr, "coordinate / force buffer missing");
    NBNXM_ASSERT(fep != nullptr, "free-energy parameters missing");
    NBNXM_ASSERT(lg->numInteractions[LISTED_GPU_LJ14] == 0 || d_q4 != nullptr, "1-4 pairs need the A/B charge buffer (q4)");
    NBNXM_ASSERT(!computeVirial || d_fshift != nullptr, "virial step without a shift-force buffer");
    ListedKernelArgs a;
    a.start[0] = 0;
    for (int t = 0; t < LISTED_GPU_NUM_TYPES; t++)
    {
        a.start[t + 1] = a.start[t] + lg->numInteractions[t];
        a.iatoms[t]    = lg->d_iatoms[t];
    }
    a.params = lg->d_params;
    a.xq     = static_cast<const float4*>(d_xq);
    a.q4     = static_cast<const float4*>(d_q4);
    a.fep    = *fep;
    a.elecScale = electrostaticsScaleFactor;
    a.epsfac    = epsfac;
    a.f      = static_cast<float*>(d_f);
    a.fshift = static_cast<float*>(d_fshift);
    a.epot   = lg->d_epot;
    a.lambda = fep->lambdaBonded;
    /* setPbcAiuc (pbcutil/pbc_aiuc.h:98-140): dimensions without PBC get a zero inverse, which makes their shift 0 */
    const int npbcdim = (pbcType == 3) ? 3 : ((pbcType == 2) ? 2 : 0);
    a.pbc.invBoxDiagZ = (npbcdim > 2) ? 1.0F / box[8] : 0.0F;
    a.pbc.invBoxDiagY = (npbcdim > 1) ? 1.0F / box[4] : 0.0F;
    a.pbc.invBoxDiagX = (npbcdim > 0) ? 1.0F / box[0] : 0.0F;
    a.pbc.boxZX = box[6]; a.pbc.boxZY = box[7]; a.pbc.boxZZ = box[8];
    a.pbc.boxYX = box[3]; a.pbc.boxYY = box[4];
    a.pbc.boxXX = box[0];
    const int  total = a.start[LISTED_GPU_NUM_TYPES];
    const dim3 grid((total + c_listedBlock - 1) / c_listedBlock);
    auto       k = computeVirial ? (computeEnergy ? listedForcesKernel<true, true> : listedForcesKernel<true, false>)
                                 : (computeEnergy ? listedForcesKernel<false, true> : listedForcesKernel<false, false>);
    hipLaunchKernelGGL(k, grid, dim3(c_listedBlock), 0, lg->stream.stream, a);
    NBNXM_HIP_CHECK(hipGetLastError());
}

void listed_gpu_launch_energy_transfer(ListedGpu* lg)
{
    NBNXM_HIP_CHECK(hipMemcpyAsync(lg->h_epot.data, lg->d_epot, sizeof(float) * c_numOut, hipMemcpyDeviceToHost, lg->stream.stream));
}

void listed_gpu_wait_accumulate_energy_terms(ListedGpu* lg, double* epot, double* dvdl)
{
    NBNXM_HIP_CHECK(hipStreamSynchronize(lg->stream.stream));
    for (int t = 0; t < LISTED_GPU_NUM_ENERGY_TERMS; t++) { epot[t] += lg->h_epot.data[t]; }
    for (int t = 0; t < LISTED_GPU_NUM_DVDL; t++) { dvdl[t] += lg->h_epot.data[LISTED_GPU_NUM_ENERGY_TERMS + t]; }
}

void listed_gpu_clear_energies(ListedGpu* lg)
{
    clearDeviceBufferAsync(&lg->d_epot, 0, c_numOut, lg->stream.stream);
}

} // extern "C"
