/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Throughput port of the cluster-pair force kernel for the CPU baseline of bench.py (cpu_baseline.kind "port"): the same list
 * walk and pair semantics as oracle_nbnxm_ref_f32 (nbnxm_ref.c, which stays the parity oracle), specialised to the force-only
 * step of the benchmark — Ewald (analytical) or reaction-field / plain cut-off electrostatics, Lennard-Jones cut-off from the
 * type table — and vectorised the way the reference's CPU SIMD kernels are (nbnxm/kernels_simd_4xm): 8 i-atoms of a cluster
 * in one 8-float vector against one j-atom at a time, exclusions and the cut-off as lane masks, the Ewald correction as a
 * rational polynomial instead of erf/exp.  GCC vector extensions; this file alone is compiled with -mavx2 -mfma; the entry
 * point returns -1 without touching anything when the CPU or the flavour is not supported, and the caller falls back to the
 * scalar oracle.  Checked against the scalar oracle in tests/test_pairlist_cpu.py.
 */
#include "nbnxm_ref.h"

#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "../gromacs-fep-gpu_amd/csrc/pme_corr_coeffs.h" /* rational fit of the Ewald force correction (tools/fit_pme_corr.py) */

#define CL 8
#define NCL 8
#define MAX_TYPES 16

typedef float v8f __attribute__((vector_size(32)));
typedef int   v8i __attribute__((vector_size(32)));

static inline v8f bcast(float a) { return (v8f){ a, a, a, a, a, a, a, a }; }
static inline float hsum(v8f a) { return ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7])); }
static inline v8f vsel(v8i mask, v8f a, v8f b) { return (v8f)(((v8i)a & mask) | ((v8i)b & ~mask)); }

static inline v8f pme_corr_f(v8f z2)
{
    v8f num = bcast(PME_CORR_P5);
    num     = num * z2 + bcast(PME_CORR_P4);
    num     = num * z2 + bcast(PME_CORR_P3);
    num     = num * z2 + bcast(PME_CORR_P2);
    num     = num * z2 + bcast(PME_CORR_P1);
    num     = num * z2 + bcast(PME_CORR_P0);
    v8f den = bcast(PME_CORR_Q4);
    den     = den * z2 + bcast(PME_CORR_Q3);
    den     = den * z2 + bcast(PME_CORR_Q2);
    den     = den * z2 + bcast(PME_CORR_Q1);
    den     = den * z2 + bcast(PME_CORR_Q0);
    return num / den;
}

static void simd_range(int nsci, const nbnxn_sci_t* sci, const nbnxn_cj_packed_t* cjPacked,
                       const nbnxn_excl_t* excl, const float* xq, const int* type, int ntype, const float* nbfp,
                       const nbnxm_ref_params_t* p, const float* shiftvec, float* f)
{
    const int   ewald = (p->elecType == NBNXM_ELEC_EWALD_ANA);
    const int   rf    = (p->elecType == NBNXM_ELEC_RF);
    const v8f   rc2   = bcast((float)(p->rcoulomb * p->rcoulomb));
    const v8f   minr2 = bcast(3.82e-07F);
    const float beta  = (float)p->ewaldcoeff_q;
    const v8f   b2    = bcast(beta * beta), b3 = bcast(beta * beta * beta);
    const v8f   twoK  = bcast(rf ? 2.0F * (float)p->k_rf : 0.0F);
    const v8f   one = bcast(1.0F), zero = bcast(0.0F);
    const v8i   laneId = { 0, 1, 2, 3, 4, 5, 6, 7 };

    for (int s = 0; s < nsci; s++)
    {
        const nbnxn_sci_t e   = sci[s];
        const int         ish = e.shift & NBNXM_CI_SHIFT_MASK;
        v8f               xi[NCL], yi[NCL], zi[NCL], qi[NCL];
        v8f               c6t[NCL][MAX_TYPES], c12t[NCL][MAX_TYPES];
        v8f               fxi[NCL], fyi[NCL], fzi[NCL];
        for (int im = 0; im < NCL; im++)
        {
            for (int ic = 0; ic < CL; ic++)
            {
                const int ia = (e.sci * NCL + im) * CL + ic;
                xi[im][ic]   = xq[4 * ia + 0] + shiftvec[3 * ish + 0];
                yi[im][ic]   = xq[4 * ia + 1] + shiftvec[3 * ish + 1];
                zi[im][ic]   = xq[4 * ia + 2] + shiftvec[3 * ish + 2];
                qi[im][ic]   = (float)p->epsfac * xq[4 * ia + 3];
                for (int t = 0; t < ntype; t++)
                {
                    c6t[im][t][ic]  = nbfp[2 * (ntype * type[ia] + t)];
                    c12t[im][t][ic] = nbfp[2 * (ntype * type[ia] + t) + 1];
                }
            }
            fxi[im] = fyi[im] = fzi[im] = zero;
        }
        for (int jp = e.cjPackedBegin; jp < e.cjPackedEnd; jp++)
        {
            const nbnxn_cj_packed_t* g = &cjPacked[jp];
            if (g->imei[0].imask == 0U) { continue; }
            for (int jm = 0; jm < NBNXM_GPU_JGROUP_SIZE; jm++)
            {
                const unsigned imaskJ = (g->imei[0].imask >> (jm * NCL)) & 0xFFU;
                if (imaskJ == 0U) { continue; }
                const int cj = g->cj[jm];
                for (int jc = 0; jc < CL; jc++)
                {
                    const int   ja = cj * CL + jc;
                    const v8f   xj = bcast(xq[4 * ja + 0]), yj = bcast(xq[4 * ja + 1]), zj = bcast(xq[4 * ja + 2]);
                    const float qj = xq[4 * ja + 3];
                    const int   tj = type[ja];
                    /* exclusion words of the 8 i-atoms against this j-atom: bit jm * 8 + im of each */
                    v8i w;
                    memcpy(&w, &excl[g->imei[jc / 4].excl_ind].pair[(jc & 3) * CL], sizeof(w));
                    float fjx = 0.0F, fjy = 0.0F, fjz = 0.0F;
                    for (int im = 0; im < NCL; im++)
                    {
                        if (!((imaskJ >> im) & 1U)) { continue; }
                        const v8f dx = xi[im] - xj, dy = yi[im] - yj, dz = zi[im] - zj;
                        v8f       r2 = dx * dx + dy * dy + dz * dz;
                        v8i       active = (r2 < rc2);
                        /* the cluster's pair with itself on the central image: j > i only */
                        if (ish == NBNXM_CENTRAL_SHIFT_INDEX && e.sci * NCL + im == cj) { active &= (laneId < jc); }
                        const v8i bitSet  = ((w >> (jm * NCL + im)) & 1) != 0;
                        const v8f int_bit = vsel(bitSet, one, zero);
                        r2                = vsel(r2 > minr2, r2, minr2);
                        const v8f inv_r2  = one / r2;
                        v8f       inv_r;
                        for (int l = 0; l < 8; l++) { inv_r[l] = __builtin_sqrtf(inv_r2[l]); }
                        const v8f inv_r6 = inv_r2 * inv_r2 * inv_r2 * int_bit;
                        v8f       F      = inv_r6 * (c12t[im][tj] * inv_r6 - c6t[im][tj]) * inv_r2;
                        const v8f qq     = qi[im] * bcast(qj);
                        const v8f coul   = int_bit * inv_r2 * inv_r;
                        if (ewald) { F += qq * (coul + pme_corr_f(b2 * r2) * b3); }
                        else { F += qq * (coul - twoK); }
                        F            = vsel(active, F, zero);
                        const v8f tx = F * dx, ty = F * dy, tz = F * dz;
                        fxi[im] += tx;
                        fyi[im] += ty;
                        fzi[im] += tz;
                        fjx += hsum(tx);
                        fjy += hsum(ty);
                        fjz += hsum(tz);
                    }
                    f[3 * ja + 0] -= fjx;
                    f[3 * ja + 1] -= fjy;
                    f[3 * ja + 2] -= fjz;
                }
            }
        }
        for (int im = 0; im < NCL; im++)
        {
            for (int ic = 0; ic < CL; ic++)
            {
                const int ia = (e.sci * NCL + im) * CL + ic;
                f[3 * ia + 0] += fxi[im][ic];
                f[3 * ia + 1] += fyi[im][ic];
                f[3 * ia + 2] += fzi[im][ic];
            }
        }
    }
}

int oracle_nbnxm_simd_f32(int nthreads, int natoms, int nsci, const nbnxn_sci_t* sci, const nbnxn_cj_packed_t* cjPacked,
                          const nbnxn_excl_t* excl, const float* xq, const int* type, int ntype, const float* nbfp,
                          const nbnxm_ref_params_t* p, const float* shiftvec, float* f)
{
    __builtin_cpu_init();
    if (!__builtin_cpu_supports("avx2") || !__builtin_cpu_supports("fma")) { return -1; }
    if (ntype > MAX_TYPES || p->vdwType != NBNXM_VDW_CUT) { return -1; }
    if (p->elecType != NBNXM_ELEC_EWALD_ANA && p->elecType != NBNXM_ELEC_RF && p->elecType != NBNXM_ELEC_CUT) { return -1; }
    if (nthreads < 1) { nthreads = 1; }
    float* fT = (float*)calloc((size_t)nthreads * 3 * natoms, sizeof(float));
#pragma omp parallel num_threads(nthreads)
    {
        const int t  = omp_get_thread_num();
        const int nt = omp_get_num_threads();
        /* blocks of equal j-list length, as oracle_nbnxm_ref_mt */
        long long total = 0;
        for (int s = 0; s < nsci; s++) { total += sci[s].cjPackedEnd - sci[s].cjPackedBegin; }
        int       s0 = nsci, s1 = nsci;
        long long acc = 0;
        for (int s = 0, found0 = 0; s <= nsci; s++)
        {
            if (!found0 && acc >= total * t / nt) { s0 = s; found0 = 1; }
            if (acc >= total * (t + 1) / nt) { s1 = s; break; }
            if (s < nsci) { acc += sci[s].cjPackedEnd - sci[s].cjPackedBegin; }
        }
        if (t == nt - 1) { s1 = nsci; }
        if (s0 < s1) { simd_range(s1 - s0, sci + s0, cjPacked, excl, xq, type, ntype, nbfp, p, shiftvec, fT + (size_t)t * 3 * natoms); }
#pragma omp barrier
#pragma omp for schedule(static)
        for (int i = 0; i < 3 * natoms; i++)
        {
            float sum = 0;
            for (int k = 0; k < nt; k++) { sum += fT[(size_t)k * 3 * natoms + i]; }
            f[i] += sum;
        }
    }
    free(fT);
    return 0;
}
