/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See nbnxm_ref.h for scope, citations and pinning.
 * Compiled twice (f64 / f32) like fep_oracle.c.
 */
#include "nbnxm_ref.h"

#include <omp.h>
#include <stdlib.h>

#include <math.h>
#include <stddef.h>
#include <string.h>

#ifndef ORACLE_REAL
#    define ORACLE_REAL double
#    define ORACLE_SUFFIX f64
#endif
typedef ORACLE_REAL real;

#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
#define FN(name) CAT(name, CAT(_, ORACLE_SUFFIX))

#define CL 8  /* atoms per cluster */
#define NCL 8 /* clusters per super-cluster */

/* nbnxm/pairlist.h:158-167 */
#ifdef ORACLE_IS_F32
static const real c_minDistSq = (real)3.82e-07;
#else
static const real c_minDistSq = (real)1.0e-36;
#endif

static double ref_pme_force_correction(double z2)
{
    if (z2 < 0.02)
    {
        double sum = 0.0, zp = 1.0, kfact = 1.0;
        for (int k = 1; k <= 10; k++)
        {
            kfact *= k;
            const double term = zp * 2.0 * k / ((2.0 * k + 1.0) * kfact);
            sum += (k & 1) ? -term : term;
            zp *= z2;
        }
        return 1.1283791670955126 * sum;
    }
    const double z = sqrt(z2);
    return (1.1283791670955126 * z * exp(-z2) - erf(z)) / (z2 * z);
}

/* Test scale, not part of the kernel: per shift vector and component, the sum of |f_i| over the i-atoms whose force was booked to
 * that shift force since the last reset — the magnitude of the terms a shift force is summed from (what a relative tolerance on
 * that sum has to be measured against).  Accumulated once per i-entry; safe from the OpenMP variant. */
static double FN(g_fshiftAbs)[3 * NBNXM_NUM_SHIFT_VECTORS];
void FN(oracle_nbnxm_fshift_abs)(double* out, int reset)
{
    for (int i = 0; i < 3 * NBNXM_NUM_SHIFT_VECTORS; i++)
    {
        if (out) { out[i] = FN(g_fshiftAbs)[i]; }
        if (reset) { FN(g_fshiftAbs)[i] = 0; }
    }
}

static int is_ewald(int elecType)
{
    return elecType == NBNXM_ELEC_EWALD_TAB || elecType == NBNXM_ELEC_EWALD_TAB_TWIN
           || elecType == NBNXM_ELEC_EWALD_ANA || elecType == NBNXM_ELEC_EWALD_ANA_TWIN;
}

void FN(oracle_nbnxm_ref)(int nsci, const nbnxn_sci_t* sci, const nbnxn_cj_packed_t* cjPacked,
                          const nbnxn_excl_t* excl, const real* xq, const int* type, int ntype,
                          const real* nbfp, const real* lj_comb, const real* nbfp_comb,
                          const nbnxm_ref_params_t* p, const real* shiftvec, int computeEnergy,
                          int computeFshift, real* f, real* fshift, double* Vc, double* Vvdw,
                          long long* npairsWithinCutoff)
{
    const int  ewald   = is_ewald(p->elecType);
    const int  twin    = (p->elecType == NBNXM_ELEC_EWALD_TAB_TWIN || p->elecType == NBNXM_ELEC_EWALD_ANA_TWIN);
    const int  tabulated = (p->elecType == NBNXM_ELEC_EWALD_TAB || p->elecType == NBNXM_ELEC_EWALD_TAB_TWIN)
                          && p->coulomb_tab != NULL;
    const int  ljEwald = (p->vdwType == NBNXM_VDW_EWALD_GEOM || p->vdwType == NBNXM_VDW_EWALD_LB);
    /* EXCLUSION_FORCES, nbnxm_cuda_kernel.cuh:69-78 */
    const int  exclForces = ewald || p->elecType == NBNXM_ELEC_RF || ljEwald
                           || (p->elecType == NBNXM_ELEC_CUT && computeEnergy);
    const real rc2     = (real)(p->rcoulomb * p->rcoulomb);
    const real rvdw2   = (real)(p->rvdw * p->rvdw);
    const real beta    = (real)p->ewaldcoeff_q;
    const real lje2    = (real)(p->ewaldcoeff_lj * p->ewaldcoeff_lj);
    const real lje6_6  = lje2 * lje2 * lje2 / (real)6;
    long long  npair   = 0;

    for (int s = 0; s < nsci; s++)
    {
        const nbnxn_sci_t e   = sci[s];
        const int         ish = e.shift & NBNXM_CI_SHIFT_MASK;
        const real        shX = shiftvec[3 * ish + 0], shY = shiftvec[3 * ish + 1], shZ = shiftvec[3 * ish + 2];
        double            vctot = 0, vvtot = 0;
        real              fsh[3] = { 0, 0, 0 };
        double            fshAbs[3] = { 0, 0, 0 };

        if (computeEnergy && exclForces && ish == NBNXM_CENTRAL_SHIFT_INDEX && e.cjPackedEnd > e.cjPackedBegin
            && cjPacked[e.cjPackedBegin].cj[0] == e.sci * NCL)
        {
            /* self terms, nbnxm_cuda_kernel.cuh:365-400 */
            double q2 = 0, c6self = 0;
            for (int a = e.sci * NCL * CL; a < (e.sci + 1) * NCL * CL; a++)
            {
                q2 += (double)xq[4 * a + 3] * (double)xq[4 * a + 3];
                if (ljEwald) { c6self += nbfp[2 * (type[a] * (ntype + 1))]; }
            }
            if (ewald) { vctot += -p->epsfac * q2 * p->ewaldcoeff_q * 0.564189583547756; }
            else { vctot += -p->epsfac * q2 * 0.5 * p->c_rf; }
            if (ljEwald) { vvtot += c6self * 0.5 * (1.0 / 6.0) * (double)lje6_6; }
        }

        for (int jp = e.cjPackedBegin; jp < e.cjPackedEnd; jp++)
        {
            const nbnxn_cj_packed_t* g = &cjPacked[jp];
            for (int jm = 0; jm < NBNXM_GPU_JGROUP_SIZE; jm++)
            {
                const int cj = g->cj[jm];
                for (int im = 0; im < NCL; im++)
                {
                    if (!((g->imei[0].imask >> (jm * NCL + im)) & 1U)) { continue; }
                    const int ci = e.sci * NCL + im;
                    for (int ic = 0; ic < CL; ic++)
                    {
                        const int  ia = ci * CL + ic;
                        const real ix = shX + xq[4 * ia + 0], iy = shY + xq[4 * ia + 1], iz = shZ + xq[4 * ia + 2];
                        const real iq = (real)p->epsfac * xq[4 * ia + 3];
                        real       fix = 0, fiy = 0, fiz = 0;
                        for (int jc = 0; jc < CL; jc++)
                        {
                            const int ja = cj * CL + jc;
                            const int half = jc / 4;
                            const unsigned int w = excl[g->imei[half].excl_ind].pair[(jc & 3) * CL + ic];
                            const real int_bit = (real)((w >> (jm * NCL + im)) & 1U);
                            if (exclForces)
                            {
                                /* diagonal: only j > i (nbnxm_cuda_kernel.cuh:406,491) */
                                if (ish == NBNXM_CENTRAL_SHIFT_INDEX && ci == cj && jc <= ic) { continue; }
                            }
                            else if (int_bit == 0) { continue; }

                            const real dx = ix - xq[4 * ja + 0], dy = iy - xq[4 * ja + 1], dz = iz - xq[4 * ja + 2];
                            real       r2 = dx * dx + dy * dy + dz * dz;
                            if (!(r2 < rc2)) { continue; }
                            if (int_bit != 0) { npair++; }
                            r2 = r2 > c_minDistSq ? r2 : c_minDistSq;
                            const real inv_r  = (real)1 / (real)sqrt((double)r2);
                            const real inv_r2 = inv_r * inv_r;
                            const real r      = r2 * inv_r;
                            const real mask   = exclForces ? int_bit : (real)1;

                            /* LJ parameters */
                            real c6, c12;
                            if (p->vdwType == NBNXM_VDW_CUT_COMB_GEOM)
                            {
                                c6  = lj_comb[2 * ia] * lj_comb[2 * ja];
                                c12 = lj_comb[2 * ia + 1] * lj_comb[2 * ja + 1];
                            }
                            else if (p->vdwType == NBNXM_VDW_CUT_COMB_LB)
                            {
                                const real sigma = lj_comb[2 * ia] + lj_comb[2 * ja];
                                const real eps   = lj_comb[2 * ia + 1] * lj_comb[2 * ja + 1];
                                const real s2    = sigma * sigma;
                                const real s6    = s2 * s2 * s2;
                                c6               = eps * s6;
                                c12              = c6 * s6;
                            }
                            else
                            {
                                const int t = ntype * type[ia] + type[ja];
                                c6          = nbfp[2 * t];
                                c12         = nbfp[2 * t + 1];
                            }
                            const real inv_r6 = inv_r2 * inv_r2 * inv_r2 * mask;
                            real       F_invr = inv_r6 * (c12 * inv_r6 - c6) * inv_r2;
                            real       E_lj   = int_bit
                                          * (c12 * (inv_r6 * inv_r6 + (real)p->rep_cpot) / (real)12
                                             - c6 * (inv_r6 + (real)p->disp_cpot) / (real)6);

                            if (p->vdwType == NBNXM_VDW_FSWITCH)
                            {
                                real rsw = r - (real)p->rvdw_switch;
                                rsw      = rsw >= 0 ? rsw : 0;
                                F_invr += -c6 * ((real)p->disp_c2 + (real)p->disp_c3 * rsw) * rsw * rsw * inv_r
                                          + c12 * ((real)p->rep_c2 + (real)p->rep_c3 * rsw) * rsw * rsw * inv_r;
                                E_lj += c6 * ((real)p->disp_c2 / 3 + (real)p->disp_c3 / 4 * rsw) * rsw * rsw * rsw
                                        - c12 * ((real)p->rep_c2 / 3 + (real)p->rep_c3 / 4 * rsw) * rsw * rsw * rsw;
                            }
                            if (ljEwald)
                            {
                                real c6grid;
                                if (p->vdwType == NBNXM_VDW_EWALD_GEOM)
                                {
                                    c6grid = nbfp_comb[2 * type[ia]] * nbfp_comb[2 * type[ja]];
                                }
                                else
                                {
                                    const real sigma = nbfp_comb[2 * type[ia]] + nbfp_comb[2 * type[ja]];
                                    const real eps   = nbfp_comb[2 * type[ia] + 1] * nbfp_comb[2 * type[ja] + 1];
                                    const real s2    = sigma * sigma;
                                    c6grid           = eps * s2 * s2 * s2;
                                }
                                const real inv_r6_nm = inv_r2 * inv_r2 * inv_r2;
                                const real cr2       = lje2 * r2;
                                const real expmcr2   = (real)exp(-(double)cr2);
                                const real poly      = 1 + cr2 + (real)0.5 * cr2 * cr2;
                                F_invr += c6grid * (inv_r6_nm - expmcr2 * (inv_r6_nm * poly + lje6_6)) * inv_r2;
                                E_lj += c6grid / (real)6
                                        * (inv_r6_nm * (1 - expmcr2 * poly) + (real)p->sh_lj_ewald * int_bit);
                            }
                            if (p->vdwType == NBNXM_VDW_PSWITCH)
                            {
                                real rsw = r - (real)p->rvdw_switch;
                                rsw      = rsw >= 0 ? rsw : 0;
                                const real sw = 1 + ((real)p->sw_c3 + ((real)p->sw_c4 + (real)p->sw_c5 * rsw) * rsw) * rsw * rsw * rsw;
                                const real dsw = (3 * (real)p->sw_c3 + (4 * (real)p->sw_c4 + 5 * (real)p->sw_c5 * rsw) * rsw) * rsw * rsw;
                                F_invr = F_invr * sw - inv_r * E_lj * dsw;
                                E_lj *= sw;
                            }
                            if (twin && !(r2 < rvdw2))
                            {
                                F_invr = 0;
                                E_lj   = 0;
                            }

                            /* electrostatics */
                            const real qq = iq * xq[4 * ja + 3];
                            real       E_el;
                            if (ewald)
                            {
                                const real b2 = beta * beta;
                                if (tabulated)
                                {
                                    /* kernel_gpu_ref.cpp:265-271: fexcl = (1 - frac) tab[ri] + frac tab[ri + 1] */
                                    const real rs   = r * (real)p->coulomb_tab_scale;
                                    int        ri   = (int)rs;
                                    if (ri > p->coulomb_tab_size - 2) { ri = p->coulomb_tab_size - 2; }
                                    const real frac = rs - (real)ri;
                                    const real fexcl = ((real)1 - frac) * (real)p->coulomb_tab[ri] + frac * (real)p->coulomb_tab[ri + 1];
                                    F_invr += qq * (mask * inv_r2 * inv_r - fexcl * inv_r);
                                }
                                else
                                {
                                    F_invr += qq * (mask * inv_r2 * inv_r + (real)ref_pme_force_correction((double)(b2 * r2)) * b2 * beta);
                                }
                                E_el = qq * (inv_r * (int_bit - (real)erf((double)(r * beta))) - int_bit * (real)p->sh_ewald);
                            }
                            else if (p->elecType == NBNXM_ELEC_RF)
                            {
                                F_invr += qq * (mask * inv_r2 * inv_r - 2 * (real)p->k_rf);
                                E_el = qq * (int_bit * inv_r + (real)p->k_rf * r2 - (real)p->c_rf);
                            }
                            else
                            {
                                F_invr += qq * mask * inv_r2 * inv_r;
                                E_el = qq * (int_bit * inv_r - (real)p->c_rf);
                            }
                            if (computeEnergy)
                            {
                                vctot += E_el;
                                vvtot += E_lj;
                            }
                            const real tx = F_invr * dx, ty = F_invr * dy, tz = F_invr * dz;
                            fix += tx;
                            fiy += ty;
                            fiz += tz;
                            f[3 * ja + 0] -= tx;
                            f[3 * ja + 1] -= ty;
                            f[3 * ja + 2] -= tz;
                        }
                        f[3 * ia + 0] += fix;
                        f[3 * ia + 1] += fiy;
                        f[3 * ia + 2] += fiz;
                        fsh[0] += fix;
                        fsh[1] += fiy;
                        fsh[2] += fiz;
                        fshAbs[0] += fabs((double)fix);
                        fshAbs[1] += fabs((double)fiy);
                        fshAbs[2] += fabs((double)fiz);
                    }
                }
            }
        }
        if (computeFshift && ish != NBNXM_CENTRAL_SHIFT_INDEX)
        {
            fshift[3 * ish + 0] += fsh[0];
            fshift[3 * ish + 1] += fsh[1];
            fshift[3 * ish + 2] += fsh[2];
            for (int d = 0; d < 3; d++)
            {
#pragma omp atomic
                FN(g_fshiftAbs)[3 * ish + d] += fshAbs[d];
            }
        }
        if (computeEnergy)
        {
            *Vc += vctot;
            *Vvdw += vvtot;
        }
    }
    if (npairsWithinCutoff) { *npairsWithinCutoff = npair; }
}

/* The same kernel on nthreads OpenMP threads: contiguous blocks of i-entries per thread, one private force /
 * shift-force buffer per thread, summed afterwards (how the reference's CPU path threads its non-bonded
 * kernels, nbnxm/atomdata.cpp reduce step).  Used for the CPU baseline of bench.py. */
void FN(oracle_nbnxm_ref_mt)(int nthreads, int natoms, int nsci, const nbnxn_sci_t* sci, const nbnxn_cj_packed_t* cjPacked,
                             const nbnxn_excl_t* excl, const real* xq, const int* type, int ntype,
                             const real* nbfp, const real* lj_comb, const real* nbfp_comb,
                             const nbnxm_ref_params_t* p, const real* shiftvec, int computeEnergy,
                             int computeFshift, real* f, real* fshift, double* Vc, double* Vvdw,
                             long long* npairsWithinCutoff)
{
    if (nthreads < 1) { nthreads = 1; }
    real*      fT   = (real*)calloc((size_t)nthreads * 3 * natoms, sizeof(real));
    real*      fsT  = (real*)calloc((size_t)nthreads * 3 * NBNXM_NUM_SHIFT_VECTORS, sizeof(real));
    double*    vcT  = (double*)calloc((size_t)nthreads * 2, sizeof(double));
    long long* npT  = (long long*)calloc((size_t)nthreads, sizeof(long long));
#pragma omp parallel num_threads(nthreads)
    {
        const int t  = omp_get_thread_num();
        const int nt = omp_get_num_threads();
        /* blocks of equal j-list length rather than equal i-entry count */
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
        if (s0 < s1)
        {
            FN(oracle_nbnxm_ref)(s1 - s0, sci + s0, cjPacked, excl, xq, type, ntype, nbfp, lj_comb, nbfp_comb, p, shiftvec,
                                 computeEnergy, computeFshift, fT + (size_t)t * 3 * natoms,
                                 fsT + (size_t)t * 3 * NBNXM_NUM_SHIFT_VECTORS, &vcT[2 * t], &vcT[2 * t + 1], &npT[t]);
        }
#pragma omp barrier
#pragma omp for schedule(static)
        for (int i = 0; i < 3 * natoms; i++)
        {
            real sum = 0;
            for (int k = 0; k < nt; k++) { sum += fT[(size_t)k * 3 * natoms + i]; }
            f[i] += sum;
        }
    }
    long long np = 0;
    for (int t = 0; t < nthreads; t++)
    {
        for (int i = 0; i < 3 * NBNXM_NUM_SHIFT_VECTORS; i++) { fshift[i] += fsT[(size_t)t * 3 * NBNXM_NUM_SHIFT_VECTORS + i]; }
        *Vc += vcT[2 * t];
        *Vvdw += vcT[2 * t + 1];
        np += npT[t];
    }
    if (npairsWithinCutoff) { *npairsWithinCutoff = np; }
    free(fT);
    free(fsT);
    free(vcT);
    free(npT);
}

#ifdef ORACLE_IS_PRIMARY
long long oracle_nbnxm_prune(int nsci, const nbnxn_sci_t* sci, nbnxn_cj_packed_t* cjPacked,
                             const float* xq, const float* shiftvec, double rlist)
{
    const float rl2  = (float)(rlist * rlist);
    long long   nset = 0;
    for (int s = 0; s < nsci; s++)
    {
        const nbnxn_sci_t e   = sci[s];
        const int         ish = e.shift & NBNXM_CI_SHIFT_MASK;
        for (int jp = e.cjPackedBegin; jp < e.cjPackedEnd; jp++)
        {
            unsigned int imask = cjPacked[jp].imei[0].imask;
            for (int jm = 0; jm < NBNXM_GPU_JGROUP_SIZE; jm++)
            {
                const int cj = cjPacked[jp].cj[jm];
                for (int im = 0; im < NCL; im++)
                {
                    const unsigned int bit = 1U << (jm * NCL + im);
                    if (!(imask & bit)) { continue; }
                    int within = 0;
                    for (int ic = 0; ic < CL && !within; ic++)
                    {
                        const int   ia = (e.sci * NCL + im) * CL + ic;
                        const float ix = xq[4 * ia] + shiftvec[3 * ish], iy = xq[4 * ia + 1] + shiftvec[3 * ish + 1],
                                    iz = xq[4 * ia + 2] + shiftvec[3 * ish + 2];
                        for (int jc = 0; jc < CL; jc++)
                        {
                            const int   ja = cj * CL + jc;
                            const float dx = ix - xq[4 * ja], dy = iy - xq[4 * ja + 1], dz = iz - xq[4 * ja + 2];
                            if (dx * dx + dy * dy + dz * dz < rl2)
                            {
                                within = 1;
                                break;
                            }
                        }
                    }
                    if (!within) { imask &= ~bit; }
                }
            }
            cjPacked[jp].imei[0].imask = imask;
            cjPacked[jp].imei[1].imask = imask;
            nset += __builtin_popcount(imask);
        }
    }
    return nset;
}
#endif
