/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See fep_oracle.h for scope, citations and pinning.
 *
 * One translation unit, compiled twice:
 *   gcc -DORACLE_REAL=double -DORACLE_SUFFIX=f64 ...   and   -DORACLE_REAL=float -DORACLE_SUFFIX=f32
 *
 * Structure (own, not the reference's): one function evaluates ONE (i,j) list pair and returns
 * its contributions; the list walker accumulates them.  The arithmetic of each branch follows the
 * scalar instantiation of nb_free_energy_kernel<> (nb_free_energy.cpp:466-1170).
 */
#include "fep_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifndef ORACLE_REAL
#    define ORACLE_REAL double
#    define ORACLE_SUFFIX f64
#endif
typedef ORACLE_REAL real;

#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
#define FN(name) CAT(name, CAT(_, ORACLE_SUFFIX))

/* nb_free_energy.cpp:99,107 */
static const real c_minDistanceSquared = (real)1.0e-12;
static const real c_maxRInvSix         = (real)1.0e15;

static inline real r_sqrt(real v) { return (real)sqrt((double)v); }
static inline real r_cbrt(real v) { return (real)cbrt((double)v); }
static inline real r_exp(real v) { return (real)exp((double)v); }
static inline real r_abs(real v) { return v < 0 ? -v : v; }
static inline real r_min(real a, real b) { return a < b ? a : b; }
static inline real r_max(real a, real b) { return a > b ? a : b; }

/* erf(z)/z as a function of z^2  (what gmx::pmePotentialCorrection approximates,
 * simd_math.h:1655-1686).  Evaluated with libm erf in double: the reference's rational
 * approximation is accurate to 1e-6 (float) / 4e-11 (double) of this. */
static double pme_potential_correction(double z2)
{
    if (z2 < 1e-8)
    {
        /* 2/sqrt(pi) (1 - z^2/3 + z^4/10) */
        return 1.1283791670955126 * (1.0 - z2 / 3.0 + z2 * z2 / 10.0);
    }
    const double z = sqrt(z2);
    return erf(z) / z;
}

/* [d/dz (erf(z)/z)] / z as a function of z^2 (gmx::pmeForceCorrection, simd_math.h:1560-1650). */
static double pme_force_correction(double z2)
{
    if (z2 < 0.02)
    {
        /* 2/sqrt(pi) * sum_{k>=1} (-1)^k z^(2k-2) 2k / ((2k+1) k!) */
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

#ifdef ORACLE_IS_PRIMARY
/* Precision-independent helpers, emitted once (by the f64 build of this file). */
void oracle_softcore_from_fepvals(oracle_fep_params_t* p, double sc_alpha, int sc_power,
                                  double sc_sigma, double sc_sigma_min, int bScCoul,
                                  int softcoreType, double gapsysLinpointLJ, double gapsysLinpointQ,
                                  double gapsysSigmaLJ)
{
    /* interaction_const.cpp:50-63 */
    p->alphaVdw               = sc_alpha;
    p->alphaCoulomb           = bScCoul ? sc_alpha : 0.0;
    p->lambdaPower            = sc_power;
    p->sigma6WithInvalidSigma = pow(sc_sigma, 6);
    p->sigma6Minimum          = bScCoul ? pow(sc_sigma_min, 6) : 0.0;
    p->softcoreType           = softcoreType;
    p->gapsysScaleLinpointVdW = gapsysLinpointLJ;
    p->gapsysScaleLinpointCoul = gapsysLinpointQ;
    p->gapsysSigma6VdW        = pow(gapsysSigmaLJ, 6);
}

/* ewald_utils.cpp:43-70: bisection on erfc(beta rc) = rtol */
double oracle_calc_ewaldcoeff_q(double rc, double rtol)
{
    double beta = 5, low, high;
    int    n, i = 0;
    do
    {
        i++;
        beta *= 2;
    } while (erfc(beta * rc) > rtol);
    n    = i + 60;
    low  = 0;
    high = beta;
    for (i = 0; i < n; i++)
    {
        beta = (low + high) / 2;
        if (erfc(beta * rc) > rtol) { low = beta; }
        else { high = beta; }
    }
    return beta;
}

static double lj_ewald_tail(double beta, double rc)
{
    const double x2 = beta * rc * beta * rc;
    return exp(-x2) * (1 + x2 + x2 * x2 / 2.0);
}

/* ewald_utils.cpp:72-112 */
double oracle_calc_ewaldcoeff_lj(double rc, double rtol)
{
    double beta = 5, low, high;
    int    n, i = 0;
    do
    {
        i++;
        beta *= 2.0;
    } while (lj_ewald_tail(beta, rc) > rtol);
    n    = i + 60;
    low  = 0;
    high = beta;
    for (i = 0; i < n; ++i)
    {
        beta = (low + high) / 2.0;
        if (lj_ewald_tail(beta, rc) > rtol) { low = beta; }
        else { high = beta; }
    }
    return beta;
}
#endif

/* ---- Gapsys soft-core pieces (nb_softcore.h) ------------------------------------------ */

/* nb_softcore.h:44-69 */
static void gapsys_quadratic_coulomb(real qq, real rInvQ, real r, real lambdaFac, real dLambdaFac,
                                     real* force, real* potential, real* dvdl)
{
    const real constFac = qq * rInvQ;
    const real linFac   = constFac * r * rInvQ;
    const real quadrFac = linFac * r * rInvQ;
    *force              = -2 * quadrFac + 3 * linFac;
    *potential          = quadrFac - 3 * (linFac - constFac);
    const real lambdaFacRevInv = (real)1 / ((real)1 - lambdaFac);
    *dvdl = dLambdaFac * (real)0.5 * (lambdaFac * lambdaFacRevInv) * (quadrFac - 2 * linFac + constFac);
}

/* nb_softcore.h:71-195, isEwald selects the "ewald modification" instead of the "rf modification" */
static void gapsys_coulomb(int isEwald, real qq, real facel, real r, real rCutoff, real lambdaFac,
                           real dLambdaFac, real alphaEff, real krf, real potentialShift,
                           real* force, real* potential, real* dvdl)
{
    if (!(lambdaFac < 1 && 0 < alphaEff && facel != 0)) { return; }
    real rQ = r_cbrt(1 - lambdaFac);
    rQ      = r_sqrt(rQ) * (1 + r_abs(qq / facel));
    rQ      = rQ * alphaEff;
    const int withinCutoff = (rQ <= rCutoff);
    if (rCutoff < rQ) { rQ = rCutoff; }
    if (!(r < rQ)) { return; }
    const real rInvQ = (real)1 / rQ;
    real       fq, vq, dq;
    gapsys_quadratic_coulomb(qq, rInvQ, r, lambdaFac, dLambdaFac, &fq, &vq, &dq);
    if (isEwald) { vq = vq - qq * potentialShift; }
    else
    {
        fq = fq - qq * 2 * krf * r * r;
        vq = vq + qq * (krf * r * r - potentialShift);
    }
    *force     = fq;
    *potential = vq;
    if (withinCutoff) { *dvdl += dq; }
}

/* nb_softcore.h:197-279 */
static void gapsys_lj(real c6, real c12, real r, real rsq, real lambdaFac, real dLambdaFac,
                      real sigma6, real alphaEff, real repulsionShift, real dispersionShift,
                      real* force, real* potential, real* dvdl)
{
    if (!(lambdaFac < 1 && 0 < alphaEff)) { return; }
    const real lambdaFacRev    = 1 - lambdaFac;
    const real lambdaFacRevInv = (real)1 / lambdaFacRev;
    real       rQ              = r_cbrt((real)(26.0 / 7.0) * sigma6 * lambdaFacRev);
    rQ                         = r_sqrt(rQ) * alphaEff;
    if (!(r < rQ)) { return; }
    const real c6s   = c6 / (real)6;
    const real c12s  = c12 / (real)12;
    const real rInvQ = (real)1 / rQ;
    real       rInv6C = rInvQ * rInvQ * rInvQ;
    rInv6C            = rInv6C * rInv6C;
    real       rInv7C = rInv6C * rInvQ;
    real       rInv8C = rInv7C * rInvQ;
    const real rInv14C = c12s * rInv7C * rInv7C * rsq;
    const real rInv13C = c12s * rInv7C * rInv6C * r;
    const real rInv12C = c12s * rInv6C * rInv6C;
    rInv8C             = rInv8C * c6s * rsq;
    rInv7C             = rInv7C * c6s * r;
    rInv6C             = rInv6C * c6s;
    const real quadrFac  = 156 * rInv14C - 42 * rInv8C;
    const real linearFac = 168 * rInv13C - 48 * rInv7C;
    const real constFac  = 91 * rInv12C - 28 * rInv6C;
    *force     = -quadrFac + linearFac;
    *potential = (real)0.5 * quadrFac - linearFac + constFac + (c12s * repulsionShift - c6s * dispersionShift);
    *dvdl += dLambdaFac * 28 * (lambdaFac * lambdaFacRevInv)
             * (((real)6.5 * rInv14C - rInv8C) - (13 * rInv13C - 2 * rInv7C)
                + ((real)6.5 * rInv12C - rInv6C));
}

/* ---- LJ-PME grid correction (nb_free_energy.cpp:121-163) ------------------------------- */
static void lj_pme_correction(real rInv, real rSq, real ewaldLJCoeffSq, real ewaldLJCoeffSixDivSix,
                              int iEqJ, real* pot, real* force)
{
#if defined(ORACLE_IS_F32)
    const real eps = (real)1.19209290e-07;
#else
    const real eps = (real)2.2204460492503131e-16;
#endif
    const real switchPoint = (real)pow(8.0 * (double)eps, 1.0 / 6.0);
    const real rInvSq      = rInv * rInv;
    const real rInvSix     = rInvSq * rInvSq * rInvSq;
    const real x           = ewaldLJCoeffSq * rSq;
    const real expNegX     = r_exp(-x);
    const real poly        = 1 + x + (real)0.5 * x * x;
    const real fullTerm    = rInvSix * (1 - expNegX * poly);
    const real approx      = ewaldLJCoeffSixDivSix * (1 + x * ((real)-0.75 + (real)0.3 * x));
    const real term        = (x < switchPoint) ? approx : fullTerm;
    *force                 = (term - expNegX * ewaldLJCoeffSixDivSix) * rInvSq;
    *pot                   = iEqJ ? (real)0.5 * ewaldLJCoeffSixDivSix : term;
}

/* ---- per-list constants ------------------------------------------------------------------ */
typedef struct
{
    int  softcore; /* 0 none, 1 Beutler, 2 Gapsys  (dispatchKernel, nb_free_energy.cpp:1315-1364) */
    int  scLambdasOrAlphasDiffer;
    real LFC[2], LFV[2], DLF[2];
    real scLFC[2], scDLFC[2], scLFV[2], scDLFV[2];
    real swV3, swV4, swV5, swF2, swF3, swF4;
    real ewaldLJCoeffSq, ewaldLJCoeffSixDivSix;
    real rCutoffMaxSq;
    real sh_ewald;
} consts_t;

static void setup_consts(const oracle_fep_params_t* p, double lambdaCoul, double lambdaVdw, consts_t* c)
{
    memset(c, 0, sizeof(*c));
    if (p->softcoreType == ORACLE_SOFTCORE_BEUTLER)
    {
        c->softcore = (p->alphaCoulomb == 0 && p->alphaVdw == 0) ? 0 : 1;
    }
    else
    {
        c->softcore = (p->gapsysScaleLinpointCoul == 0 && p->gapsysScaleLinpointVdW == 0) ? 0 : 2;
    }
    /* nb_free_energy.cpp:1405-1419 */
    c->scLambdasOrAlphasDiffer = 1;
    if (p->alphaCoulomb == 0 && p->alphaVdw == 0) { c->scLambdasOrAlphasDiffer = 0; }
    else if ((real)lambdaCoul == (real)lambdaVdw && p->alphaCoulomb == p->alphaVdw)
    {
        c->scLambdasOrAlphasDiffer = 0;
    }
    /* :420-449 */
    c->LFC[0] = (real)1 - (real)lambdaCoul;
    c->LFV[0] = (real)1 - (real)lambdaVdw;
    c->LFC[1] = (real)lambdaCoul;
    c->LFV[1] = (real)lambdaVdw;
    c->DLF[0] = -1;
    c->DLF[1] = 1;
    const real lp = (real)p->lambdaPower;
    for (int k = 0; k < 2; k++)
    {
        const real oc = 1 - c->LFC[k], ov = 1 - c->LFV[k];
        c->scLFC[k]   = (p->lambdaPower == 2) ? oc * oc : oc;
        c->scDLFC[k]  = c->DLF[k] * lp / (real)6 * ((p->lambdaPower == 2) ? oc : (real)1);
        c->scLFV[k]   = (p->lambdaPower == 2) ? ov * ov : ov;
        c->scDLFV[k]  = c->DLF[k] * lp / (real)6 * ((p->lambdaPower == 2) ? ov : (real)1);
    }
    if (p->vdwPotSwitch)
    {
        /* :361-370 */
        const real d = (real)p->rvdw - (real)p->rvdw_switch;
        c->swV3      = (real)-10.0 / (d * d * d);
        c->swV4      = (real)15.0 / (d * d * d * d);
        c->swV5      = (real)-6.0 / (d * d * d * d * d);
        c->swF2      = (real)-30.0 / (d * d * d);
        c->swF3      = (real)60.0 / (d * d * d * d);
        c->swF4      = (real)-30.0 / (d * d * d * d * d);
    }
    if (p->vdwIsEwald)
    {
        c->ewaldLJCoeffSq        = (real)p->ewaldcoeff_lj * (real)p->ewaldcoeff_lj;
        c->ewaldLJCoeffSixDivSix = c->ewaldLJCoeffSq * c->ewaldLJCoeffSq * c->ewaldLJCoeffSq / (real)6;
    }
    real rmax       = (real)(p->rcoulomb > p->rvdw ? p->rcoulomb : p->rvdw);
    c->rCutoffMaxSq = rmax * rmax;
    c->sh_ewald     = (p->elecIsEwald || p->vdwIsEwald) ? (real)p->sh_ewald : (real)0;
}

typedef struct
{
    real fscal;      /* scalar force / r  (force on i = fscal * (xi - xj)) */
    real vCoul, vVdw;
    real dvdlCoul, dvdlVdw;
    int  skipped;    /* pair was beyond the cut-off and not an exclusion */
} pair_out_t;

/* One list pair.  d = x_i(+shift) - x_j. */
static void eval_pair(const oracle_fep_params_t* p, const consts_t* c, int computeForces,
                      real dX, real dY, real dZ, int pairIncluded, int iEqJ,
                      const real qq[2], const real c6[2], const real c12[2], const real c6grid[2],
                      pair_out_t* o)
{
    memset(o, 0, sizeof(*o));
    real rSq = dX * dX + dY * dY + dZ * dZ;
    const int withinCutoff = (rSq < c->rCutoffMaxSq);
    const int pairExcluded = !pairIncluded;
    if (!withinCutoff && !pairExcluded)
    {
        o->skipped = 1;
        return; /* :667-678 */
    }

    /* per-pair soft-core inputs (:553-628) */
    real sigma6[2] = { 0, 0 }, gapsysSigma6[2] = { 0, 0 };
    real alphaVdwEff = 0, alphaCoulEff = 0, gapsysLinV = 0, gapsysLinC = 0;
    for (int k = 0; k < 2; k++)
    {
        if (c6[k] > 0 && c12[k] > 0)
        {
            sigma6[k] = (real)0.5 * c12[k] / c6[k];
            gapsysSigma6[k] = sigma6[k];
            if (sigma6[k] < (real)p->sigma6Minimum) { sigma6[k] = (real)p->sigma6Minimum; }
        }
        else
        {
            sigma6[k]       = (real)p->sigma6WithInvalidSigma;
            gapsysSigma6[k] = (real)p->gapsysSigma6VdW;
        }
    }
    if (!(c12[0] > 0 && c12[1] > 0))
    {
        alphaVdwEff  = (real)p->alphaVdw;
        alphaCoulEff = (real)p->alphaCoulomb;
        gapsysLinV   = (real)p->gapsysScaleLinpointVdW;
        gapsysLinC   = (real)p->gapsysScaleLinpointCoul;
    }

    rSq             = r_max(rSq, c_minDistanceSquared); /* :723 */
    const real rInv = (real)1 / r_sqrt(rSq);
    const real r    = rSq * rInv;

    real rp, rpm2;
    if (c->softcore == 1)
    {
        rpm2 = rSq * rSq;
        rp   = rpm2 * rSq;
    }
    else
    {
        rpm2 = rInv * rInv;
        rp   = 1;
    }

    real fscal = 0;

    if (withinCutoff && pairIncluded)
    {
        real fC[2] = { 0, 0 }, fV[2] = { 0, 0 }, vC[2] = { 0, 0 }, vV[2] = { 0, 0 };
        for (int k = 0; k < 2; k++)
        {
            if (!(qq[k] != 0 || c6[k] != 0 || c12[k] != 0)) { continue; }
            real rInvC, rInvV, rC, rV, rPInvC, rPInvV;
            if (c->softcore == 1)
            {
                /* :767-788, sixthRoot :166-172 */
                rPInvC = (real)1 / (alphaCoulEff * c->scLFC[k] * sigma6[k] + rp);
                {
                    /* sixthRoot(rPInvC, &rInvC, &rC): rC = invsqrt(cbrt(rPInvC)), rInvC = 1/rC */
                    const real invSixth = (real)1 / r_sqrt(r_cbrt(rPInvC));
                    rC                  = invSixth;
                    rInvC               = (real)1 / invSixth;
                }
                if (c->scLambdasOrAlphasDiffer)
                {
                    rPInvV              = (real)1 / (alphaVdwEff * c->scLFV[k] * sigma6[k] + rp);
                    const real invSixth = (real)1 / r_sqrt(r_cbrt(rPInvV));
                    rV                  = invSixth;
                    rInvV               = (real)1 / invSixth;
                }
                else
                {
                    rPInvV = rPInvC;
                    rInvV  = rInvC;
                    rV     = rC;
                }
            }
            else
            {
                rPInvC = 1;
                rInvC  = rInv;
                rC     = r;
                rPInvV = 1;
                rInvV  = rInv;
                rV     = r;
            }

            /* Coulomb (:804-874) */
            int doElec;
            if (p->elecIsEwald) { doElec = (r < (real)p->rcoulomb && qq[k] != 0); }
            else { doElec = (rC < (real)p->rcoulomb && qq[k] != 0); }
            if (doElec)
            {
                if (p->elecIsEwald)
                {
                    vC[k] = qq[k] * (rInvC - c->sh_ewald);
                    fC[k] = qq[k] * rInvC;
                    if (c->softcore == 2)
                    {
                        gapsys_coulomb(1, qq[k], (real)p->epsfac, rC, (real)p->rcoulomb, c->LFC[k],
                                       c->DLF[k], gapsysLinC, 0, c->sh_ewald, &fC[k], &vC[k],
                                       &o->dvdlCoul);
                    }
                }
                else
                {
                    vC[k] = qq[k] * (rInvC + (real)p->k_rf * rC * rC - (real)p->c_rf);
                    fC[k] = qq[k] * (rInvC - 2 * (real)p->k_rf * rC * rC);
                    if (c->softcore == 2)
                    {
                        gapsys_coulomb(0, qq[k], (real)p->epsfac, rC, (real)p->rcoulomb, c->LFC[k],
                                       c->DLF[k], gapsysLinC, (real)p->k_rf, (real)p->c_rf, &fC[k],
                                       &vC[k], &o->dvdlCoul);
                    }
                }
            }

            /* Van der Waals (:880-971) */
            int doVdw;
            if (p->vdwIsEwald) { doVdw = (r < (real)p->rvdw && (c6[k] != 0 || c12[k] != 0)); }
            else { doVdw = (rV < (real)p->rvdw && (c6[k] != 0 || c12[k] != 0)); }
            if (doVdw)
            {
                real rInv6;
                if (c->softcore == 1) { rInv6 = rPInvV; }
                else
                {
                    rInv6 = rInvV * rInvV;
                    rInv6 = rInv6 * rInv6 * rInv6;
                }
                rInv6            = r_min(rInv6, c_maxRInvSix);
                const real v6    = c6[k] * rInv6;
                const real v12   = c12[k] * rInv6 * rInv6;
                vV[k] = (v12 + c12[k] * (real)p->repulsion_shift_cpot) / (real)12
                        - (v6 + c6[k] * (real)p->dispersion_shift_cpot) / (real)6;
                fV[k] = v12 - v6;
                if (c->softcore == 2)
                {
                    gapsys_lj(c6[k], c12[k], r, rSq, c->LFV[k], c->DLF[k], gapsysSigma6[k],
                              gapsysLinV, (real)p->repulsion_shift_cpot,
                              (real)p->dispersion_shift_cpot, &fV[k], &vV[k], &o->dvdlVdw);
                }
                if (p->vdwIsEwald)
                {
                    vV[k] += c6grid[k] * (real)p->sh_lj_ewald / (real)6; /* :937-944 */
                }
                if (p->vdwPotSwitch)
                {
                    /* :946-963 */
                    real d = rV - (real)p->rvdw_switch;
                    if (!(0 < d)) { d = 0; }
                    const real d2  = d * d;
                    const real sw  = 1 + d2 * d * (c->swV3 + d * (c->swV4 + d * c->swV5));
                    const real dsw = d2 * (c->swF2 + d * (c->swF3 + d * c->swF4));
                    if (rV < (real)p->rvdw)
                    {
                        fV[k] = fV[k] * sw - rV * vV[k] * dsw;
                        vV[k] = vV[k] * sw;
                    }
                    else
                    {
                        fV[k] = 0;
                        vV[k] = 0;
                    }
                }
            }
            fC[k] *= rPInvC; /* :980-981 */
            fV[k] *= rPInvV;
        }

        /* assemble states (:987-1020) */
        for (int k = 0; k < 2; k++)
        {
            o->vCoul += c->LFC[k] * vC[k];
            o->vVdw += c->LFV[k] * vV[k];
            fscal += c->LFC[k] * fC[k] * rpm2;
            fscal += c->LFV[k] * fV[k] * rpm2;
            if (c->softcore == 1)
            {
                o->dvdlCoul += vC[k] * c->DLF[k] + c->LFC[k] * alphaCoulEff * c->scDLFC[k] * fC[k] * sigma6[k];
                o->dvdlVdw += vV[k] * c->DLF[k] + c->LFV[k] * alphaVdwEff * c->scDLFV[k] * fV[k] * sigma6[k];
            }
            else
            {
                o->dvdlCoul += vC[k] * c->DLF[k];
                o->dvdlVdw += vV[k] * c->DLF[k];
            }
        }
    }

    /* excluded pair, plain cut-off / reaction-field (:1023-1054) */
    if (!p->elecIsEwald && pairExcluded)
    {
        const real FF = -2 * (real)p->k_rf;
        real       VV = (real)p->k_rf * rSq - (real)p->c_rf;
        if (iEqJ) { VV *= (real)0.5; }
        for (int k = 0; k < 2; k++)
        {
            o->vCoul += c->LFC[k] * qq[k] * VV;
            fscal += c->LFC[k] * qq[k] * FF;
            o->dvdlCoul += c->DLF[k] * qq[k] * VV;
        }
    }

    /* Ewald: remove the reciprocal-space part (:1056-1101) */
    if (p->elecIsEwald && (pairExcluded || r < (real)p->rcoulomb))
    {
        const real beta = (real)p->ewaldcoeff_q;
        const real brsq = rSq * beta * beta;
        real       v_lr = beta * (real)pme_potential_correction((double)brsq);
        real       f_lr = -brsq * beta * (real)pme_force_correction((double)brsq);
        f_lr            = f_lr * rInv * rInv;
        if (iEqJ) { v_lr *= (real)0.5; }
        for (int k = 0; k < 2; k++)
        {
            o->vCoul -= c->LFC[k] * qq[k] * v_lr;
            fscal -= c->LFC[k] * qq[k] * f_lr;
            o->dvdlCoul -= c->DLF[k] * qq[k] * v_lr;
        }
    }

    /* LJ-PME: remove the grid part (:1103-1136) */
    if (p->vdwIsEwald && (pairExcluded || r < (real)p->rvdw))
    {
        real v_lr, f_lr;
        lj_pme_correction(rInv, rSq, c->ewaldLJCoeffSq, c->ewaldLJCoeffSixDivSix, iEqJ, &v_lr, &f_lr);
        v_lr = v_lr / (real)6;
        for (int k = 0; k < 2; k++)
        {
            o->vVdw += c->LFV[k] * c6grid[k] * v_lr;
            fscal += c->LFV[k] * c6grid[k] * f_lr;
            o->dvdlVdw += c->DLF[k] * c6grid[k] * v_lr;
        }
    }

    o->fscal = computeForces ? fscal : 0;
}

/* Test scale, not part of the kernel: sums of |V_coul|, |V_vdw|, |dV/dl_coul|, |dV/dl_vdw| over the pairs of the last call —
 * the magnitude of the terms the energies and dV/dlambda of the perturbed pairs are summed from (what a relative
 * tolerance on those sums has to be measured against). */
static double FN(g_absSums)[4];
void FN(oracle_fep_last_abs_sums)(double* out)
{
    for (int i = 0; i < 4; i++) { out[i] = FN(g_absSums)[i]; }
}
/* the same for the shift forces of the last call: per shift vector and component the sum of |f_i| of the i-entries booked to it
 * (at most ORACLE_MAX_SHIFT shift vectors are tracked) */
#define ORACLE_MAX_SHIFT 45
static double FN(g_fshiftAbs)[3 * ORACLE_MAX_SHIFT];
void FN(oracle_fep_last_fshift_abs)(double* out)
{
    for (int i = 0; i < 3 * ORACLE_MAX_SHIFT; i++) { out[i] = FN(g_fshiftAbs)[i]; }
}

void FN(oracle_nb_free_energy_kernel)(int nri, const int* iinr, const int* jindex, const int* jjnr,
                                      const int* shift, const int* excl_fep, const real* x,
                                      int ntype, const oracle_fep_params_t* p, const real* shiftvec,
                                      const real* nbfp, const real* nbfp_grid, const real* chargeA,
                                      const real* chargeB, const int* typeA, const int* typeB,
                                      int flags, double lambdaCoul, double lambdaVdw, real* f,
                                      real* fshift, double* Vc, double* Vv, double* dvdl)
{
    consts_t c;
    setup_consts(p, lambdaCoul, lambdaVdw, &c);
    const int computeForces = (flags & ORACLE_DO_FORCE) != 0;
    const int doShift       = (flags & ORACLE_DO_SHIFTFORCE) != 0;
    const int doPotential   = (flags & ORACLE_DO_POTENTIAL) != 0;

    real dvdlCoul = 0, dvdlVdw = 0;
    double absSums[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < 3 * ORACLE_MAX_SHIFT; i++) { FN(g_fshiftAbs)[i] = 0; }

    for (int n = 0; n < nri; n++)
    {
        const int  is = shift[n];
        const int  ii = iinr[n];
        const real ix = shiftvec[3 * is + 0] + x[3 * ii + 0];
        const real iy = shiftvec[3 * is + 1] + x[3 * ii + 1];
        const real iz = shiftvec[3 * is + 2] + x[3 * ii + 2];
        const real iqA = (real)p->epsfac * chargeA[ii];
        const real iqB = (real)p->epsfac * chargeB[ii];
        const int  ntiA = ntype * typeA[ii];
        const int  ntiB = ntype * typeB[ii];
        real       vCoulTot = 0, vVdwTot = 0, fIX = 0, fIY = 0, fIZ = 0;
        int        havePairsWithinCutoff = 0;

        for (int k = jindex[n]; k < jindex[n + 1]; k++)
        {
            const int jnr      = jjnr[k];
            const int included = (excl_fep == NULL || excl_fep[k]) ? 1 : 0;
            const int tiA = ntiA + typeA[jnr], tiB = ntiB + typeB[jnr];
            real      qq[2]  = { iqA * chargeA[jnr], iqB * chargeB[jnr] };
            real      c6[2]  = { nbfp[2 * tiA], nbfp[2 * tiB] };
            real      c12[2] = { nbfp[2 * tiA + 1], nbfp[2 * tiB + 1] };
            real      c6g[2] = { 0, 0 };
            if (p->vdwIsEwald)
            {
                c6g[0] = nbfp_grid[2 * tiA];
                c6g[1] = nbfp_grid[2 * tiB];
            }
            const real dX = ix - x[3 * jnr + 0];
            const real dY = iy - x[3 * jnr + 1];
            const real dZ = iz - x[3 * jnr + 2];
            pair_out_t o;
            eval_pair(p, &c, computeForces, dX, dY, dZ, included, ii == jnr, qq, c6, c12, c6g, &o);
            if (o.skipped) { continue; }
            havePairsWithinCutoff = 1;
            vCoulTot += o.vCoul;
            vVdwTot += o.vVdw;
            dvdlCoul += o.dvdlCoul;
            dvdlVdw += o.dvdlVdw;
            absSums[0] += fabs((double)o.vCoul);
            absSums[1] += fabs((double)o.vVdw);
            absSums[2] += fabs((double)o.dvdlCoul);
            absSums[3] += fabs((double)o.dvdlVdw);
            if (computeForces && o.fscal != 0)
            {
                const real tX = o.fscal * dX, tY = o.fscal * dY, tZ = o.fscal * dZ;
                fIX += tX;
                fIY += tY;
                fIZ += tZ;
                f[3 * jnr + 0] -= tX;
                f[3 * jnr + 1] -= tY;
                f[3 * jnr + 2] -= tZ;
            }
        }
        if (havePairsWithinCutoff)
        {
            if (computeForces)
            {
                f[3 * ii + 0] += fIX;
                f[3 * ii + 1] += fIY;
                f[3 * ii + 2] += fIZ;
                if (doShift)
                {
                    fshift[3 * is + 0] += fIX;
                    fshift[3 * is + 1] += fIY;
                    fshift[3 * is + 2] += fIZ;
                    if (is < ORACLE_MAX_SHIFT)
                    {
                        FN(g_fshiftAbs)[3 * is + 0] += fabs((double)fIX);
                        FN(g_fshiftAbs)[3 * is + 1] += fabs((double)fIY);
                        FN(g_fshiftAbs)[3 * is + 2] += fabs((double)fIZ);
                    }
                }
            }
            if (doPotential)
            {
                *Vc += vCoulTot;
                *Vv += vVdwTot;
            }
        }
    }
    dvdl[0] += dvdlCoul;
    dvdl[1] += dvdlVdw;
    for (int i = 0; i < 4; i++) { FN(g_absSums)[i] = absSums[i]; }
}

/* Energies and dV/dlambda at the current lambda (index 0) and n_lambda foreign lambdas
 * (freeenergydispatch.cpp:236-307): energies-only passes of the same kernel. */
/* test scale: the abs sums (|V_coul|, |V_vdw|, |dV/dl_coul|, |dV/dl_vdw| over the pairs) of every lambda index of the last
 * oracle_fep_foreign call, 4 per index, at most ORACLE_MAX_FOREIGN indices */
#define ORACLE_MAX_FOREIGN 64
static double FN(g_foreignAbs)[4 * ORACLE_MAX_FOREIGN];
void FN(oracle_fep_foreign_abs_sums)(double* out, int numIndices)
{
    for (int i = 0; i < 4 * numIndices && i < 4 * ORACLE_MAX_FOREIGN; i++) { out[i] = FN(g_foreignAbs)[i]; }
}

void FN(oracle_fep_foreign)(int nri, const int* iinr, const int* jindex, const int* jjnr,
                            const int* shift, const int* excl_fep, const real* x, int ntype,
                            const oracle_fep_params_t* p, const real* shiftvec, const real* nbfp,
                            const real* nbfp_grid, const real* chargeA, const real* chargeB,
                            const int* typeA, const int* typeB, double lambdaCoul, double lambdaVdw,
                            int n_lambda, const double* allLambdaCoul, const double* allLambdaVdw,
                            double* eVdw, double* eCoul, double* dvdlVdw, double* dvdlCoul)
{
    for (int i = 0; i <= n_lambda; i++)
    {
        const double lc = (i == 0) ? lambdaCoul : allLambdaCoul[i - 1];
        const double lv = (i == 0) ? lambdaVdw : allLambdaVdw[i - 1];
        double       Vc = 0, Vv = 0, dvdl[2] = { 0, 0 };
        FN(oracle_nb_free_energy_kernel)(nri, iinr, jindex, jjnr, shift, excl_fep, x, ntype, p,
                                         shiftvec, nbfp, nbfp_grid, chargeA, chargeB, typeA, typeB,
                                         ORACLE_DO_POTENTIAL, lc, lv, NULL, NULL, &Vc, &Vv, dvdl);
        eVdw[i]     = Vv;
        eCoul[i]    = Vc;
        dvdlCoul[i] = dvdl[0];
        dvdlVdw[i]  = dvdl[1];
        if (i < ORACLE_MAX_FOREIGN)
        {
            for (int k = 0; k < 4; k++) { FN(g_foreignAbs)[4 * i + k] = FN(g_absSums)[k]; }
        }
    }
}
