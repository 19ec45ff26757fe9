/* See listed_ref.h.  Plain C, double precision, one interaction at a time. */
#define _GNU_SOURCE
#include "listed_ref.h"

#include <math.h>
#include <stddef.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define DEG2RAD (M_PI / 180.0)
#define CENTRAL 22

typedef struct
{
    double v[3];
} vec3;

static vec3 vsub(vec3 a, vec3 b)
{
    vec3 r = { { a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2] } };
    return r;
}
static double vdot(vec3 a, vec3 b)
{
    return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2];
}
static vec3 vcross(vec3 a, vec3 b)
{
    vec3 r = { { a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0] } };
    return r;
}
static vec3 vscale(double s, vec3 a)
{
    vec3 r = { { s * a.v[0], s * a.v[1], s * a.v[2] } };
    return r;
}
static vec3 getx(const double* x, int a)
{
    vec3 r = { { x[3 * a], x[3 * a + 1], x[3 * a + 2] } };
    return r;
}
static void addf(double* f, int a, vec3 v, double sign)
{
    for (int d = 0; d < 3; d++) { f[3 * a + d] += sign * v.v[d]; }
}

/* minimum-image difference in a rectangular box; returns the shift index (pbc_dx_aiuc semantics) */
static int pbc_dx(const double* box, int npbcdim, vec3 a, vec3 b, vec3* dx)
{
    int is[3] = { 0, 0, 0 };
    *dx       = vsub(a, b);
    for (int d = 0; d < npbcdim; d++)
    {
        const double s = rint(dx->v[d] / box[d]);
        dx->v[d] -= s * box[d];
        is[d] = (int)(-s);
    }
    /* xyzToShiftIndex: (z + 1) * 15 + (y + 1) * 5 + (x + 2)  (pbcutil/ishift.h) */
    return (is[2] + 1) * 15 + (is[1] + 1) * 5 + (is[0] + 2);
}

static void add_fshift(double* fshift, int idx, vec3 v, double sign)
{
    if (fshift != NULL) { addf(fshift, idx, v, sign); }
}

/* V = 1/2 k (x - x0)^2 with k, x0 interpolated; returns dV/dlambda (bonded.cpp:188-214) */
static double harmonic(double kA, double kB, double xA, double xB, double x, double lambda, double* V, double* F)
{
    const double L1 = 1.0 - lambda;
    const double kk = L1 * kA + lambda * kB;
    const double x0 = L1 * xA + lambda * xB;
    const double dx = x - x0;
    *F              = -kk * dx;
    *V              = 0.5 * kk * dx * dx;
    return 0.5 * (kB - kA) * dx * dx + (xA - xB) * kk * dx;
}

static double bond_angle(const double* x, const double* box, int npbcdim, int ai, int aj, int ak, vec3* r_ij, vec3* r_kj,
                         double* costh, int* t1, int* t2)
{
    *t1 = pbc_dx(box, npbcdim, getx(x, ai), getx(x, aj), r_ij);
    *t2 = pbc_dx(box, npbcdim, getx(x, ak), getx(x, aj), r_kj);
    double c = vdot(*r_ij, *r_kj) / sqrt(vdot(*r_ij, *r_ij) * vdot(*r_kj, *r_kj));
    if (c > 1.0) { c = 1.0; }
    if (c < -1.0) { c = -1.0; }
    *costh = c;
    return acos(c);
}

static void angle_forces(double dVdt, double costh, vec3 r_ij, vec3 r_kj, int ai, int aj, int ak, int t1, int t2, double* f,
                         double* fshift)
{
    const double c2 = costh * costh;
    if (c2 < 1.0)
    {
        const double st    = dVdt / sqrt(1.0 - c2);
        const double sth   = st * costh;
        const double nrij2 = vdot(r_ij, r_ij), nrkj2 = vdot(r_kj, r_kj);
        const double cik = st / sqrt(nrij2 * nrkj2), cii = sth / nrij2, ckk = sth / nrkj2;
        vec3         f_i, f_k, f_j;
        for (int d = 0; d < 3; d++)
        {
            f_i.v[d] = cii * r_ij.v[d] - cik * r_kj.v[d];
            f_k.v[d] = ckk * r_kj.v[d] - cik * r_ij.v[d];
            f_j.v[d] = -f_i.v[d] - f_k.v[d];
        }
        addf(f, ai, f_i, 1);
        addf(f, aj, f_j, 1);
        addf(f, ak, f_k, 1);
        add_fshift(fshift, t1, f_i, 1);
        add_fshift(fshift, CENTRAL, f_j, 1);
        add_fshift(fshift, t2, f_k, 1);
    }
}

static double dih_angle(const double* x, const double* box, int npbcdim, int ai, int aj, int ak, int al, vec3* r_ij, vec3* r_kj,
                        vec3* r_kl, vec3* m, vec3* n, int* t1, int* t2, int* t3)
{
    *t1 = pbc_dx(box, npbcdim, getx(x, ai), getx(x, aj), r_ij);
    *t2 = pbc_dx(box, npbcdim, getx(x, ak), getx(x, aj), r_kj);
    *t3 = pbc_dx(box, npbcdim, getx(x, ak), getx(x, al), r_kl);
    *m  = vcross(*r_ij, *r_kj);
    *n  = vcross(*r_kj, *r_kl);
    const vec3   w   = vcross(*m, *n);
    const double phi = atan2(sqrt(vdot(w, w)), vdot(*m, *n)); /* gmx_angle */
    return (vdot(*r_ij, *n) < 0.0) ? -phi : phi;
}

static void do_dih_fup(int ai, int aj, int ak, int al, double ddphi, vec3 r_ij, vec3 r_kj, vec3 r_kl, vec3 m, vec3 n, double* f,
                       double* fshift, const double* x, const double* box, int npbcdim, int t1, int t2)
{
    const double iprm = vdot(m, m), iprn = vdot(n, n), nrkj2 = vdot(r_kj, r_kj);
    const double toler = nrkj2 * 2.2e-16;
    if (iprm > toler && iprn > toler)
    {
        const double nrkj = sqrt(nrkj2);
        const vec3   f_i  = vscale(-ddphi * nrkj / iprm, m);
        const vec3   f_l  = vscale(ddphi * nrkj / iprn, n);
        const double p    = vdot(r_ij, r_kj) / nrkj2;
        const double q    = vdot(r_kl, r_kj) / nrkj2;
        vec3         svec, f_j, f_k;
        for (int d = 0; d < 3; d++)
        {
            svec.v[d] = p * f_i.v[d] - q * f_l.v[d];
            f_j.v[d]  = f_i.v[d] - svec.v[d];
            f_k.v[d]  = f_l.v[d] + svec.v[d];
        }
        addf(f, ai, f_i, 1);
        addf(f, aj, f_j, -1);
        addf(f, ak, f_k, -1);
        addf(f, al, f_l, 1);
        if (fshift != NULL)
        {
            vec3      dx_jl;
            const int t3 = pbc_dx(box, npbcdim, getx(x, al), getx(x, aj), &dx_jl);
            add_fshift(fshift, t1, f_i, 1);
            add_fshift(fshift, CENTRAL, f_j, -1);
            add_fshift(fshift, t2, f_k, -1);
            add_fshift(fshift, t3, f_l, 1);
        }
    }
}

void oracle_listed(int ftype, int n, const int* iatoms, const listed_iparams_t* params, const double* x, const double* box,
                   int npbcdim, double lambda, double* f, double* fshift, double* epot, double* dvdl)
{
    const double L1   = 1.0 - lambda;
    const int    nral = (ftype == LISTED_BONDS || ftype == LISTED_RESTRBONDS) ? 2 : ((ftype == LISTED_ANGLES || ftype == LISTED_UREY_BRADLEY) ? 3 : 4);
    for (int i = 0; i < n; i++)
    {
        const int*              ia = iatoms + (size_t)i * (1 + nral);
        const listed_iparams_t* ip = &params[ia[0]];
        const double*           p  = ip->p;
        if (ftype == LISTED_BONDS)
        {
            vec3         dx;
            const int    ki  = pbc_dx(box, npbcdim, getx(x, ia[1]), getx(x, ia[2]), &dx);
            const double dr2 = vdot(dx, dx), dr = sqrt(dr2);
            double       vb, fb;
            *dvdl += harmonic(p[1], p[3], p[0], p[2], dr, lambda, &vb, &fb);
            *epot += vb;
            if (dr2 != 0.0)
            {
                const vec3 fij = vscale(fb / dr, dx);
                addf(f, ia[1], fij, 1);
                addf(f, ia[2], fij, -1);
                add_fshift(fshift, ki, fij, 1);
                add_fshift(fshift, CENTRAL, fij, -1);
            }
        }
        else if (ftype == LISTED_RESTRBONDS)
        {
            vec3         dx;
            const int    ki  = pbc_dx(box, npbcdim, getx(x, ia[1]), getx(x, ia[2]), &dx);
            const double dr2 = vdot(dx, dx), dr = sqrt(dr2);
            const double low = L1 * p[0] + lambda * p[4], dlow = p[4] - p[0];
            const double up1 = L1 * p[1] + lambda * p[5], dup1 = p[5] - p[1];
            const double up2 = L1 * p[2] + lambda * p[6], dup2 = p[6] - p[2];
            const double k   = L1 * p[3] + lambda * p[7], dk = p[7] - p[3];
            double       vb = 0, fb = 0, dv = 0;
            if (dr < low)
            {
                const double drh = dr - low;
                vb = 0.5 * k * drh * drh;
                fb = -k * drh;
                dv = 0.5 * dk * drh * drh - k * dlow * drh;
            }
            else if (dr <= up1) {}
            else if (dr <= up2)
            {
                const double drh = dr - up1;
                vb = 0.5 * k * drh * drh;
                fb = -k * drh;
                dv = 0.5 * dk * drh * drh - k * dup1 * drh;
            }
            else
            {
                const double drh = dr - up2;
                vb = k * (up2 - up1) * (0.5 * (up2 - up1) + drh);
                fb = -k * (up2 - up1);
                dv = dk * (up2 - up1) * (0.5 * (up2 - up1) + drh) + k * (dup2 - dup1) * (up2 - up1 + drh) - k * (up2 - up1) * dup2;
            }
            *dvdl += dv;
            if (dr2 != 0.0)
            {
                *epot += vb;
                const vec3 fij = vscale(fb / dr, dx);
                addf(f, ia[1], fij, 1);
                addf(f, ia[2], fij, -1);
                add_fshift(fshift, ki, fij, 1);
                add_fshift(fshift, CENTRAL, fij, -1);
            }
        }
        else if (ftype == LISTED_ANGRES)
        {
            vec3         r_ij, r_kl;
            const int    t1 = pbc_dx(box, npbcdim, getx(x, ia[2]), getx(x, ia[1]), &r_ij);
            const int    t2 = pbc_dx(box, npbcdim, getx(x, ia[4]), getx(x, ia[3]), &r_kl);
            const double nij2 = vdot(r_ij, r_ij), nkl2 = vdot(r_kl, r_kl);
            double       cos_phi = vdot(r_ij, r_kl) / sqrt(nij2 * nkl2);
            if (cos_phi > 1.0) { cos_phi = 1.0; }
            if (cos_phi < -1.0) { cos_phi = -1.0; }
            const double phi = acos(cos_phi);
            /* dopdihs_min: V = cp (1 - cos(mult (phi - phi0))) */
            const double phi0  = (L1 * p[0] + lambda * p[2]) * DEG2RAD;
            const double dph0  = (p[2] - p[0]) * DEG2RAD;
            const double cp    = L1 * p[1] + lambda * p[3];
            const double mdphi = ip->mult * (phi - phi0);
            const double v1    = 1.0 - cos(mdphi);
            const double dVdphi = cp * ip->mult * sin(mdphi);
            *dvdl += (p[3] - p[1]) * v1 + cp * dph0 * sin(mdphi);
            *epot += cp * v1;
            const double cos_phi2 = cos_phi * cos_phi;
            if (cos_phi2 < 1.0)
            {
                const double st  = -dVdphi / sqrt(1.0 - cos_phi2);
                const double sth = st * cos_phi;
                const double c   = st / sqrt(nij2 * nkl2), cij = sth / nij2, ckl = sth / nkl2;
                const vec3   f_i = vsub(vscale(c, r_kl), vscale(cij, r_ij));
                const vec3   f_k = vsub(vscale(c, r_ij), vscale(ckl, r_kl));
                addf(f, ia[1], f_i, 1);
                addf(f, ia[2], f_i, -1);
                addf(f, ia[3], f_k, 1);
                addf(f, ia[4], f_k, -1);
                add_fshift(fshift, t1, f_i, 1);
                add_fshift(fshift, CENTRAL, f_i, -1);
                add_fshift(fshift, t2, f_k, 1);
                add_fshift(fshift, CENTRAL, f_k, -1);
            }
        }
        else if (ftype == LISTED_ANGLES || ftype == LISTED_UREY_BRADLEY)
        {
            vec3         r_ij, r_kj;
            double       costh;
            int          t1, t2;
            const double theta = bond_angle(x, box, npbcdim, ia[1], ia[2], ia[3], &r_ij, &r_kj, &costh, &t1, &t2);
            double       va, dVdt;
            if (ftype == LISTED_ANGLES) { *dvdl += harmonic(p[1], p[3], p[0] * DEG2RAD, p[2] * DEG2RAD, theta, lambda, &va, &dVdt); }
            else
            {
                *dvdl += harmonic(p[1], p[5], p[0] * DEG2RAD, p[4] * DEG2RAD, theta, lambda, &va, &dVdt);
                vec3         r_ik;
                const int    ki  = pbc_dx(box, npbcdim, getx(x, ia[1]), getx(x, ia[3]), &r_ik);
                const double dr2 = vdot(r_ik, r_ik), dr = sqrt(dr2);
                double       vb, fb;
                *dvdl += harmonic(p[3], p[7], p[2], p[6], dr, lambda, &vb, &fb);
                *epot += vb;
                if (dr2 != 0.0)
                {
                    const vec3 fik = vscale(fb / dr, r_ik);
                    addf(f, ia[1], fik, 1);
                    addf(f, ia[3], fik, -1);
                    add_fshift(fshift, ki, fik, 1);
                    add_fshift(fshift, CENTRAL, fik, -1);
                }
            }
            *epot += va;
            angle_forces(dVdt, costh, r_ij, r_kj, ia[1], ia[2], ia[3], t1, t2, f, fshift);
        }
        else
        {
            vec3         r_ij, r_kj, r_kl, m, nn;
            int          t1, t2, t3;
            double       phi = dih_angle(x, box, npbcdim, ia[1], ia[2], ia[3], ia[4], &r_ij, &r_kj, &r_kl, &m, &nn, &t1, &t2, &t3);
            double       ddphi;
            if (ftype == LISTED_PDIHS)
            {
                const double phi0  = (L1 * p[0] + lambda * p[2]) * DEG2RAD;
                const double dph0  = (p[2] - p[0]) * DEG2RAD;
                const double cp    = L1 * p[1] + lambda * p[3];
                const double mdphi = ip->mult * phi - phi0;
                const double v1    = 1.0 + cos(mdphi);
                ddphi              = -cp * ip->mult * sin(mdphi);
                *dvdl += (p[3] - p[1]) * v1 + cp * dph0 * sin(mdphi);
                *epot += cp * v1;
            }
            else if (ftype == LISTED_DIHRES)
            {
                const double phi0A = p[0] * DEG2RAD, dphiA = p[1] * DEG2RAD, kfacA = p[2];
                const double phi0B = p[3] * DEG2RAD, dphiB = p[4] * DEG2RAD, kfacB = p[5];
                const double phi0 = L1 * phi0A + lambda * phi0B, dphi = L1 * dphiA + lambda * dphiB, kfac = L1 * kfacA + lambda * kfacB;
                double       dp   = phi - phi0;
                if (dp >= M_PI) { dp -= 2 * M_PI; }
                else if (dp < -M_PI) { dp += 2 * M_PI; }
                const double ddp = (dp > dphi) ? dp - dphi : ((dp < -dphi) ? dp + dphi : 0.0);
                *epot += 0.5 * kfac * ddp * ddp;
                *dvdl += 0.5 * (kfacB - kfacA) * ddp * ddp;
                if (ddp > 0) { *dvdl -= kfac * ddp * ((dphiB - dphiA) + (phi0B - phi0A)); }
                else if (ddp < 0) { *dvdl += kfac * ddp * ((dphiB - dphiA) - (phi0B - phi0A)); }
                ddphi = kfac * ddp;
            }
            else if (ftype == LISTED_IDIHS)
            {
                const double kk   = L1 * p[1] + lambda * p[3];
                const double phi0 = (L1 * p[0] + lambda * p[2]) * DEG2RAD;
                const double dph0 = (p[2] - p[0]) * DEG2RAD;
                double       dp   = phi - phi0;
                if (dp >= M_PI) { dp -= 2 * M_PI; }
                else if (dp < -M_PI) { dp += 2 * M_PI; }
                *dvdl += 0.5 * (p[3] - p[1]) * dp * dp - kk * dph0 * dp;
                *epot += 0.5 * kk * dp * dp;
                ddphi = kk * dp; /* do_dih_fup gets -(-kk dp) */
            }
            else
            {
                /* Ryckaert-Bellemans: polymer convention, psi = phi - pi */
                if (phi >= M_PI) { phi -= M_PI; }
                else { phi += M_PI; }
                const double cosphi = cos(phi), sinphi = sin(phi);
                double       v = 0, dd = 0, cosfac = 1.0;
                for (int j = 0; j < 6; j++)
                {
                    const double rbp = L1 * p[j] + lambda * p[6 + j];
                    if (j > 0)
                    {
                        dd += j * rbp * cosfac;
                        cosfac *= cosphi;
                    }
                    v += cosfac * rbp;
                    *dvdl += cosfac * (p[6 + j] - p[j]);
                }
                ddphi = -dd * sinphi;
                *epot += v;
            }
            do_dih_fup(ia[1], ia[2], ia[3], ia[4], ddphi, r_ij, r_kj, r_kl, m, nn, f, fshift, x, box, npbcdim, t1, t2);
            (void)t3;
        }
    }
}

void oracle_listed_simple_pairs(int kind, int n, const int* iatoms, const listed_iparams_t* params, const double* x, const double* box,
                                int npbcdim, double epsfac, double* f, double* fshift, double* eLJ, double* eCoul)
{
    for (int i = 0; i < n; i++)
    {
        const int*    ia = iatoms + (size_t)i * 3;
        const double* p  = params[ia[0]].p;
        const double  qq = (kind == 1) ? p[0] * p[1] * p[2] : p[0] * p[1];
        const double  c6 = (kind == 1) ? p[3] : p[2], c12 = (kind == 1) ? p[4] : p[3];
        vec3          dr;
        const int     ki    = pbc_dx(box, npbcdim, getx(x, ia[1]), getx(x, ia[2]), &dr);
        const double  r2    = vdot(dr, dr);
        const double  rinv2 = 1.0 / r2, rinv = sqrt(rinv2), rinv6 = rinv2 * rinv2 * rinv2;
        const double  velec = epsfac * qq * rinv;
        const double  fr    = (12.0 * c12 * rinv6 - 6.0 * c6) * rinv6 + velec;
        const vec3    fij   = vscale(fr * rinv2, dr);
        *eLJ += (c12 * rinv6 - c6) * rinv6;
        *eCoul += velec;
        addf(f, ia[1], fij, 1);
        addf(f, ia[2], fij, -1);
        if (ki != CENTRAL)
        {
            add_fshift(fshift, ki, fij, 1);
            add_fshift(fshift, CENTRAL, fij, -1);
        }
    }
}

void oracle_listed_pairs(int n, const int* iatoms, const listed_iparams_t* params, const double* x, const double* qA, const double* qB,
                         const double* box, int npbcdim, const listed_pairs_fep_t* fep, double elecScale, double* f, double* fshift,
                         double* eLJ, double* eCoul, double* dvdlVdw, double* dvdlCoul)
{
    for (int i = 0; i < n; i++)
    {
        const int     ai = iatoms[3 * i + 1], aj = iatoms[3 * i + 2];
        const double* p  = params[iatoms[3 * i]].p;
        const double  qq[2]  = { qA[ai] * qA[aj], qB[ai] * qB[aj] };
        const double  c6[2]  = { p[0], p[2] };
        const double  c12[2] = { p[1], p[3] };
        vec3          dr;
        const int     ki = pbc_dx(box, npbcdim, getx(x, ai), getx(x, aj), &dr);
        const double  r2 = vdot(dr, dr), rinv2 = 1.0 / r2, rinv6 = rinv2 * rinv2 * rinv2;
        double        finvr = 0;
        if (qq[0] == qq[1] && c6[0] == c6[1] && c12[0] == c12[1])
        {
            const double velec = elecScale * qq[0] * sqrt(rinv2);
            *eCoul += velec;
            *eLJ += (c12[0] * rinv6 - c6[0]) * rinv6;
            finvr = ((12.0 * c12[0] * rinv6 - 6.0 * c6[0]) * rinv6 + velec) * rinv2;
        }
        else
        {
            const double rpm2 = r2 * r2, rp = rpm2 * r2;
            const int    hard = (c12[0] > 0 && c12[1] > 0);
            const double alphaV = hard ? 0.0 : fep->alphaVdw, alphaC = hard ? 0.0 : fep->alphaCoul;
            for (int k = 0; k < 2; k++)
            {
                const double LFC = (k == 0) ? 1.0 - fep->lambdaCoul : fep->lambdaCoul;
                const double LFV = (k == 0) ? 1.0 - fep->lambdaVdw : fep->lambdaVdw;
                const double DLF = (k == 0) ? -1.0 : 1.0;
                const int    pw  = fep->lambdaPower;
                const double scC = (pw == 2) ? (1 - LFC) * (1 - LFC) : (1 - LFC), scV = (pw == 2) ? (1 - LFV) * (1 - LFV) : (1 - LFV);
                const double dscC = DLF * pw / 6.0 * ((pw == 2) ? (1 - LFC) : 1.0), dscV = DLF * pw / 6.0 * ((pw == 2) ? (1 - LFV) : 1.0);
                double       sigma6;
                if (c6[k] > 0 && c12[k] > 0)
                {
                    sigma6 = c12[k] / c6[k];
                    if (sigma6 < fep->sc_sigma6_min) { sigma6 = fep->sc_sigma6_min; }
                }
                else { sigma6 = fep->sc_sigma6; }
                double FC = 0, FV = 0, VC = 0, VV = 0;
                if (qq[k] != 0 || c6[k] != 0 || c12[k] != 0)
                {
                    const double rpinvC = 1.0 / (alphaC * scC * sigma6 + rp);
                    const double rpinvV = 1.0 / (alphaV * scV * sigma6 + rp);
                    const double rinvC  = pow(rpinvC, 1.0 / 6.0);
                    const double V6 = c6[k] * rpinvV, V12 = c12[k] * rpinvV * rpinvV;
                    VV = V12 - V6;
                    FV = (12.0 * V12 - 6.0 * V6) * rpinvV;
                    VC = elecScale * qq[k] * rinvC;
                    FC = VC * rpinvC;
                }
                *eCoul += LFC * VC;
                *eLJ += LFV * VV;
                *dvdlCoul += VC * DLF + LFC * alphaC * dscC * FC * sigma6;
                *dvdlVdw += VV * DLF + LFV * alphaV * dscV * FV * sigma6;
                finvr += (LFC * FC + LFV * FV) * rpm2;
            }
        }
        const vec3 fv = vscale(finvr, dr);
        addf(f, ai, fv, 1);
        addf(f, aj, fv, -1);
        add_fshift(fshift, ki, fv, 1);
        add_fshift(fshift, CENTRAL, fv, -1);
    }
}
