/* TEST INFRASTRUCTURE ONLY — see update_ref.h. */
#include "update_ref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* minimum image for a rectangular or triclinic box with the first npbcdim dimensions periodic
 * (pbcutil/pbc_aiuc.h:142-183: shifts found from z to x with the inverse diagonal) */
static void pbc_dx(int pbcType, const double* box, const double* a, const double* b, double* dx)
{
    const int npbcdim = (pbcType == 3) ? 3 : ((pbcType == 2) ? 2 : 0);
    for (int d = 0; d < 3; d++) { dx[d] = a[d] - b[d]; }
    for (int d = npbcdim - 1; d >= 0; d--)
    {
        const double sh = nearbyint(dx[d] / box[4 * d]);
        for (int e = 0; e <= d; e++) { dx[e] -= sh * box[3 * d + e]; }
    }
}

void oracle_leapfrog(int n, double* x, double* xp, double* v, const double* f, const double* invmass, double dt, int numTempScaleValues,
                     const double* lambdas, const unsigned short* groups, const double* prDiag)
{
    for (int a = 0; a < n; a++)
    {
        double lambda = 1.0;
        if (numTempScaleValues == 1) { lambda = lambdas[0]; }
        else if (numTempScaleValues > 1) { lambda = lambdas[groups[a]]; }
        for (int d = 0; d < 3; d++)
        {
            const double vOld = v[3 * a + d];
            double       vNew = lambda * vOld + f[3 * a + d] * invmass[a] * dt;
            if (prDiag) { vNew -= prDiag[d] * vOld; }
            xp[3 * a + d] = x[3 * a + d];
            v[3 * a + d]  = vNew;
            x[3 * a + d] += vNew * dt;
        }
    }
}

static void cross(const double* a, const double* b, double* c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

void oracle_settle(int nsettle, const int* atoms, double mO, double mH, double dOH, double dHH, const double* x, double* xp, double* v,
                   double invdt, double* virial, int pbcType, const double* box)
{
    /* settleParameters: the triangle in its own frame, O on the +y axis, the hydrogens at (-+rc, -rb) */
    const double wohh = mO + 2.0 * mH;
    const double wh   = mH / wohh;
    const double rc   = dHH / 2.0;
    const double ra   = 2.0 * mH * sqrt(dOH * dOH - rc * rc) / wohh;
    const double rb   = sqrt(dOH * dOH - rc * rc) - ra;
    const double irc2 = 1.0 / dHH;

    for (int s = 0; s < nsettle; s++)
    {
        const int     io = atoms[3 * s], ih2 = atoms[3 * s + 1], ih3 = atoms[3 * s + 2];
        const double *xo = x + 3 * io, *xh2 = x + 3 * ih2, *xh3 = x + 3 * ih3;
        double        dist21[3], dist31[3], doh2[3], doh3[3], a1[3], b1[3], c1[3];
        pbc_dx(pbcType, box, xh2, xo, dist21);
        pbc_dx(pbcType, box, xh3, xo, dist31);
        pbc_dx(pbcType, box, xp + 3 * ih2, xp + 3 * io, doh2);
        pbc_dx(pbcType, box, xp + 3 * ih3, xp + 3 * io, doh3);
        /* positions after the update relative to the (mass-weighted) centre, with O as origin */
        for (int d = 0; d < 3; d++)
        {
            a1[d] = (-doh2[d] - doh3[d]) * wh;
            b1[d] = doh2[d] + a1[d];
            c1[d] = doh3[d] + a1[d];
        }
        /* orthonormal frame: z normal to the old triangle, x = a1 x z, y = z x x */
        double az[3], ax[3], ay[3];
        cross(dist21, dist31, az);
        cross(a1, az, ax);
        cross(az, ax, ay);
        const double nx = 1.0 / sqrt(dot(ax, ax)), ny = 1.0 / sqrt(dot(ay, ay)), nz = 1.0 / sqrt(dot(az, az));
        for (int d = 0; d < 3; d++)
        {
            ax[d] *= nx;
            ay[d] *= ny;
            az[d] *= nz;
        }
        const double b0dx = dot(ax, dist21), b0dy = dot(ay, dist21);
        const double c0dx = dot(ax, dist31), c0dy = dot(ay, dist31);
        const double a1dz = dot(az, a1);
        const double b1dx = dot(ax, b1), b1dy = dot(ay, b1), b1dz = dot(az, b1);
        const double c1dx = dot(ax, c1), c1dy = dot(ay, c1), c1dz = dot(az, c1);

        const double sinphi = a1dz / ra;
        double       tmp2   = 1.0 - sinphi * sinphi;
        if (tmp2 < 1e-12) { tmp2 = 1e-12; }
        const double tmp    = 1.0 / sqrt(tmp2);
        const double cosphi = tmp2 * tmp;
        const double sinpsi = (b1dz - c1dz) * irc2 * tmp;
        const double cospsi = sqrt(1.0 - sinpsi * sinpsi);

        const double a2dy = ra * cosphi;
        const double b2dx = -rc * cospsi;
        const double t1   = -rb * cosphi;
        const double t2   = rc * sinpsi * sinphi;
        const double b2dy = t1 - t2;
        const double c2dy = t1 + t2;

        const double alpha  = b2dx * (b0dx - c0dx) + b0dy * b2dy + c0dy * c2dy;
        const double beta   = b2dx * (c0dy - b0dy) + b0dx * b2dy + c0dx * c2dy;
        const double gamma  = b0dx * b1dy - b1dx * b0dy + c0dx * c1dy - c1dx * c0dy;
        const double al2be2 = alpha * alpha + beta * beta;
        const double sinthe = (alpha * gamma - beta * sqrt(al2be2 - gamma * gamma)) / al2be2;
        const double costhe = sqrt(1.0 - sinthe * sinthe);

        const double a3d[3] = { -a2dy * sinthe, a2dy * costhe, a1dz };
        const double b3d[3] = { b2dx * costhe - b2dy * sinthe, b2dx * sinthe + b2dy * costhe, b1dz };
        const double c3d[3] = { -b2dx * costhe - c2dy * sinthe, -b2dx * sinthe + c2dy * costhe, c1dz };

        double dxO[3], dxH2[3], dxH3[3];
        for (int d = 0; d < 3; d++)
        {
            dxO[d]  = ax[d] * a3d[0] + ay[d] * a3d[1] + az[d] * a3d[2] - a1[d];
            dxH2[d] = ax[d] * b3d[0] + ay[d] * b3d[1] + az[d] * b3d[2] - b1[d];
            dxH3[d] = ax[d] * c3d[0] + ay[d] * c3d[1] + az[d] * c3d[2] - c1[d];
        }
        for (int d = 0; d < 3; d++)
        {
            xp[3 * io + d] += dxO[d];
            xp[3 * ih2 + d] += dxH2[d];
            xp[3 * ih3 + d] += dxH3[d];
            if (v)
            {
                v[3 * io + d] += dxO[d] * invdt;
                v[3 * ih2 + d] += dxH2[d] * invdt;
                v[3 * ih3 + d] += dxH3[d] * invdt;
            }
        }
        if (virial)
        {
            double mdo[3], mdb[3], mdc[3];
            for (int d = 0; d < 3; d++)
            {
                mdb[d] = mH * dxH2[d];
                mdc[d] = mH * dxH3[d];
                mdo[d] = mO * dxO[d] + mdb[d] + mdc[d];
            }
            for (int d2 = 0; d2 < 3; d2++)
            {
                for (int d = 0; d < 3; d++) { virial[3 * d2 + d] -= xo[d2] * mdo[d] + dist21[d2] * mdb[d] + dist31[d2] * mdc[d]; }
            }
        }
    }
}

void oracle_lincs(int ncons, const int* iatoms, const double* lengths, int natoms, const double* invmass, int numIterations, int expansionOrder,
                  const double* x, double* xp, double* v, double invdt, double* virial, int pbcType, const double* box)
{
    if (ncons == 0) { return; }
    (void)natoms;
    double* r      = (double*)malloc(sizeof(double) * 3 * ncons);
    double* blc    = (double*)malloc(sizeof(double) * ncons); /* 1 / sqrt(invmass_i + invmass_j) */
    double* rhs1   = (double*)malloc(sizeof(double) * ncons);
    double* rhs2   = (double*)malloc(sizeof(double) * ncons);
    double* sol    = (double*)malloc(sizeof(double) * ncons);
    double* lambda = (double*)calloc(ncons, sizeof(double));
    /* coupling: constraints sharing an atom; coefficient -+ invmass(shared) blc_b blc_k (r_b . r_k), the sign by whether the
     * shared atom sits at the same end of both constraints */
    int* start = (int*)calloc(ncons + 1, sizeof(int));
    for (int b = 0; b < ncons; b++)
    {
        for (int k = 0; k < ncons; k++)
        {
            if (k == b) { continue; }
            for (int eb = 1; eb <= 2; eb++)
            {
                for (int ek = 1; ek <= 2; ek++) { start[b + 1] += (iatoms[3 * b + eb] == iatoms[3 * k + ek]); }
            }
        }
    }
    for (int b = 0; b < ncons; b++) { start[b + 1] += start[b]; }
    const int ncc   = start[ncons];
    int*      nbr   = (int*)malloc(sizeof(int) * (ncc + 1));
    double*   mfac  = (double*)malloc(sizeof(double) * (ncc + 1));
    double*   blcc  = (double*)malloc(sizeof(double) * (ncc + 1));
    for (int b = 0; b < ncons; b++)
    {
        blc[b] = 1.0 / sqrt(invmass[iatoms[3 * b + 1]] + invmass[iatoms[3 * b + 2]]);
    }
    for (int b = 0, n = 0; b < ncons; b++)
    {
        for (int k = 0; k < ncons; k++)
        {
            if (k == b) { continue; }
            for (int eb = 1; eb <= 2; eb++)
            {
                for (int ek = 1; ek <= 2; ek++)
                {
                    if (iatoms[3 * b + eb] == iatoms[3 * k + ek])
                    {
                        const double sign = (eb == ek) ? -1.0 : 1.0;
                        nbr[n]            = k;
                        mfac[n]           = sign * invmass[iatoms[3 * b + eb]] * blc[b] * blc[k];
                        n++;
                    }
                }
            }
        }
    }

    for (int b = 0; b < ncons; b++)
    {
        double dx[3];
        pbc_dx(pbcType, box, x + 3 * iatoms[3 * b + 1], x + 3 * iatoms[3 * b + 2], dx);
        const double rlen = 1.0 / sqrt(dot(dx, dx));
        for (int d = 0; d < 3; d++) { r[3 * b + d] = rlen * dx[d]; }
    }
    for (int b = 0; b < ncons; b++)
    {
        for (int n = start[b]; n < start[b + 1]; n++) { blcc[n] = mfac[n] * dot(r + 3 * b, r + 3 * nbr[n]); }
        double dx[3];
        pbc_dx(pbcType, box, xp + 3 * iatoms[3 * b + 1], xp + 3 * iatoms[3 * b + 2], dx);
        rhs1[b] = blc[b] * (dot(r + 3 * b, dx) - lengths[iatoms[3 * b]]);
        sol[b]  = rhs1[b];
    }
    for (int iter = -1; iter < numIterations; iter++)
    {
        if (iter >= 0)
        {
            /* correction for the lengthening by rotation */
            for (int b = 0; b < ncons; b++)
            {
                double dx[3];
                pbc_dx(pbcType, box, xp + 3 * iatoms[3 * b + 1], xp + 3 * iatoms[3 * b + 2], dx);
                const double len   = lengths[iatoms[3 * b]];
                const double dlen2 = 2.0 * len * len - dot(dx, dx);
                const double p     = (dlen2 > 0) ? len - sqrt(dlen2) : len;
                rhs1[b]            = blc[b] * p;
                sol[b]             = rhs1[b];
            }
        }
        /* (1 - A)^-1 ~ 1 + A + A^2 + ... */
        double *cur = rhs1, *next = rhs2;
        for (int rec = 0; rec < expansionOrder; rec++)
        {
            for (int b = 0; b < ncons; b++)
            {
                double mvb = 0;
                for (int n = start[b]; n < start[b + 1]; n++) { mvb += blcc[n] * cur[nbr[n]]; }
                next[b] = mvb;
                sol[b] += mvb;
            }
            double* t = cur;
            cur       = next;
            next      = t;
        }
        for (int b = 0; b < ncons; b++)
        {
            const double mvb = blc[b] * sol[b];
            lambda[b] += mvb;
            const int i = iatoms[3 * b + 1], j = iatoms[3 * b + 2];
            for (int d = 0; d < 3; d++)
            {
                xp[3 * i + d] -= invmass[i] * mvb * r[3 * b + d];
                xp[3 * j + d] += invmass[j] * mvb * r[3 * b + d];
            }
        }
    }
    for (int b = 0; b < ncons; b++)
    {
        const int i = iatoms[3 * b + 1], j = iatoms[3 * b + 2];
        if (v)
        {
            for (int d = 0; d < 3; d++)
            {
                v[3 * i + d] -= invmass[i] * lambda[b] * invdt * r[3 * b + d];
                v[3 * j + d] += invmass[j] * lambda[b] * invdt * r[3 * b + d];
            }
        }
        if (virial)
        {
            const double mult = lengths[iatoms[3 * b]] * lambda[b];
            for (int d2 = 0; d2 < 3; d2++)
            {
                for (int d = 0; d < 3; d++) { virial[3 * d2 + d] += mult * r[3 * b + d2] * r[3 * b + d]; }
            }
        }
    }
    free(r);
    free(blc);
    free(rhs1);
    free(rhs2);
    free(sol);
    free(lambda);
    free(start);
    free(nbr);
    free(mfac);
    free(blcc);
}
