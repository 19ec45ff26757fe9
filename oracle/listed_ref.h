/*
 * CPU oracle of the perturbed listed (bonded) interactions: TEST INFRASTRUCTURE ONLY (see fep_oracle.h).
 * Restates, in double precision, the semantics of the reference's CPU kernels
 *   listed_forces/bonded.cpp:  harmonic :188-214, bonds :216-276, angles :941-1031, urey_bradley :1172-1278,
 *                              dih_angle :1414-1434, do_dih_fup :1437-1500, dopdihs :1564-1590, pdihs :1648-1710,
 *                              idihs :1968-2035, rbdihs :2560-2680
 * which its GPU path (listed_forces_gpu_internal.cu:781-1363, *_fep_gpu) follows for the perturbed types.
 * Pinned by the reference's own known answers (tests/golden/listed_refdata.json, test_oracle_golden.py).
 */
#ifndef LISTED_REF_H
#define LISTED_REF_H

#ifdef __cplusplus
extern "C"
{
#endif

enum
{
    LISTED_BONDS = 0,    /* F_BONDS          2 atoms  p: rA krA rB krB */
    LISTED_ANGLES,       /* F_ANGLES         3 atoms  p: thA kA thB kB (degrees) */
    LISTED_UREY_BRADLEY, /* F_UREY_BRADLEY   3 atoms  p: thetaA kthetaA r13A kUBA thetaB kthetaB r13B kUBB */
    LISTED_PDIHS,        /* F_PDIHS/F_PIDIHS 4 atoms  p: phiA cpA phiB cpB, mult */
    LISTED_RBDIHS,       /* F_RBDIHS         4 atoms  p: rbcA[6] rbcB[6] */
    LISTED_IDIHS,        /* F_IDIHS          4 atoms  p: xA kA xB kB (degrees) */
    /* restraints (lambda = the restraint lambda): bonded.cpp restraint_bonds :619-712, low_angres :2337-2420 (F_ANGRES),
     * dihres :2472-2560; the fork's GPU twins listed_forces_gpu_internal.cu:1605-1872 */
    LISTED_RESTRBONDS,   /* F_RESTRBONDS     2 atoms  p: lowA up1A up2A kA lowB up1B up2B kB */
    LISTED_ANGRES,       /* F_ANGRES         4 atoms  p: phiA cpA phiB cpB, mult (angle between i->j and k->l) */
    LISTED_DIHRES,       /* F_DIHRES         4 atoms  p: phiA dphiA kfacA phiB dphiB kfacB (degrees) */
    LISTED_NUM_TYPES
};

/* double here (the reference's t_iparams are `real`, and its known answers hold to 1e-8 in double); the GPU path's
 * listed_iparams_t (include/listed_hip.h) has the same layout in float */
typedef struct
{
    double p[12];
    int    mult;
    int    pad;
} listed_iparams_t;

/* x: 3 doubles per atom; box: 3 diagonal lengths (rectangular), npbcdim: 0 none, 2 xy, 3 xyz;
 * iatoms: (1 + nral) ints per interaction: parameter index, atoms; f: 3 doubles per atom (+=);
 * fshift: 45 x 3 (+=, may be NULL). */
void oracle_listed(int ftype, int numInteractions, const int* iatoms, const listed_iparams_t* params, const double* x,
                   const double* box, int npbcdim, double lambda, double* f, double* fshift, double* epot, double* dvdl);

/* Perturbed 1-4 pairs (F_LJ14): Beutler soft-core LJ + plain Coulomb between the A and B states, no cut-off.
 * Semantics: free_energy_evaluate_single (listed_forces/pairs.cpp:130-330, soft-core "beutler") with its table look-ups
 * replaced by the functions they tabulate, which is also what pairs_fep_gpu does.  params[type].p = c6A c12A c6B c12B. */
typedef struct
{
    double alphaCoul, alphaVdw;
    int    lambdaPower, pad;
    double sc_sigma6, sc_sigma6_min;
    double lambdaCoul, lambdaVdw;
} listed_pairs_fep_t;

/* Unperturbed pairs with their own parameters (pairs_gpu pType 1, 2; listed_forces/pairs.cpp do_pairs_simple / general):
 * kind 1 = F_LJC14_Q   p: qi qj fqq c6 c12   qq = qi qj fqq
 * kind 2 = F_LJC_PAIRS_NB p: qi qj c6 c12     qq = qi qj;   Coulomb scaled by epsfac in both */
void oracle_listed_simple_pairs(int kind, int numPairs, const int* iatoms, const listed_iparams_t* params, const double* x,
                                const double* box, int npbcdim, double epsfac, double* f, double* fshift, double* eLJ, double* eCoul);

void oracle_listed_pairs(int numPairs, const int* iatoms, const listed_iparams_t* params, const double* x, const double* qA,
                         const double* qB, const double* box, int npbcdim, const listed_pairs_fep_t* fep, double elecScale,
                         double* f, double* fshift, double* eLJ, double* eCoul, double* dvdlVdw, double* dvdlCoul);

#ifdef __cplusplus
}
#endif
#endif
