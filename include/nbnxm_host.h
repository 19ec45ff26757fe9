/*
 * nbnxm_host.h — C ABI of the host-side producers that sit in front of the GPU path:
 * the synthetic benchmark system, the cluster grid and the pair-list builders.
 *
 * In mdrun these inputs come from the reference's own host code; this library exists so
 * that tests and bench.py can feed the GPU path (and the CPU oracle) with lists in exactly the
 * reference's format without linking GROMACS.  Format definitions followed:
 *   nbnxm/benchmark/bench_system.cpp:66-211   SPC/E-like water box recipe
 *   nbnxm/grid.cpp, nbnxm/gridset.cpp         columns, 64-atom super-clusters of 8x8, fillers, fepBits
 *   nbnxm/pairlist.cpp:1776-1942              make_fep_list (GPU flavour): which pairs go to t_nblist
 *   nbnxm/pairlist.cpp:2867-2961              combine_fep_lists (one list per locality)
 *   nbnxm/atomdata.cpp:930-966                nbnxn_atomdata_mask_fep (q/type of perturbed atoms zeroed)
 * The search algorithm itself is this library's own (not a translation of pairlist.cpp).
 */
#ifndef NBNXM_HOST_H
#define NBNXM_HOST_H

#include "nbnxm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic system ---------------------------------------------------------------------- */
/* Rigid 3-site waters on a jittered lattice, nmx*nmy*nmz molecules, lattice spacing `spacing` (nm).
 * The first numPerturbedMolecules molecules closest to the box centre form the "ligand":
 * A state = water, B state = decoupled (qB = 0, typeB = 2 = zero-LJ type).
 * Types: 0 = OW, 1 = HW, 2 = decoupled dummy.  Outputs are caller-allocated, N = 3*nmx*nmy*nmz:
 *   x[3N], qA[N], qB[N], typeA[N], typeB[N], molId[N]; box[3] (rectangular). */
void nbnxm_host_make_water_box(int nmx, int nmy, int nmz, double spacing, double jitter,
                               unsigned int seed, int numPerturbedMolecules, float* x, float* qA,
                               float* qB, int* typeA, int* typeB, int* molId, float* box);

/* ---- grid ---------------------------------------------------------------------------------- */
typedef struct NbnxmHostGrid NbnxmHostGrid;

/* Puts atoms (wrapped into the rectangular box) on the cluster grid.
 * ntype is the number of topology atom types; the grid-order type arrays use numTypes = ntype + 1,
 * the extra last type being the non-interacting filler type (as nbnxn_atomdata_t does).
 * perturbed[N] (may be NULL): which atoms are perturbed (the reference derives this from the
 * topology, mdatoms bPerturbed); NULL = "A and B parameters differ". */
NbnxmHostGrid* nbnxm_host_grid_create(int natoms, const float* x, const float* box, const float* qA,
                                      const float* qB, const int* typeA, const int* typeB, int ntype,
                                      const unsigned char* perturbed);
/* The grid of ONE DOMAIN of a decomposed run (nbnxm/gridset.cpp: local grid + non-local grid): natomsHome home atoms followed
 * by natomsHalo halo atoms, gridded as two zones (home slots first).  periodic[d] != 0: coordinates are wrapped into the box along
 * d and the list builder searches the images along d; 0 for a decomposed dimension, where the halo coordinates arrive already
 * shifted into the domain's own frame (domdec/gpuhaloexchange_impl_gpu.cu:62-88) and nothing is wrapped. */
NbnxmHostGrid* nbnxm_host_grid_create_dd(int natomsHome, int natomsHalo, const float* x, const float* box, const int* periodic,
                                         const float* qA, const float* qB, const int* typeA, const int* typeB, int ntype,
                                         const unsigned char* perturbed);
int  nbnxm_host_grid_num_atoms_home(const NbnxmHostGrid* g); /* padded slots of the home zone (= numAtomsLocal of the GPU module) */
void nbnxm_host_grid_free(NbnxmHostGrid* g);
int  nbnxm_host_grid_num_atoms(const NbnxmHostGrid* g);    /* padded: 64 * numSuperClusters */
int  nbnxm_host_grid_num_clusters(const NbnxmHostGrid* g); /* 8 * numSuperClusters */
/* Copies out grid-order arrays (any pointer may be NULL):
 *  xq[4Np]      x,y,z,q  with q = qA, perturbed atoms' q = 0 (atomdata.cpp:955-961)
 *  type[Np]     typeA, perturbed atoms and fillers = ntype (the zero type)
 *  qA,qB[Np], typeA,typeB[Np]   unmasked A/B parameters (fillers: 0 / ntype)
 *  atomIndices[Np]  grid index -> topology atom id, -1 for fillers (gridSet.atomIndices())
 *  fepBits[Np/8]    bit i = atom i of the cluster is perturbed (Grid::fepBits)
 *  xWrapped[3N]     topology-order coordinates after wrapping into the box (what the lists refer to) */
void nbnxm_host_grid_get(const NbnxmHostGrid* g, float* xq, int* type, float* qA, float* qB,
                         int* typeA, int* typeB, int* atomIndices, unsigned char* fepBits,
                         float* xWrapped);
/* Re-sorts new topology-order coordinates into grid order (x only changes between searches). */
void nbnxm_host_grid_update_xq(const NbnxmHostGrid* g, const float* x, float* xq);
/* 45 shift vectors for the box (pbcutil/pbc.cpp:1218-1233), 135 floats */
void nbnxm_host_shift_vectors(const float* box, float* shiftVec);

/* Number of interacting atom pairs within rc at the grid's (wrapped) coordinates: every pair once, minimum image, the
 * topology exclusions (CSR as for the list builder, may be NULL) taken out — the "useful" pairs of the reference's
 * benchmark (nbnxm/benchmark/bench_setup.cpp: the pair count of the plain-C kernel), next to the pairs a list makes a
 * kernel evaluate. */
long long nbnxm_host_count_pairs_within(const NbnxmHostGrid* g, float rc, const int* exclIndex, const int* exclAtoms);

/* ---- Ewald splitting parameters -------------------------------------------------------------- */
/* beta with erfc(beta rc) = rtol (ewald/ewald_utils.cpp:43-70, calc_ewaldcoeff_q) and the LJ-PME analogue with
 * exp(-x^2) (1 + x^2 + x^4 / 2) = rtol, x = beta rc (:72-112, calc_ewaldcoeff_lj) */
double nbnxm_host_calc_ewaldcoeff_q(double rc, double rtol);
double nbnxm_host_calc_ewaldcoeff_lj(double rc, double rtol);

/* ---- pair lists ---------------------------------------------------------------------------- */
typedef struct NbnxmHostPairlist NbnxmHostPairlist;

/* Builds the GPU-layout cluster pair list for one locality.
 *  exclIndex[N+1], exclAtoms[]: CSR topology exclusions (each atom lists its partners incl. itself or not)
 *  rlist: list cut-off; maxCjPackedPerSci: split i-entries longer than this (0 = never), the
 *         reference's list balancing (pairlist.cpp split_sci_entry / gpu_min_ci_balanced);
 *  carveFep != 0: reference behaviour for -fep gpu — perturbed pairs within rlistFep go to the
 *         atom-pair FEP list (i-entries capped at 64 j, pairlist.cpp:1509) and their bits are
 *         cleared in the cluster list (pairlist.cpp:1919);
 *  carveFep == 0: perturbed pairs stay in the cluster list with their topology exclusion bits
 *         (input of the fused kernel, to be used together with fepBits). */
NbnxmHostPairlist* nbnxm_host_pairlist_build(const NbnxmHostGrid* g, const int* exclIndex,
                                             const int* exclAtoms, float rlist,
                                             int maxCjPackedPerSci, int carveFep, float rlistFep);
/* The two lists of a domain (InteractionLocality::Local / NonLocal, nbnxm/pairlist.cpp:3960-4100): nonLocal == 0: home i-clusters x
 * home j-clusters, every pair once; nonLocal != 0: home i-clusters x halo j-clusters, all images (a halo cluster never is an
 * i-cluster).  Perturbed pairs stay in the cluster list (fused mode).  Exclusions: CSR over the grid's own atom numbering; with
 * whole molecules per domain no exclusion reaches into the halo, so the non-local list takes none. */
NbnxmHostPairlist* nbnxm_host_pairlist_build_dd(const NbnxmHostGrid* g, int nonLocal, const int* exclIndex, const int* exclAtoms,
                                                float rlist, int maxCjPackedPerSci);
void nbnxm_host_pairlist_free(NbnxmHostPairlist* pl);
/* sizes[0..5] = nsci, ncjPacked, nexcl, fep nri, fep nrj, number of set imask bits (cluster pairs) */
void nbnxm_host_pairlist_sizes(const NbnxmHostPairlist* pl, long long* sizes);
void nbnxm_host_pairlist_get(const NbnxmHostPairlist* pl, nbnxn_sci_t* sci,
                             nbnxn_cj_packed_t* cjPacked, nbnxn_excl_t* excl);
/* FEP list in TOPOLOGY atom ids (t_nblist, mdtypes/nblist.h:40-54): iinr[nri], shift[nri],
 * jindex[nri+1], jjnr[nrj], excl_fep[nrj] */
void nbnxm_host_pairlist_get_fep(const NbnxmHostPairlist* pl, int* iinr, int* shift, int* jindex,
                                 int* jjnr, int* excl_fep);

int nbnxm_host_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
