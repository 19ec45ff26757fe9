/*
 * Host-side producers for the MI355X non-bonded FEP path: synthetic water box, cluster grid,
 * GPU-layout pair list (sci / cjPacked / excl, split 2) and the perturbed atom-pair list.
 * C ABI in include/nbnxm_host.h (format definitions cited there).  The search algorithm is
 * this file's own: column grid + bounding boxes, one OpenMP task per i-super-cluster.
 */
#include "nbnxm_host.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace
{

constexpr int   CL      = NBNXM_GPU_CLUSTER_SIZE;
constexpr int   NCL     = NBNXM_GPU_NUM_CLUSTER_PER_SUPERCLUSTER;
constexpr int   SC      = CL * NCL; // atoms per super-cluster
constexpr int   JG      = NBNXM_GPU_JGROUP_SIZE;
constexpr int   c_maxNrjFep = 64;   // pairlist.cpp:1509
constexpr float c_farAway   = -100000.0F;

struct BB
{
    float lo[3], hi[3];
    bool  empty() const { return lo[0] > hi[0]; }
};

inline BB emptyBB()
{
    BB b;
    for (int d = 0; d < 3; d++)
    {
        b.lo[d] = 1e30F;
        b.hi[d] = -1e30F;
    }
    return b;
}

inline void extend(BB& b, const float* x)
{
    for (int d = 0; d < 3; d++)
    {
        b.lo[d] = std::min(b.lo[d], x[d]);
        b.hi[d] = std::max(b.hi[d], x[d]);
    }
}

inline float bbDist2(const BB& a, const float* shiftA, const BB& b)
{
    float d2 = 0;
    for (int d = 0; d < 3; d++)
    {
        const float dl = (a.lo[d] + shiftA[d]) - b.hi[d];
        const float dh = b.lo[d] - (a.hi[d] + shiftA[d]);
        const float m  = std::max(0.0F, std::max(dl, dh));
        d2 += m * m;
    }
    return d2;
}

// Small deterministic generator (splitmix64) so that the synthetic box is identical everywhere.
struct Rng
{
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL) {}
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z          = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    double uniform() { return (next() >> 11) * (1.0 / 9007199254740992.0); } // [0,1)
};

} // namespace

// One gridded set of atoms: columns in x/y over [lo, lo + size), atoms sorted along z inside a column.  A single-domain
// grid has one zone over the box; a domain of a decomposed run has two: home atoms, then halo atoms (as the reference grids
// its local and non-local atoms separately, nbnxm/gridset.cpp).
struct GridZone
{
    int              atomBegin = 0, atomEnd = 0; // input atoms [atomBegin, atomEnd)
    float            lo[2] = { 0, 0 }, size[2] = { 1, 1 };
    int              ncx = 1, ncy = 1;
    int              scBegin = 0, scEnd = 0;     // super-clusters of this zone
    std::vector<int> colScBegin;                 // ncx*ncy+1, absolute super-cluster indices
};

struct NbnxmHostGrid
{
    int                natoms = 0; // topology atoms
    int                ntype  = 0; // topology types
    float              box[3] = { 0, 0, 0 };
    bool               periodic[3] = { true, true, true }; // dims along which coordinates are wrapped and images are searched
    std::vector<GridZone> zones;
    int                nsc = 0;             // super-clusters (all zones)
    int                nscHome() const { return zones.empty() ? 0 : zones[0].scEnd; }
    std::vector<float> xw;                  // wrapped topology-order x (3N)
    std::vector<int>   atomIndices;         // grid -> topo (-1 filler)
    std::vector<int>   gridIndex;           // topo -> grid
    std::vector<float> xq;                  // 4*Np (q = unmasked qA here; masking in get)
    std::vector<float> qA, qB;              // Np
    std::vector<int>   typeA, typeB;        // Np
    std::vector<unsigned char> fepBits;     // Np/8
    std::vector<BB>    bbCluster;           // Np/8
    std::vector<BB>    bbSc;                // nsc
    int numAtomsPadded() const { return nsc * SC; }
};

struct NbnxmHostPairlist
{
    std::vector<nbnxn_sci_t>       sci;
    std::vector<nbnxn_cj_packed_t> cjPacked;
    std::vector<nbnxn_excl_t>      excl;
    std::vector<int>               iinr, shift, jindex, jjnr, exclFep;
    long long                      numClusterPairs = 0;
};

extern "C" {

int nbnxm_host_abi_version(void)
{
    return 1;
}

void nbnxm_host_shift_vectors(const float* box, float* shiftVec)
{
    int n = 0;
    for (int m = -1; m <= 1; m++)
    {
        for (int l = -1; l <= 1; l++)
        {
            for (int k = -2; k <= 2; k++, n++)
            {
                shiftVec[3 * n + 0] = k * box[0];
                shiftVec[3 * n + 1] = l * box[1];
                shiftVec[3 * n + 2] = m * box[2];
            }
        }
    }
}

void nbnxm_host_make_water_box(int nmx, int nmy, int nmz, double spacing, double jitter,
                               unsigned int seed, int numPerturbedMolecules, float* x, float* qA,
                               float* qB, int* typeA, int* typeB, int* molId, float* box)
{
    const int nmol = nmx * nmy * nmz;
    box[0]         = static_cast<float>(nmx * spacing);
    box[1]         = static_cast<float>(nmy * spacing);
    box[2]         = static_cast<float>(nmz * spacing);
    Rng rng(seed);

    // SPC/E geometry (bench_system.cpp recipe): r(OH) = 0.1 nm, HOH = 109.47 deg
    const double rOH   = 0.1;
    const double halfA = 0.5 * 109.47 * M_PI / 180.0;
    const double h1[3] = { rOH * std::sin(halfA), 0.0, rOH * std::cos(halfA) };
    const double h2[3] = { -rOH * std::sin(halfA), 0.0, rOH * std::cos(halfA) };

    std::vector<double> pos(static_cast<size_t>(nmol) * 9);
    auto molIndex = [&](int i, int j, int k) { return (k * nmy + j) * nmx + i; };

    for (int k = 0; k < nmz; k++)
    {
        for (int j = 0; j < nmy; j++)
        {
            for (int i = 0; i < nmx; i++)
            {
                const int m = molIndex(i, j, k);
                double    c[3];
                c[0] = (i + 0.5) * spacing + jitter * (2 * rng.uniform() - 1);
                c[1] = (j + 0.5) * spacing + jitter * (2 * rng.uniform() - 1);
                c[2] = (k + 0.5) * spacing + jitter * (2 * rng.uniform() - 1);
                // pick, out of a few random orientations, the one that keeps the hydrogens
                // farthest from the atoms of the already placed lattice neighbours
                double best[9];
                double bestD = -1;
                for (int trial = 0; trial < 12; trial++)
                {
                    // random unit quaternion
                    double q[4], n2 = 0;
                    do
                    {
                        n2 = 0;
                        for (double& qi : q)
                        {
                            qi = 2 * rng.uniform() - 1;
                            n2 += qi * qi;
                        }
                    } while (n2 > 1.0 || n2 < 1e-4);
                    const double inv = 1.0 / std::sqrt(n2);
                    for (double& qi : q) { qi *= inv; }
                    const double R[3][3] = {
                        { 1 - 2 * (q[2] * q[2] + q[3] * q[3]), 2 * (q[1] * q[2] - q[0] * q[3]), 2 * (q[1] * q[3] + q[0] * q[2]) },
                        { 2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[1] * q[1] + q[3] * q[3]), 2 * (q[2] * q[3] - q[0] * q[1]) },
                        { 2 * (q[1] * q[3] - q[0] * q[2]), 2 * (q[2] * q[3] + q[0] * q[1]), 1 - 2 * (q[1] * q[1] + q[2] * q[2]) }
                    };
                    double cand[9];
                    for (int d = 0; d < 3; d++)
                    {
                        cand[d]     = c[d];
                        cand[3 + d] = c[d] + R[d][0] * h1[0] + R[d][1] * h1[1] + R[d][2] * h1[2];
                        cand[6 + d] = c[d] + R[d][0] * h2[0] + R[d][1] * h2[1] + R[d][2] * h2[2];
                    }
                    double dmin = 1e9;
                    for (int dk = -1; dk <= 1; dk++)
                    {
                        for (int dj = -1; dj <= 1; dj++)
                        {
                            for (int di = -1; di <= 1; di++)
                            {
                                const int ii = (i + di + nmx) % nmx, jj = (j + dj + nmy) % nmy, kk = (k + dk + nmz) % nmz;
                                const int mo = molIndex(ii, jj, kk);
                                if (mo >= m) { continue; } // not placed yet (or self)
                                for (int a = 1; a < 3; a++)
                                {
                                    for (int b = 0; b < 3; b++)
                                    {
                                        double d2 = 0;
                                        for (int d = 0; d < 3; d++)
                                        {
                                            double dd = cand[3 * a + d] - pos[static_cast<size_t>(mo) * 9 + 3 * b + d];
                                            dd -= box[d] * std::nearbyint(dd / box[d]);
                                            d2 += dd * dd;
                                        }
                                        dmin = std::min(dmin, d2);
                                    }
                                }
                            }
                        }
                    }
                    if (dmin > bestD)
                    {
                        bestD = dmin;
                        std::memcpy(best, cand, sizeof(best));
                    }
                    if (bestD > 0.16 * 0.16) { break; }
                }
                std::memcpy(&pos[static_cast<size_t>(m) * 9], best, sizeof(best));
            }
        }
    }

    // the ligand: the molecules nearest to the box centre
    std::vector<std::pair<double, int>> byDist(nmol);
    for (int m = 0; m < nmol; m++)
    {
        double d2 = 0;
        for (int d = 0; d < 3; d++)
        {
            const double dd = pos[static_cast<size_t>(m) * 9 + d] - 0.5 * box[d];
            d2 += dd * dd;
        }
        byDist[m] = { d2, m };
    }
    std::sort(byDist.begin(), byDist.end());
    std::vector<char> perturbed(nmol, 0);
    for (int p = 0; p < std::min(numPerturbedMolecules, nmol); p++) { perturbed[byDist[p].second] = 1; }

    for (int m = 0; m < nmol; m++)
    {
        for (int a = 0; a < 3; a++)
        {
            const int n = 3 * m + a;
            for (int d = 0; d < 3; d++) { x[3 * n + d] = static_cast<float>(pos[static_cast<size_t>(m) * 9 + 3 * a + d]); }
            qA[n]    = (a == 0) ? -0.8476F : 0.4238F;
            typeA[n] = (a == 0) ? 0 : 1;
            qB[n]    = perturbed[m] ? 0.0F : qA[n];
            typeB[n] = perturbed[m] ? 2 : typeA[n];
            molId[n] = m;
        }
    }
}

/* ---- Ewald splitting parameters ------------------------------------------------------------ */

} // extern "C"
namespace
{
// smallest x >= 0 with tail(x) <= rtol for a tail function that decreases monotonically to 0
template<typename Tail>
double solveTail(Tail tail, double rtol)
{
    double hi = 1.0;
    while (tail(hi) > rtol) { hi *= 2.0; }
    double lo = 0.0;
    for (int it = 0; it < 200 && hi - lo > 1e-16 * hi; it++)
    {
        const double mid = 0.5 * (lo + hi);
        if (tail(mid) > rtol) { lo = mid; }
        else { hi = mid; }
    }
    return 0.5 * (lo + hi);
}
} // namespace
extern "C" {

double nbnxm_host_calc_ewaldcoeff_q(double rc, double rtol)
{
    return solveTail([](double x) { return std::erfc(x); }, rtol) / rc;
}

double nbnxm_host_calc_ewaldcoeff_lj(double rc, double rtol)
{
    return solveTail([](double x) { const double x2 = x * x; return std::exp(-x2) * (1.0 + x2 + 0.5 * x2 * x2); }, rtol) / rc;
}

/* ---- pair count ----------------------------------------------------------------------------- */

long long nbnxm_host_count_pairs_within(const NbnxmHostGrid* g, float rc, const int* exclIndex, const int* exclAtoms)
{
    // uniform cells of edge >= rc over the wrapped coordinates; every pair once (minimum image), excluded pairs taken out
    const int   n = g->natoms;
    int         nc[3];
    float       cs[3];
    for (int d = 0; d < 3; d++)
    {
        nc[d] = std::max(1, static_cast<int>(g->box[d] / rc));
        cs[d] = g->box[d] / nc[d];
    }
    const int              ncell = nc[0] * nc[1] * nc[2];
    std::vector<int>       cellOf(n), start(ncell + 1, 0), order(n);
    for (int a = 0; a < n; a++)
    {
        int c[3];
        for (int d = 0; d < 3; d++) { c[d] = std::min(nc[d] - 1, static_cast<int>(g->xw[3 * a + d] / cs[d])); }
        cellOf[a] = (c[0] * nc[1] + c[1]) * nc[2] + c[2];
        start[cellOf[a] + 1]++;
    }
    for (int c = 0; c < ncell; c++) { start[c + 1] += start[c]; }
    {
        std::vector<int> fill(start.begin(), start.end() - 1);
        for (int a = 0; a < n; a++) { order[fill[cellOf[a]]++] = a; }
    }
    const float rc2   = rc * rc;
    long long   count = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : count)
    for (int c = 0; c < ncell; c++)
    {
        const int cx = c / (nc[1] * nc[2]), cy = (c / nc[2]) % nc[1], cz = c % nc[2];
        // neighbour cells without duplicates (few cells per dimension: the same cell may be reached through several offsets)
        std::vector<int> nbr;
        for (int dx = -1; dx <= 1; dx++)
        {
            for (int dy = -1; dy <= 1; dy++)
            {
                for (int dz = -1; dz <= 1; dz++)
                {
                    const int o = (((cx + dx + nc[0]) % nc[0]) * nc[1] + (cy + dy + nc[1]) % nc[1]) * nc[2] + (cz + dz + nc[2]) % nc[2];
                    if (std::find(nbr.begin(), nbr.end(), o) == nbr.end()) { nbr.push_back(o); }
                }
            }
        }
        for (int ia = start[c]; ia < start[c + 1]; ia++)
        {
            const int    a  = order[ia];
            const float* xa = &g->xw[3 * a];
            for (int o : nbr)
            {
                for (int ib = start[o]; ib < start[o + 1]; ib++)
                {
                    const int b = order[ib];
                    if (b <= a) { continue; }
                    float r2 = 0;
                    for (int d = 0; d < 3; d++)
                    {
                        float dd = xa[d] - g->xw[3 * b + d];
                        dd -= g->box[d] * std::nearbyint(dd / g->box[d]);
                        r2 += dd * dd;
                    }
                    if (r2 < rc2)
                    {
                        bool excluded = false;
                        if (exclIndex != nullptr)
                        {
                            for (int k = exclIndex[a]; k < exclIndex[a + 1] && !excluded; k++) { excluded = (exclAtoms[k] == b); }
                        }
                        if (!excluded) { count++; }
                    }
                }
            }
        }
    }
    return count;
}

/* ---- grid ---------------------------------------------------------------------------------- */

} // extern "C"

namespace
{
// Grids the atoms [zone.atomBegin, zone.atomEnd) of g->xw: fills zone.colScBegin, g->atomIndices, g->gridIndex for its slots.
void gridZone(NbnxmHostGrid* g, GridZone& zone, double density)
{
    const std::vector<float>& xw   = g->xw;
    const int                 n    = zone.atomEnd - zone.atomBegin;
    const double              colSize = std::cbrt(SC / density);
    zone.ncx                       = std::max(1, static_cast<int>(zone.size[0] / colSize));
    zone.ncy                       = std::max(1, static_cast<int>(zone.size[1] / colSize));
    const int ncol                 = zone.ncx * zone.ncy;
    std::vector<std::vector<int>> colAtoms(ncol);
    for (int a = zone.atomBegin; a < zone.atomEnd; a++)
    {
        const int cx = std::min(zone.ncx - 1, std::max(0, static_cast<int>((xw[3 * a + 0] - zone.lo[0]) / zone.size[0] * zone.ncx)));
        const int cy = std::min(zone.ncy - 1, std::max(0, static_cast<int>((xw[3 * a + 1] - zone.lo[1]) / zone.size[1] * zone.ncy)));
        colAtoms[cx * zone.ncy + cy].push_back(a);
    }
    zone.colScBegin.assign(ncol + 1, zone.scBegin);
    for (int c = 0; c < ncol; c++)
    {
        zone.colScBegin[c + 1] = zone.colScBegin[c] + static_cast<int>((colAtoms[c].size() + SC - 1) / SC);
    }
    zone.scEnd = zone.colScBegin[ncol];
    g->atomIndices.resize(static_cast<size_t>(zone.scEnd) * SC, -1);
    (void)n;

    auto sortBy = [&xw](int* begin, int* end, int dim) {
        std::stable_sort(begin, end, [&xw, dim](int a, int b) { return xw[3 * a + dim] < xw[3 * b + dim]; });
    };
#pragma omp parallel for schedule(dynamic)
    for (int c = 0; c < ncol; c++)
    {
        std::vector<int>& atoms = colAtoms[c];
        sortBy(atoms.data(), atoms.data() + atoms.size(), 2);
        const int nscCol = zone.colScBegin[c + 1] - zone.colScBegin[c];
        for (int s = 0; s < nscCol; s++)
        {
            int*      first = atoms.data() + static_cast<size_t>(s) * SC;
            const int n     = std::min<int>(SC, static_cast<int>(atoms.size()) - s * SC);
            // 2 x 2 x 2 sub-sort: halves in z (already sorted), then y, then x.  Real atoms are
            // dealt to the 8 clusters as evenly as possible so that a partly filled super-cluster
            // still has compact clusters.
            const int nz[2] = { (n + 1) / 2, n / 2 };
            int       off   = 0;
            int       slot  = (zone.colScBegin[c] + s) * SC;
            for (int iz = 0; iz < 2; iz++)
            {
                int* zb = first + off;
                sortBy(zb, zb + nz[iz], 1);
                const int ny[2] = { (nz[iz] + 1) / 2, nz[iz] / 2 };
                int       offy  = 0;
                for (int iy = 0; iy < 2; iy++)
                {
                    int* yb = zb + offy;
                    sortBy(yb, yb + ny[iy], 0);
                    const int nx[2] = { (ny[iy] + 1) / 2, ny[iy] / 2 };
                    int       offx  = 0;
                    for (int ix = 0; ix < 2; ix++)
                    {
                        for (int k = 0; k < nx[ix]; k++)
                        {
                            const int a          = yb[offx + k];
                            g->atomIndices[slot + k] = a;
                            g->gridIndex[a]      = slot + k;
                        }
                        offx += nx[ix];
                        slot += CL;
                    }
                    offy += ny[iy];
                }
                off += nz[iz];
            }
        }
    }
}

// natomsHome atoms of the domain, then the halo atoms; periodic[d]: wrap coordinates into the box along d and let the list
// builder search the images along d (a decomposed dimension has neither: its halo arrives already shifted)
NbnxmHostGrid* gridCreate(int natoms, int natomsHome, const float* x, const float* box, const int* periodic, const float* qA,
                          const float* qB, const int* typeA, const int* typeB, int ntype, const unsigned char* perturbed)
{
    auto* g   = new NbnxmHostGrid;
    g->natoms = natoms;
    g->ntype  = ntype;
    for (int d = 0; d < 3; d++)
    {
        g->box[d]      = box[d];
        g->periodic[d] = (periodic == nullptr) || (periodic[d] != 0);
    }
    g->xw.resize(static_cast<size_t>(natoms) * 3);
    for (int a = 0; a < natoms; a++)
    {
        for (int d = 0; d < 3; d++)
        {
            float v = x[3 * a + d];
            if (g->periodic[d])
            {
                v -= box[d] * std::floor(v / box[d]);
                if (v >= box[d]) { v = 0; }
            }
            g->xw[3 * a + d] = v;
        }
    }
    g->gridIndex.assign(natoms, -1);
    const int numZones = (natomsHome < natoms) ? 2 : 1;
    g->zones.resize(numZones);
    double density = 0;
    for (int z = 0; z < numZones; z++)
    {
        GridZone& zone = g->zones[z];
        zone.atomBegin = (z == 0) ? 0 : natomsHome;
        zone.atomEnd   = (z == 0) ? natomsHome : natoms;
        zone.scBegin   = (z == 0) ? 0 : g->zones[0].scEnd;
        float lo[3], hi[3];
        for (int d = 0; d < 3; d++)
        {
            lo[d] = g->periodic[d] ? 0.0F : 1e30F;
            hi[d] = g->periodic[d] ? box[d] : -1e30F;
        }
        for (int a = zone.atomBegin; a < zone.atomEnd; a++)
        {
            for (int d = 0; d < 3; d++)
            {
                if (!g->periodic[d])
                {
                    lo[d] = std::min(lo[d], g->xw[3 * a + d]);
                    hi[d] = std::max(hi[d], g->xw[3 * a + d]);
                }
            }
        }
        for (int d = 0; d < 3; d++)
        {
            if (hi[d] <= lo[d]) { hi[d] = lo[d] + 1e-3F; }
        }
        for (int d = 0; d < 2; d++)
        {
            zone.lo[d]   = lo[d];
            zone.size[d] = (hi[d] - lo[d]) * (g->periodic[d] ? 1.0F : 1.0001F);
        }
        if (z == 0)
        {
            // number density of the home atoms; the halo (a shell around them) is gridded with columns of the same size
            density = std::max(1, zone.atomEnd - zone.atomBegin)
                      / (static_cast<double>(hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]));
        }
        gridZone(g, zone, density);
    }
    g->nsc       = g->zones.back().scEnd;
    const int np = g->numAtomsPadded();
    g->atomIndices.resize(np, -1);

    g->xq.assign(static_cast<size_t>(np) * 4, 0.0F);
    g->qA.assign(np, 0.0F);
    g->qB.assign(np, 0.0F);
    g->typeA.assign(np, ntype);
    g->typeB.assign(np, ntype);
    g->fepBits.assign(np / CL, 0);
    g->bbCluster.assign(np / CL, emptyBB());
    g->bbSc.assign(g->nsc, emptyBB());
    for (int gi = 0; gi < np; gi++)
    {
        const int a = g->atomIndices[gi];
        if (a >= 0)
        {
            for (int d = 0; d < 3; d++) { g->xq[4 * static_cast<size_t>(gi) + d] = g->xw[3 * a + d]; }
            g->xq[4 * static_cast<size_t>(gi) + 3] = qA[a];
            g->qA[gi]                              = qA[a];
            g->qB[gi]                              = qB[a];
            g->typeA[gi]                           = typeA[a];
            g->typeB[gi]                           = typeB[a];
            const bool isPerturbed = perturbed ? (perturbed[a] != 0) : (qA[a] != qB[a] || typeA[a] != typeB[a]);
            if (isPerturbed) { g->fepBits[gi / CL] |= (1U << (gi % CL)); }
            extend(g->bbCluster[gi / CL], &g->xq[4 * static_cast<size_t>(gi)]);
            extend(g->bbSc[gi / SC], &g->xq[4 * static_cast<size_t>(gi)]);
        }
        else
        {
            // fillers: far away, mutually separated, never inside any cut-off
            g->xq[4 * static_cast<size_t>(gi) + 0] = c_farAway - 4.0F * (gi % SC);
            g->xq[4 * static_cast<size_t>(gi) + 1] = c_farAway;
            g->xq[4 * static_cast<size_t>(gi) + 2] = c_farAway - 4.0F * ((gi / SC) % 8192);
        }
    }
    return g;
}
} // namespace

extern "C" {

NbnxmHostGrid* nbnxm_host_grid_create(int natoms, const float* x, const float* box, const float* qA,
                                      const float* qB, const int* typeA, const int* typeB, int ntype,
                                      const unsigned char* perturbed)
{
    return gridCreate(natoms, natoms, x, box, nullptr, qA, qB, typeA, typeB, ntype, perturbed);
}

NbnxmHostGrid* nbnxm_host_grid_create_dd(int natomsHome, int natomsHalo, const float* x, const float* box, const int* periodic,
                                         const float* qA, const float* qB, const int* typeA, const int* typeB, int ntype,
                                         const unsigned char* perturbed)
{
    return gridCreate(natomsHome + natomsHalo, natomsHome, x, box, periodic, qA, qB, typeA, typeB, ntype, perturbed);
}

int nbnxm_host_grid_num_atoms_home(const NbnxmHostGrid* g)
{
    return g->nscHome() * SC;
}

void nbnxm_host_grid_free(NbnxmHostGrid* g)
{
    delete g;
}

int nbnxm_host_grid_num_atoms(const NbnxmHostGrid* g)
{
    return g->numAtomsPadded();
}

int nbnxm_host_grid_num_clusters(const NbnxmHostGrid* g)
{
    return g->numAtomsPadded() / CL;
}

void nbnxm_host_grid_get(const NbnxmHostGrid* g, float* xq, int* type, float* qA, float* qB,
                         int* typeA, int* typeB, int* atomIndices, unsigned char* fepBits,
                         float* xWrapped)
{
    const int np = g->numAtomsPadded();
    for (int gi = 0; gi < np; gi++)
    {
        const bool pert = (g->fepBits[gi / CL] >> (gi % CL)) & 1U;
        if (xq)
        {
            for (int d = 0; d < 3; d++) { xq[4 * static_cast<size_t>(gi) + d] = g->xq[4 * static_cast<size_t>(gi) + d]; }
            xq[4 * static_cast<size_t>(gi) + 3] = pert ? 0.0F : g->xq[4 * static_cast<size_t>(gi) + 3];
        }
        if (type) { type[gi] = pert ? g->ntype : g->typeA[gi]; }
    }
    if (qA) { std::copy(g->qA.begin(), g->qA.end(), qA); }
    if (qB) { std::copy(g->qB.begin(), g->qB.end(), qB); }
    if (typeA) { std::copy(g->typeA.begin(), g->typeA.end(), typeA); }
    if (typeB) { std::copy(g->typeB.begin(), g->typeB.end(), typeB); }
    if (atomIndices) { std::copy(g->atomIndices.begin(), g->atomIndices.end(), atomIndices); }
    if (fepBits) { std::copy(g->fepBits.begin(), g->fepBits.end(), fepBits); }
    if (xWrapped) { std::copy(g->xw.begin(), g->xw.end(), xWrapped); }
}

void nbnxm_host_grid_update_xq(const NbnxmHostGrid* g, const float* x, float* xq)
{
    const int np = g->numAtomsPadded();
    for (int gi = 0; gi < np; gi++)
    {
        const int a = g->atomIndices[gi];
        if (a >= 0)
        {
            for (int d = 0; d < 3; d++) { xq[4 * static_cast<size_t>(gi) + d] = x[3 * a + d]; }
        }
    }
}

/* ---- pair list ----------------------------------------------------------------------------- */

namespace
{

struct SciWork
{
    std::vector<nbnxn_sci_t>       sci; // cjPacked indices are local to this work item
    std::vector<nbnxn_cj_packed_t> cjPacked;
    std::vector<nbnxn_excl_t>      excl; // local indices start at 1 (0 = shared all-ones)
    std::vector<int>               iinr, shift, jindexLocal, jjnr, exclFep;
    long long                      numClusterPairs = 0;
};

inline nbnxn_excl_t& exclusionMask(SciWork& w, int group, int half)
{
    nbnxn_im_ei_t& e = w.cjPacked[group].imei[half];
    if (e.excl_ind == 0)
    {
        nbnxn_excl_t m;
        for (unsigned int& p : m.pair) { p = 0xffffffffU; }
        w.excl.push_back(m);
        e.excl_ind = static_cast<int>(w.excl.size()); // local index + 1
    }
    return w.excl[e.excl_ind - 1];
}

} // namespace

} // extern "C"

namespace
{
// jZone: which zone the j-clusters come from.  0: the home zone, every pair once (half of the shift vectors, and on the
// central image only j super-clusters >= i); 1: the halo zone — its clusters never act as i-clusters, so every image is searched.
NbnxmHostPairlist* pairlistBuild(const NbnxmHostGrid* g, const int jZone, const int* exclIndex, const int* exclAtoms, float rlist,
                                 int maxCjPackedPerSci, int carveFep, float rlistFep)
{
    const float rl2    = rlist * rlist;
    const float rlFep2 = rlistFep * rlistFep;
    float       shiftVec[3 * NBNXM_NUM_SHIFT_VECTORS];
    nbnxm_host_shift_vectors(g->box, shiftVec);

    const int       nscI  = g->nscHome(); // only home super-clusters act as i
    const GridZone& zoneJ = g->zones[jZone];
    const bool      halfList = (jZone == 0);
    // extent of the j zone's atoms (to skip images that cannot reach it)
    BB bbZone = emptyBB();
    for (int scj = zoneJ.scBegin; scj < zoneJ.scEnd; scj++)
    {
        if (!g->bbSc[scj].empty())
        {
            extend(bbZone, g->bbSc[scj].lo);
            extend(bbZone, g->bbSc[scj].hi);
        }
    }

    std::vector<SciWork> work(nscI);

#pragma omp parallel for schedule(dynamic, 4)
    for (int sci = 0; sci < nscI; sci++)
    {
        SciWork&  w   = work[sci];
        const BB& bbI = g->bbSc[sci];
        if (bbI.empty() || bbZone.empty()) { continue; }
        std::vector<int> cand;
        std::vector<int> entryCj; // cj of the open entry, ascending
        for (int s = (halfList ? NBNXM_CENTRAL_SHIFT_INDEX : 0); s < NBNXM_NUM_SHIFT_VECTORS; s++)
        {
            const float* S       = &shiftVec[3 * s];
            const bool   central = (s == NBNXM_CENTRAL_SHIFT_INDEX);
            // images only along the periodic dimensions; the shifted i box must come within rlist of the j zone's atoms
            bool reach = true;
            for (int d = 0; d < 3; d++)
            {
                if (!g->periodic[d] && S[d] != 0.0F) { reach = false; }
                if (bbI.lo[d] + S[d] - rlist > bbZone.hi[d] || bbI.hi[d] + S[d] + rlist < bbZone.lo[d]) { reach = false; }
            }
            if (!reach) { continue; }
            const int cx0 = std::max(0, static_cast<int>(std::floor((bbI.lo[0] + S[0] - rlist - zoneJ.lo[0]) / zoneJ.size[0] * zoneJ.ncx)));
            const int cx1 = std::min(zoneJ.ncx - 1, static_cast<int>(std::floor((bbI.hi[0] + S[0] + rlist - zoneJ.lo[0]) / zoneJ.size[0] * zoneJ.ncx)));
            const int cy0 = std::max(0, static_cast<int>(std::floor((bbI.lo[1] + S[1] - rlist - zoneJ.lo[1]) / zoneJ.size[1] * zoneJ.ncy)));
            const int cy1 = std::min(zoneJ.ncy - 1, static_cast<int>(std::floor((bbI.hi[1] + S[1] + rlist - zoneJ.lo[1]) / zoneJ.size[1] * zoneJ.ncy)));
            cand.clear();
            for (int cx = cx0; cx <= cx1; cx++)
            {
                for (int cy = cy0; cy <= cy1; cy++)
                {
                    const int col = cx * zoneJ.ncy + cy;
                    for (int scj = zoneJ.colScBegin[col]; scj < zoneJ.colScBegin[col + 1]; scj++)
                    {
                        if (halfList && central && scj < sci) { continue; }
                        if (bbDist2(bbI, S, g->bbSc[scj]) < rl2) { cand.push_back(scj); }
                    }
                }
            }
            if (cand.empty()) { continue; }
            std::sort(cand.begin(), cand.end());

            const int groupBegin = static_cast<int>(w.cjPacked.size());
            entryCj.clear();
            for (int scj : cand)
            {
                for (int cjl = 0; cjl < NCL; cjl++)
                {
                    const int cj = scj * NCL + cjl;
                    if (g->bbCluster[cj].empty()) { continue; }
                    unsigned int mask = 0;
                    for (int cil = 0; cil < NCL; cil++)
                    {
                        if (halfList && central && scj == sci && cjl < cil) { continue; }
                        const int ci = sci * NCL + cil;
                        if (g->bbCluster[ci].empty()) { continue; }
                        if (bbDist2(g->bbCluster[ci], S, g->bbCluster[cj]) < rl2) { mask |= (1U << cil); }
                    }
                    if (mask == 0) { continue; }
                    const int pos = static_cast<int>(entryCj.size());
                    const int jm  = pos % JG;
                    if (jm == 0)
                    {
                        nbnxn_cj_packed_t grp;
                        std::memset(&grp, 0, sizeof(grp));
                        w.cjPacked.push_back(grp);
                    }
                    nbnxn_cj_packed_t& grp = w.cjPacked.back();
                    grp.cj[jm]             = cj;
                    grp.imei[0].imask |= mask << (jm * NCL);
                    grp.imei[1].imask = grp.imei[0].imask;
                    entryCj.push_back(cj);
                    w.numClusterPairs += __builtin_popcount(mask);
                }
            }
            if (entryCj.empty()) { continue; }
            const int groupEnd = static_cast<int>(w.cjPacked.size());

            auto pairBit = [&](int pos, int cil) { return 1U << ((pos % JG) * NCL + cil); };
            auto clearPair = [&](int pos, int cil, int ic, int jc) {
                nbnxn_excl_t& m = exclusionMask(w, groupBegin + pos / JG, jc / 4);
                m.pair[(jc & 3) * CL + ic] &= ~pairBit(pos, cil);
            };
            auto pairIncluded = [&](int pos, int cil, int ic, int jc) -> bool {
                const int excl_ind = w.cjPacked[groupBegin + pos / JG].imei[jc / 4].excl_ind;
                if (excl_ind == 0) { return true; }
                return (w.excl[excl_ind - 1].pair[(jc & 3) * CL + ic] & pairBit(pos, cil)) != 0;
            };

            // diagonal: on the central image a cluster paired with itself keeps only j > i
            if (central && halfList)
            {
                for (int cil = 0; cil < NCL; cil++)
                {
                    const int ci = sci * NCL + cil;
                    auto      it = std::lower_bound(entryCj.begin(), entryCj.end(), ci);
                    if (it == entryCj.end() || *it != ci) { continue; }
                    const int pos = static_cast<int>(it - entryCj.begin());
                    for (int ic = 0; ic < CL; ic++)
                    {
                        for (int jc = 0; jc <= ic; jc++) { clearPair(pos, cil, ic, jc); }
                    }
                }
            }
            // topology exclusions
            if (exclIndex != nullptr)
            {
                for (int il = 0; il < SC; il++)
                {
                    const int gi = sci * SC + il;
                    const int a  = g->atomIndices[gi];
                    if (a < 0) { continue; }
                    const int cil = il / CL, ic = il % CL;
                    for (int e = exclIndex[a]; e < exclIndex[a + 1]; e++)
                    {
                        const int b = exclAtoms[e];
                        if (b == a) { continue; }
                        const int gj = g->gridIndex[b];
                        const int cj = gj / CL, jc = gj % CL;
                        auto      it = std::lower_bound(entryCj.begin(), entryCj.end(), cj);
                        if (it == entryCj.end() || *it != cj) { continue; }
                        const int pos = static_cast<int>(it - entryCj.begin());
                        if (!(w.cjPacked[groupBegin + pos / JG].imei[0].imask & pairBit(pos, cil))) { continue; }
                        // only the minimum image of an excluded pair is excluded
                        bool minImage = true;
                        for (int d = 0; d < 3; d++)
                        {
                            const float dd = g->xq[4 * static_cast<size_t>(gi) + d] + S[d] - g->xq[4 * static_cast<size_t>(gj) + d];
                            if (g->periodic[d] && std::fabs(dd) > 0.5F * g->box[d]) { minImage = false; }
                        }
                        if (!minImage) { continue; }
                        if (halfList && central && cj == sci * NCL + cil && jc <= ic) { continue; } // already cleared
                        clearPair(pos, cil, ic, jc);
                    }
                }
            }

            // perturbed pairs -> atom-pair list (make_fep_list semantics, GPU flavour)
            if (carveFep)
            {
                for (int il = 0; il < SC; il++)
                {
                    const int gi = sci * SC + il;
                    const int ai = g->atomIndices[gi];
                    if (ai < 0) { continue; }
                    const int  cil  = il / CL, ic = il % CL;
                    const bool fepI = (g->fepBits[gi / CL] >> ic) & 1U;
                    int        nriOpenStart = -1; // first j of the open i-entry
                    for (int pos = 0; pos < static_cast<int>(entryCj.size()); pos++)
                    {
                        if (!(w.cjPacked[groupBegin + pos / JG].imei[0].imask & pairBit(pos, cil))) { continue; }
                        const int cj = entryCj[pos];
                        if (!fepI && g->fepBits[cj] == 0) { continue; }
                        for (int jc = 0; jc < CL; jc++)
                        {
                            const int gj = cj * CL + jc;
                            const int aj = g->atomIndices[gj];
                            if (aj < 0) { continue; }
                            if (!(fepI || ((g->fepBits[cj] >> jc) & 1U))) { continue; }
                            if (halfList && central && gj < gi) { continue; }
                            float d2 = 0;
                            for (int d = 0; d < 3; d++)
                            {
                                const float dd = g->xq[4 * static_cast<size_t>(gj) + d] - (g->xq[4 * static_cast<size_t>(gi) + d] + S[d]);
                                d2 += dd * dd;
                            }
                            if (d2 < rlFep2)
                            {
                                if (nriOpenStart < 0 || static_cast<int>(w.jjnr.size()) - nriOpenStart >= c_maxNrjFep)
                                {
                                    w.iinr.push_back(ai);
                                    w.shift.push_back(s);
                                    w.jindexLocal.push_back(static_cast<int>(w.jjnr.size()));
                                    nriOpenStart = static_cast<int>(w.jjnr.size());
                                }
                                w.jjnr.push_back(aj);
                                w.exclFep.push_back(pairIncluded(pos, cil, ic, jc) ? 1 : 0);
                            }
                            clearPair(pos, cil, ic, jc);
                        }
                    }
                }
            }

            // close the entry, optionally split into chunks (list balancing); when the perturbed pairs stay
            // in the cluster list, entries whose i-clusters hold perturbed atoms cost several times more per
            // j-cluster and are the kernel's critical path: they are cut into single-group entries
            bool perturbedI = false;
            for (int cil = 0; cil < NCL; cil++) { perturbedI = perturbedI || (g->fepBits[sci * NCL + cil] != 0); }
            int chunk = (maxCjPackedPerSci > 0) ? maxCjPackedPerSci : (groupEnd - groupBegin);
            if (!carveFep && perturbedI && maxCjPackedPerSci > 0) { chunk = 1; }
            for (int b = groupBegin; b < groupEnd; b += chunk)
            {
                nbnxn_sci_t e;
                e.sci           = sci;
                e.shift         = s;
                e.cjPackedBegin = b;
                e.cjPackedEnd   = std::min(groupEnd, b + chunk);
                w.sci.push_back(e);
            }
        }
    }

    // merge in super-cluster order
    auto* pl = new NbnxmHostPairlist;
    nbnxn_excl_t allOnes;
    for (unsigned int& p : allOnes.pair) { p = 0xffffffffU; }
    pl->excl.push_back(allOnes);
    pl->jindex.push_back(0);
    for (int sci = 0; sci < nscI; sci++)
    {
        SciWork&  w       = work[sci];
        const int cjBase  = static_cast<int>(pl->cjPacked.size());
        const int exBase  = static_cast<int>(pl->excl.size()) - 1; // local (1-based) -> global
        const int jBase   = static_cast<int>(pl->jjnr.size());
        for (nbnxn_cj_packed_t grp : w.cjPacked)
        {
            for (auto& im : grp.imei)
            {
                if (im.excl_ind != 0) { im.excl_ind += exBase; }
            }
            pl->cjPacked.push_back(grp);
        }
        pl->excl.insert(pl->excl.end(), w.excl.begin(), w.excl.end());
        for (nbnxn_sci_t e : w.sci)
        {
            e.cjPackedBegin += cjBase;
            e.cjPackedEnd += cjBase;
            pl->sci.push_back(e);
        }
        for (size_t n = 0; n < w.iinr.size(); n++)
        {
            pl->iinr.push_back(w.iinr[n]);
            pl->shift.push_back(w.shift[n]);
            const int jEnd = (n + 1 < w.iinr.size()) ? w.jindexLocal[n + 1] : static_cast<int>(w.jjnr.size());
            pl->jindex.push_back(jBase + jEnd);
        }
        pl->jjnr.insert(pl->jjnr.end(), w.jjnr.begin(), w.jjnr.end());
        pl->exclFep.insert(pl->exclFep.end(), w.exclFep.begin(), w.exclFep.end());
        pl->numClusterPairs += w.numClusterPairs;
        w = SciWork();
    }
    // most expensive entries first (the reference sorts the sci list by work, pairlist.cpp sort_sci);
    // with un-carved lists, entries with perturbed i-clusters count 4x
    auto cost = [&](const nbnxn_sci_t& e) {
        int w = e.cjPackedEnd - e.cjPackedBegin;
        if (carveFep) { return w; }
        bool perturbedI = false;
        for (int cil = 0; cil < NCL; cil++) { perturbedI = perturbedI || (g->fepBits[e.sci * NCL + cil] != 0); }
        if (perturbedI) { return 4 * w + 1000000; }
        // a j-cluster with perturbed atoms costs about as much as four ordinary packed groups
        int pertJ = 0;
        for (int jp = e.cjPackedBegin; jp < e.cjPackedEnd; jp++)
        {
            for (int jm = 0; jm < JG; jm++)
            {
                const unsigned mj = (pl->cjPacked[jp].imei[0].imask >> (jm * NCL)) & 0xFFU;
                if (mj != 0U && g->fepBits[pl->cjPacked[jp].cj[jm]] != 0) { pertJ++; }
            }
        }
        return w + 4 * pertJ + (pertJ > 0 ? 1000 : 0);
    };
    std::stable_sort(pl->sci.begin(), pl->sci.end(), [&](const nbnxn_sci_t& a, const nbnxn_sci_t& b) { return cost(a) > cost(b); });
    return pl;
}
} // namespace

extern "C" {

NbnxmHostPairlist* nbnxm_host_pairlist_build(const NbnxmHostGrid* g, const int* exclIndex,
                                             const int* exclAtoms, float rlist,
                                             int maxCjPackedPerSci, int carveFep, float rlistFep)
{
    return pairlistBuild(g, 0, exclIndex, exclAtoms, rlist, maxCjPackedPerSci, carveFep, rlistFep);
}

NbnxmHostPairlist* nbnxm_host_pairlist_build_dd(const NbnxmHostGrid* g, int nonLocal, const int* exclIndex, const int* exclAtoms,
                                                float rlist, int maxCjPackedPerSci)
{
    if (nonLocal && g->zones.size() < 2)
    {
        auto*        pl = new NbnxmHostPairlist; // no halo: an empty list (with the shared all-ones exclusion entry)
        nbnxn_excl_t allOnes;
        for (unsigned int& p : allOnes.pair) { p = 0xffffffffU; }
        pl->excl.push_back(allOnes);
        pl->jindex.push_back(0);
        return pl;
    }
    return pairlistBuild(g, nonLocal ? 1 : 0, nonLocal ? nullptr : exclIndex, nonLocal ? nullptr : exclAtoms, rlist, maxCjPackedPerSci, 0,
                         rlist);
}

void nbnxm_host_pairlist_free(NbnxmHostPairlist* pl)
{
    delete pl;
}

void nbnxm_host_pairlist_sizes(const NbnxmHostPairlist* pl, long long* sizes)
{
    sizes[0] = static_cast<long long>(pl->sci.size());
    sizes[1] = static_cast<long long>(pl->cjPacked.size());
    sizes[2] = static_cast<long long>(pl->excl.size());
    sizes[3] = static_cast<long long>(pl->iinr.size());
    sizes[4] = static_cast<long long>(pl->jjnr.size());
    sizes[5] = pl->numClusterPairs;
}

void nbnxm_host_pairlist_get(const NbnxmHostPairlist* pl, nbnxn_sci_t* sci,
                             nbnxn_cj_packed_t* cjPacked, nbnxn_excl_t* excl)
{
    if (sci) { std::copy(pl->sci.begin(), pl->sci.end(), sci); }
    if (cjPacked) { std::copy(pl->cjPacked.begin(), pl->cjPacked.end(), cjPacked); }
    if (excl) { std::copy(pl->excl.begin(), pl->excl.end(), excl); }
}

void nbnxm_host_pairlist_get_fep(const NbnxmHostPairlist* pl, int* iinr, int* shift, int* jindex,
                                 int* jjnr, int* excl_fep)
{
    if (iinr) { std::copy(pl->iinr.begin(), pl->iinr.end(), iinr); }
    if (shift) { std::copy(pl->shift.begin(), pl->shift.end(), shift); }
    if (jindex) { std::copy(pl->jindex.begin(), pl->jindex.end(), jindex); }
    if (jjnr) { std::copy(pl->jjnr.begin(), pl->jjnr.end(), jjnr); }
    if (excl_fep) { std::copy(pl->exclFep.begin(), pl->exclFep.end(), excl_fep); }
}

} // extern "C"
