/*
 * Work partition of the cluster-pair kernel (MI355X extension, no counterpart in the reference, whose
 * balancing is the host-side sci splitting of nbnxm/pairlist.cpp:2283-2400 against gpu_min_ci_balanced).
 *
 * The list is cut into as many contiguous ranges of packed j-groups as the device has resident wave slots,
 * all of the same weight (an instruction-count model of the kernel's loop levels, see nbnxmWorkWeightKernel).  In
 * fused mode the same pass marks the cluster pairs that touch a perturbed atom (they are left to
 * nbnxmFepClusterKernel) and lists them.  Three small kernels, run on the list's own
 * stream after every (re)prune, because pruning changes the masks on the device:
 *   nbnxmWorkWeightKernel  one thread per group: weight + per-256-group sums (+ fused mode: the group's mask of
 *                          perturbed cluster pairs, staged by the cluster kernel together with the list words, and
 *                          the list of those pairs)
 *   nbnxmWorkScanKernel    one workgroup: exclusive scan of the sums
 *   nbnxmWorkRangesKernel  one thread per group: global prefix -> the range borders that fall on this group,
 *                          for both partitions (4 and 5 waves per SIMD), plus the i-entry each range starts in
 */
#ifndef NBNXM_WORK_PARTITION_H
#define NBNXM_WORK_PARTITION_H

#include "nbnxm_hip_types.h"

constexpr int c_workBlockSize = 256;
#ifndef NBNXM_WEIGHT_GROUP
#define NBNXM_WEIGHT_SLOT 4
#define NBNXM_WEIGHT_GROUP 16
#define NBNXM_WEIGHT_ENTRY 128
#endif
/* slot / group / i-entry start, relative to 8 per cluster pair.  Round 1: 2 / 30 / 76; round 2 (lane-swap reduction of the i-forces): 0 / 34 / 60;
 * round 3 (tools/calibrate_weights.py on the kernel of profiles/r03: the pair block and the slot got cheaper — no clamp, no address arithmetic per
 * slot —, which makes an entry start, with its i-atom loads and the transposed force write, relatively dearer): the fit says 4.4 / 15.4 / 127,
 * and 4 / 16 / 128 measured 56.1 us against 56.9 - 57.9 us for 0 / 34 / 60 (2 / 24 / 96: 56.2; 0 / 34 / 120: 56.4 - 56.8) */
constexpr int c_weightPair    = 8;
constexpr int c_weightSlot    = NBNXM_WEIGHT_SLOT;
constexpr int c_weightGroup   = NBNXM_WEIGHT_GROUP;
#ifndef NBNXM_WEIGHT_EMPTY_GROUP
#define NBNXM_WEIGHT_EMPTY_GROUP NBNXM_WEIGHT_GROUP
#endif
constexpr int c_weightEmptyGroup = NBNXM_WEIGHT_EMPTY_GROUP;
constexpr int c_weightEntry   = NBNXM_WEIGHT_ENTRY;
/* short lists (updateWorkPartition): group and i-entry start weigh more where a wave has only a few of either */
constexpr int c_weightGroupShortList = 48;
constexpr int c_weightEntryShortList = 200;

/* the cost model as a kernel argument: the host picks it by the kind of list (updateWorkPartition) */
struct NbWorkWeights
{
    int pair, slot, group, emptyGroup, entry;
};

/* largest k with sciSorted[k].cjPackedBegin <= group (entries ordered by (cjPackedBegin, cjPackedEnd)); -1 if none */
__device__ __forceinline__ int findSciOfGroup(const nbnxn_sci_t* __restrict__ sciSorted, int nsci, int group)
{
    int lo = 0, hi = nsci; /* first k with begin > group */
    while (lo < hi)
    {
        const int mid = (lo + hi) >> 1;
        if (sciSorted[mid].cjPackedBegin <= group) { lo = mid + 1; }
        else { hi = mid; }
    }
    return lo - 1;
}

/* inclusive scan over the workgroup; returns this thread's inclusive prefix, total in *blockTotal */
__device__ __forceinline__ int blockInclusiveScan(int v, int* lds, int* blockTotal)
{
    const int t = static_cast<int>(threadIdx.x);
    lds[t]      = v;
    __syncthreads();
    for (int d = 1; d < c_workBlockSize; d <<= 1)
    {
        const int add = (t >= d) ? lds[t - d] : 0;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    const int incl = lds[t];
    *blockTotal    = lds[c_workBlockSize - 1];
    __syncthreads();
    return incl;
}

__launch_bounds__(c_workBlockSize) __global__
        void nbnxmWorkWeightKernel(const nbnxn_cj_packed_t* __restrict__ cjPacked,
                                   const int                             ncjPacked,
                                   const nbnxn_sci_t* __restrict__       sciSorted,
                                   const int                             nsci,
                                   const unsigned char* __restrict__     fepBits, /* nullptr: not the fused mode */
                                   const int                             buildSlowList, /* 0: groupSlowMask is up to date; 1: build the mask and append the
                                                                                         * HEAVY slow pairs (c_slowPairHeavy); 2: append the others
                                                                                         * (a second launch: the heavy ones come first in the list) */
                                   const unsigned* __restrict__          outerMask, /* gpu_plist::imask of a list that has been pruned, else nullptr */
                                   unsigned* __restrict__                groupSlowMask,
                                   NbSlowPair* __restrict__              slowPairs,   /* every listed slow pair with what its wave needs to start */
                                   const int                             slowCapacity,
                                   int* __restrict__                     slowCount,
                                   int* __restrict__                     groupWeight,
                                   int* __restrict__                     blockSum,
                                   const NbWorkWeights                   weights)
{
    __shared__ int lds[c_workBlockSize];
    const int      g = static_cast<int>(blockIdx.x) * c_workBlockSize + static_cast<int>(threadIdx.x);
    int            w = 0;
    if (g < ncjPacked)
    {
        const unsigned imask = cjPacked[g].imei[0].imask;
        const int      k     = findSciOfGroup(sciSorted, nsci, g);
        const bool     owned = (k >= 0 && g < sciSorted[k].cjPackedEnd);
        /* fused mode: the cluster pairs of this group that touch a perturbed atom (Grid::fepBits): every i-cluster when
         * the j-cluster holds one, otherwise the i-clusters that hold one.  Independent of the distance pruning, so it is
         * computed once per list; the cluster kernel masks them out, nbnxmFepClusterKernel evaluates them. */
        unsigned slow = 0U;
        if (fepBits != nullptr && !buildSlowList) { slow = groupSlowMask[g]; }
        if (fepBits != nullptr && buildSlowList && owned)
        {
            const int sci          = sciSorted[k].sci;
            unsigned  iClusterMask = 0U;
            for (int i = 0; i < c_numClPerSupercl; i++)
            {
                if (fepBits[sci * c_numClPerSupercl + i] != 0) { iClusterMask |= (1U << i); }
            }
            for (int jm = 0; jm < c_jGroupSize; jm++)
            {
                const unsigned m = (fepBits[cjPacked[g].cj[jm]] != 0) ? 0xFFU : iClusterMask;
                slow |= m << (jm * c_numClPerSupercl);
            }
            /* the pairs have to be a superset of what any later pruning leaves: the masks of the fresh list when the
             * launcher builds this before the first prune; once the list has been pruned (a mode or fepBits change on a live
             * list), the outer-pruned masks the rolling pass re-adds pairs from — the working mask is checked at run time */
            unsigned  todo = slow & (outerMask != nullptr ? outerMask[g * NBNXM_GPU_CLUSTERPAIR_SPLIT] : imask);
            /* The list starts with the cluster pairs that hold MANY perturbed atom pairs (both clusters carry perturbed atoms: a ligand's
             * pairs with itself): on a dH/dlambda step such a pair is the longest chain of the kernel's trailing work (20 us against 7),
             * and the waves take the list in order — the long ones must not be the last to start. */
            {
                unsigned heavy = 0U;
                for (unsigned rest = todo; rest != 0U; rest &= rest - 1U)
                {
                    const int bit = __ffs(rest) - 1;
                    const int pi  = __popc(static_cast<unsigned>(fepBits[sci * c_numClPerSupercl + (bit & 7)]));
                    const int pj  = __popc(static_cast<unsigned>(fepBits[cjPacked[g].cj[bit >> 3]]));
                    if (c_clSize * c_clSize - (c_clSize - pi) * (c_clSize - pj) > c_slowPairHeavy) { heavy |= 1U << bit; }
                }
                todo = (buildSlowList == 1) ? heavy : (todo & ~heavy);
            }
            const int n    = __popc(todo);
            if (n > 0)
            {
                int idx = atomicAdd(slowCount, n);
                while (todo != 0U && idx < slowCapacity)
                {
                    const int bit    = __ffs(todo) - 1;
                    NbSlowPair rec;
                    rec.entry      = g * 32 + bit;
                    rec.sciShift   = sciSorted[k].sci * 64 + (sciSorted[k].shift & NBNXM_CI_SHIFT_MASK);
                    rec.cj         = cjPacked[g].cj[bit >> 3];
                    rec.exclInd[0] = cjPacked[g].imei[0].excl_ind;
                    rec.exclInd[1] = cjPacked[g].imei[1].excl_ind;
                    rec.pad[0] = rec.pad[1] = rec.pad[2] = 0;
                    slowPairs[idx] = rec;
                    todo &= todo - 1U;
                    idx++;
                }
            }
        }
        if (buildSlowList == 1) { groupSlowMask[g] = slow; }
        /* cost model in units of 1/8 cluster pair (fitted to per-SIMD finish times, tools/calibrate_weights.py): a
         * cluster pair, a non-empty j-cluster slot (staged reads, j-force reduction, atomic), a group (staging loads,
         * waits), and the start of an i-entry (i-atom loads, i-force reduction and atomics) */
        const unsigned fast  = imask & ~slow;
        int            slots = 0;
        for (int jm = 0; jm < c_jGroupSize; jm++) { slots += ((fast >> (jm * c_numClPerSupercl)) & 0xFFU) != 0U ? 1 : 0; }
        /* a group with nothing left for this kernel (pruned away, or all of it perturbed) still costs the wave its pipeline step: a run of
         * them at weight 0 ends up in ONE range (3k-atom box with 48 perturbed atoms: 47 groups in one range, kernel 23.6 instead of 12 us) */
        w = weights.pair * __popc(fast) + weights.slot * slots + (fast != 0U ? weights.group : weights.emptyGroup);
        if (owned && sciSorted[k].cjPackedBegin == g) { w += weights.entry; }
        groupWeight[g] = w;
    }
    int total;
    (void)blockInclusiveScan(w, lds, &total);
    if (threadIdx.x == 0) { blockSum[blockIdx.x] = total; }
}

/* in place: blockSum[b] <- sum of blockSum[0..b), blockSum[numBlocks] <- total */
__launch_bounds__(c_workBlockSize) __global__ void nbnxmWorkScanKernel(int* __restrict__ blockSum, const int numBlocks)
{
    __shared__ int lds[c_workBlockSize];
    int            carry = 0;
    for (int base = 0; base < numBlocks; base += c_workBlockSize)
    {
        const int idx = base + static_cast<int>(threadIdx.x);
        const int v   = (idx < numBlocks) ? blockSum[idx] : 0;
        int       total;
        const int incl = blockInclusiveScan(v, lds, &total);
        if (idx < numBlocks) { blockSum[idx] = carry + incl - v; }
        carry += total;
    }
    if (threadIdx.x == 0) { blockSum[numBlocks] = carry; }
}

struct WorkPartitionOut
{
    int  numRanges;
    int* rangeStart; /* numRanges + 1 */
    int* firstSci;   /* numRanges */
    /* Share of the total weight each range gets, as a running sum: range a owns the weight fractions
     * [shareCum[a], shareCum[a + 1]), shareCum[0] = 0, shareCum[numRanges] = 1.  nullptr: equal shares.
     * Unequal shares make up for what the weights cannot know: the arbiter of a SIMD favours its older waves (with equal
     * work they finish one after the other and the youngest runs alone at the end), and some SIMDs are persistently
     * slower than others (measured: the same ones on every launch). */
    const float* shareCum;
};

/* the range that owns a group with weight e in front of it, out of total */
__device__ __forceinline__ int rangeOfWeight(const WorkPartitionOut& out, long long e, long long total)
{
    const int nW = out.numRanges;
    if (out.shareCum == nullptr) { return static_cast<int>(min(static_cast<long long>(nW - 1), e * nW / total)); }
    const float phi = static_cast<float>(static_cast<double>(e) / static_cast<double>(total));
    int         lo = 0, hi = nW; /* largest a with shareCum[a] <= phi */
    while (hi - lo > 1)
    {
        const int mid = (lo + hi) >> 1;
        if (out.shareCum[mid] <= phi) { lo = mid; }
        else { hi = mid; }
    }
    return lo;
}

__launch_bounds__(c_workBlockSize) __global__
        void nbnxmWorkRangesKernel(const int* __restrict__         groupWeight,
                                   const int* __restrict__         blockPrefix,
                                   const int                       ncjPacked,
                                   const int                       numBlocks,
                                   const nbnxn_sci_t* __restrict__ sciSorted,
                                   const int                       nsci,
                                   const WorkPartitionOut          out0,
                                   const WorkPartitionOut          out1,
                                   const WorkPartitionOut          out2)
{
    __shared__ int lds[c_workBlockSize];
    const int      g = static_cast<int>(blockIdx.x) * c_workBlockSize + static_cast<int>(threadIdx.x);
    const int      w = (g < ncjPacked) ? groupWeight[g] : 0;
    int            blockTotal;
    const int      incl = blockInclusiveScan(w, lds, &blockTotal);
    if (g >= ncjPacked) { return; }
    const long long total = max(1, blockPrefix[numBlocks]);
    const long long e     = blockPrefix[blockIdx.x] + incl - w;         /* weight before this group */
    const long long ePrev = (g == 0) ? -1 : e - groupWeight[g - 1];     /* ... and before the previous one */
    int             sciOfGroup = -2;
#pragma unroll
    for (int p = 0; p < c_numWorkPartitions; p++)
    {
        const WorkPartitionOut& out = (p == 0) ? out0 : ((p == 1) ? out1 : out2);
        const int               nW  = out.numRanges;
        if (nW <= 0) { continue; }
        /* range a owns the groups whose preceding weight lies in its share of the total (equal shares: [a, a + 1) total / nW) */
        const int a  = rangeOfWeight(out, e, total);
        const int lo = (g == 0) ? 0 : rangeOfWeight(out, ePrev, total) + 1;
        if (lo <= a && sciOfGroup == -2) { sciOfGroup = max(0, findSciOfGroup(sciSorted, nsci, g)); }
        for (int r = lo; r <= a; r++)
        {
            out.rangeStart[r] = g;
            out.firstSci[r]   = sciOfGroup;
        }
        if (g == ncjPacked - 1)
        {
            for (int r = a + 1; r <= nW; r++)
            {
                out.rangeStart[r] = ncjPacked; /* empty ranges behind the last group, and the end sentinel */
                if (r < nW) { out.firstSci[r] = nsci; }
            }
        }
    }
}

/* One thread per range: the range's start record (NbWorkDesc, nbnxm_hip_types.h) from the borders the kernel above has written. */
__launch_bounds__(c_workBlockSize) __global__
        void nbnxmWorkDescKernel(const int* __restrict__               rangeStart,
                                 const int* __restrict__               firstSci,
                                 const int                             numRanges,
                                 const nbnxn_sci_t* __restrict__       sciSorted,
                                 const int                             nsci,
                                 const nbnxn_cj_packed_t* __restrict__ cjPacked,
                                 NbWorkDesc* __restrict__              desc)
{
    const int r = static_cast<int>(blockIdx.x) * c_workBlockSize + static_cast<int>(threadIdx.x);
    if (r >= numRanges) { return; }
    const int  b = rangeStart[r], e = rangeStart[r + 1];
    NbWorkDesc d;
    d.rangeBegin = d.rangeEnd = b; /* empty unless an i-entry owns a group of [b, e) */
    d.sciIdx     = nsci;
    d.firstGroup = b;
    d.entry      = nbnxn_sci_t{ 0, 0, 0, 0 };
    d.cj[0] = d.cj[1] = d.cj[2] = d.cj[3] = 0;
    d.exclInd[0] = d.exclInd[1] = 0;
    d.pad[0] = d.pad[1] = 0;
    for (int k = firstSci[r]; b < e && k < nsci; k++)
    {
        const nbnxn_sci_t s = sciSorted[k];
        if (s.cjPackedBegin >= e) { break; }
        const int pb = max(b, s.cjPackedBegin), pe = min(e, s.cjPackedEnd);
        if (pb < pe)
        {
            d.rangeEnd   = e;
            d.sciIdx     = k;
            d.firstGroup = pb;
            d.entry      = s;
            for (int jm = 0; jm < c_jGroupSize; jm++) { d.cj[jm] = cjPacked[pb].cj[jm]; }
            d.exclInd[0] = cjPacked[pb].imei[0].excl_ind;
            d.exclInd[1] = cjPacked[pb].imei[1].excl_ind;
            break;
        }
    }
    desc[r] = d;
}

#endif
