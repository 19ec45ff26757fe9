/*
 * Device-side building blocks shared by the cluster-pair kernel, the atom-pair FEP kernels and
 * the foreign-lambda kernel (gfx950, wave64).
 *
 * What is computed follows the reference (citations at each function); how it is computed is
 * CDNA4-specific: DPP row operations for the 8-lane j-force reduction, 64-lane butterflies,
 * segmented wave reductions, a locally fitted rational Ewald correction.
 */
#ifndef NBNXM_DEVICE_HELPERS_H
#define NBNXM_DEVICE_HELPERS_H

#include <hip/hip_runtime.h>

#include "nbnxm_hip_types.h"
#include "pme_corr_coeffs.h"

/* kernel flavour tags (compile-time) */
enum
{
    ELK_CUT = 0,
    ELK_RF,
    ELK_EWALD_ANA,
    ELK_EWALD_TAB
};
enum
{
    VDK_CUT = 0, /* nbfp table */
    VDK_COMB_GEOM,
    VDK_COMB_LB,
    VDK_FSWITCH,
    VDK_PSWITCH,
    VDK_EWALD_GEOM, /* nbfp table + LJ-PME grid correction, geometric combination of the grid C6 (nbfp_comb) */
    VDK_EWALD_LB    /* ... Lorentz-Berthelot combination */
};

#define NB_DEVINL __device__ __forceinline__

constexpr float c_oneSixth    = 1.0F / 6.0F;
constexpr float c_oneTwelfth  = 1.0F / 12.0F;
constexpr float c_oneOverSqrtPi = 0.564189583547756F;

/* ---- cross-lane ------------------------------------------------------------------------------ */

/* DPP controls (LLVM AMDGPU): quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
 * row_ror:8 = 0x128 */
template<int CTRL>
NB_DEVINL float dppMove(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

/* Sum over the 8 consecutive lanes that share a j atom (same lane >> 3); every lane gets the sum. */
NB_DEVINL float reduceOver8Lanes(float v)
{
    v += dppMove<0xB1>(v);  /* lane ^ 1 */
    v += dppMove<0x4E>(v);  /* lane ^ 2 */
    v += dppMove<0x141>(v); /* i <-> 7 - i within each half row */
    return v;
}

/* x, y and z summed over the 8 consecutive lanes that share a j atom, delivered where the j-force atomic wants them: lane
 * (l & 7) == 0 gets sum x, 1 gets sum y, 2 gets sum z (the other lanes hold partial sums).  Transposing while reducing takes
 * 4 DPP adds + 4 selects; three separate reductions take 9 DPP adds and a select chain. */
NB_DEVINL float reduceXyzOver8Lanes(const float3 v, const unsigned lane)
{
    const bool odd  = (lane & 1U) != 0U;
    const bool high = (lane & 2U) != 0U;
    float      keep = odd ? v.y : v.x;
    const float give = odd ? v.x : v.y;
    keep += dppMove<0xB1>(give); /* lane ^ 1: even lanes x + x', odd lanes y + y' */
    const float zz    = v.z + dppMove<0xB1>(v.z);
    const float give2 = high ? keep : zz;
    float       keep2 = high ? zz : keep;
    keep2 += dppMove<0x4E>(give2);  /* lane ^ 2: lanes 0, 1 of a quad: x, y over the quad; lanes 2, 3: z over the quad */
    keep2 += dppMove<0x104>(keep2); /* row_shl:4: lane l += lane l + 4 (zero beyond the row) */
    return keep2;
}

/* Sum over the 8 lanes with the same (lane & 7), i.e. over tidxj; every lane gets the sum. */
NB_DEVINL float reduceOverTidxj(float v)
{
    v += dppMove<0x128>(v); /* lane ^ 8 (rotate by 8 within the 16-lane row) */
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

/* gfx950 lane-swap instructions (no selects, no LDS crossbar): v_permlane32_swap exchanges lanes 32-63 of the first register
 * with lanes 0-31 of the second; v_permlane16_swap exchanges the odd 16-lane rows of the first with the even rows of the second.
 * After the swap, a + b holds the first register's sum over the swapped lane bit in the lanes where that bit is 0 and the second
 * register's in the lanes where it is 1.  Inline asm (operands read and written); the s_nop keep the neighbouring VALU writes at
 * a safe distance, which the compiler's hazard recogniser cannot do for asm. */
NB_DEVINL float sumPairOverLaneBit5(float a, float b)
{
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
NB_DEVINL float sumPairOverLaneBit4(float a, float b)
{
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

/* The 8 per-i-cluster accumulators of one force component, summed over tidxj (lane bits 3-5): lane (tidxj, tidxi) gets the sum
 * of cluster tidxj.  A transposing reduction: 8 -> 4 -> 2 -> 1 registers with 6 lane swaps, 7 adds and 2 selects; reducing each
 * accumulator by itself takes 8 x (1 DPP + 2 LDS-crossbar shuffles) and a select chain. */
NB_DEVINL float reduceOverTidxjTransposed(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, const unsigned lane)
{
    const float r0 = sumPairOverLaneBit5(a0, a4); /* lanes 0-31: cluster 0, lanes 32-63: cluster 4 */
    const float r1 = sumPairOverLaneBit5(a1, a5);
    const float r2 = sumPairOverLaneBit5(a2, a6);
    const float r3 = sumPairOverLaneBit5(a3, a7);
    const float t0 = sumPairOverLaneBit4(r0, r2); /* lane bit 4 = 0: clusters 0 | 4, = 1: clusters 2 | 6 */
    const float t1 = sumPairOverLaneBit4(r1, r3);
    const bool  b3   = (lane & 8U) != 0U;
    float       keep = b3 ? t1 : t0;
    const float give = b3 ? t0 : t1;
    keep += dppMove<0x128>(give); /* lane ^ 8 */
    return keep;
}

NB_DEVINL float waveSum(float v)
{
    v += dppMove<0xB1>(v);
    v += dppMove<0x4E>(v);
    v += dppMove<0x141>(v);
    v += dppMove<0x128>(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

/* Four wave sums at the price of one and a half: a, b, c, d are transposed while they are reduced (as reduceXyzOver8Lanes does for
 * three), so that lane l ends up with the sum of a (l & 3 == 0), b (1), c (2) or d (3) over all 64 lanes.  7 adds (5 DPP, 2 through the
 * LDS crossbar) and 6 selects; four waveSum calls take 24 adds, 8 of them through the crossbar. */
NB_DEVINL float waveSum4Transposed(const float a, const float b, const float c, const float d, const unsigned lane)
{
    const bool bit0 = (lane & 1U) != 0U;
    const bool bit1 = (lane & 2U) != 0U;
    float       ab     = bit0 ? b : a;
    const float abGive = bit0 ? a : b;
    ab += dppMove<0xB1>(abGive); /* lane ^ 1: even lanes a + a', odd lanes b + b' */
    float       cd     = bit0 ? d : c;
    const float cdGive = bit0 ? c : d;
    cd += dppMove<0xB1>(cdGive); /* even lanes c + c', odd lanes d + d' */
    float       v    = bit1 ? cd : ab;
    const float give = bit1 ? ab : cd;
    v += dppMove<0x4E>(give);  /* lane ^ 2: lane & 3 == 0: a over the quad, 1: b, 2: c, 3: d */
    v += dppMove<0x124>(v);    /* row_ror:4 and row_ror:8 keep lane & 3: the four quads of a 16-lane row */
    v += dppMove<0x128>(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

/* ---- math ------------------------------------------------------------------------------------ */

/* [d/dz (erf z / z)] / z as a function of z^2; definition as gmx::pmeForceCorrection
 * (simd/simd_math.h:1560-1650); own [5/4] rational fit on z^2 <= PME_CORR_XMAX (tools/fit_pme_corr.py,
 * relative error < 1e-6, the accuracy class of the reference's approximation). */
NB_DEVINL float pmeCorrF(float z2)
{
    float num = PME_CORR_P5;
    num       = fmaf(num, z2, PME_CORR_P4);
    num       = fmaf(num, z2, PME_CORR_P3);
    num       = fmaf(num, z2, PME_CORR_P2);
    num       = fmaf(num, z2, PME_CORR_P1);
    num       = fmaf(num, z2, PME_CORR_P0);
    float den = PME_CORR_Q4;
    den       = fmaf(den, z2, PME_CORR_Q3);
    den       = fmaf(den, z2, PME_CORR_Q2);
    den       = fmaf(den, z2, PME_CORR_Q1);
    den       = fmaf(den, z2, PME_CORR_Q0);
    return num * __builtin_amdgcn_rcpf(den);
}

/* erf(z)/z as a function of z^2 (definition as gmx::pmePotentialCorrection, simd/simd_math.h:1660-1760); own
 * [5/4] rational fit on z^2 <= PME_CORR_XMAX (tools/fit_pme_corr.py, relative error < 5e-7 in fp32). */
NB_DEVINL float pmeCorrV(float z2)
{
    float num = PME_CORR_V_P5;
    num       = fmaf(num, z2, PME_CORR_V_P4);
    num       = fmaf(num, z2, PME_CORR_V_P3);
    num       = fmaf(num, z2, PME_CORR_V_P2);
    num       = fmaf(num, z2, PME_CORR_V_P1);
    num       = fmaf(num, z2, PME_CORR_V_P0);
    float den = PME_CORR_V_Q4;
    den       = fmaf(den, z2, PME_CORR_V_Q3);
    den       = fmaf(den, z2, PME_CORR_V_Q2);
    den       = fmaf(den, z2, PME_CORR_V_Q1);
    den       = fmaf(den, z2, PME_CORR_V_Q0);
    return num * __builtin_amdgcn_rcpf(den);
}

/* Linear interpolation in the Ewald force table (nbnxm_cuda_kernel_utils.cuh:448-459) */
NB_DEVINL float interpolateCoulombForceR(const NBParamGpu& nbp, float r)
{
    const float normalized = nbp.coulomb_tab_scale * r;
    const int   index      = static_cast<int>(normalized);
    const float fraction   = normalized - static_cast<float>(index);
    const float d0         = nbp.coulomb_tab[index];
    const float d1         = nbp.coulomb_tab[index + 1];
    return fmaf(fraction, d1, fmaf(-fraction, d0, d0));
}

/* LJ-PME: C6 of the grid part from the per-type parameters NBParamGpu::nbfp_comb (nbnxm_cuda_kernel_utils.cuh:221-229, 283-297) */
NB_DEVINL float ljGridC6(int vdwKind, const float2& a, const float2& b)
{
    if (vdwKind == VDK_EWALD_GEOM) { return a.x * b.x; }
    const float sigma  = a.x + b.x;
    const float sigma2 = sigma * sigma;
    return a.y * b.y * sigma2 * sigma2 * sigma2;
}

/* the perturbed-pair kernels have no VdW template parameter: grid C6 of the A and B states of a pair, zero without LJ-PME */
NB_DEVINL bool isLjPme(const NBParamGpu& nbp)
{
    return nbp.vdwType == NBNXM_VDW_EWALD_GEOM || nbp.vdwType == NBNXM_VDW_EWALD_LB;
}
NB_DEVINL void ljGridC6AB(const NBParamGpu& nbp, const int4& t4i, const int4& t4j, float (&c6grid)[2])
{
    c6grid[0] = c6grid[1] = 0.0F;
    if (isLjPme(nbp))
    {
        const int kind = (nbp.vdwType == NBNXM_VDW_EWALD_GEOM) ? VDK_EWALD_GEOM : VDK_EWALD_LB;
        c6grid[0]      = ljGridC6(kind, nbp.nbfp_comb[t4i.x], nbp.nbfp_comb[t4j.x]);
        c6grid[1]      = ljGridC6(kind, nbp.nbfp_comb[t4i.y], nbp.nbfp_comb[t4j.y]);
    }
}

/* Byte address of the Ewald correction table's entry for r^2 (entries of STRIDE bytes from the table's base; NBParamGpu::ewaldCorrTab):
 * y = r^2 * (STRIDE * entries per unit of r^2) + 2^23 is a float with an integer ulp, so its low mantissa bits ARE round(r^2 * ...):
 * one FMA and one AND (which drops the bits below the stride) give the entry's address — no separate multiply, no convert, no
 * shift.  Entry k is therefore the line over [(k - 1/16) h, (k + 15/16) h] for STRIDE 8 (uploadEwaldCorrectionTable), from (k - 1/32) h
 * for the 16-byte entries of the energy flavours (four bits dropped). */
template<int STRIDE>
NB_DEVINL unsigned ewaldTabAddress(const float r2, const float scaleTimesStride)
{
    /* (the scale arrives in a vector register, nbnxmKernel sets it once: an FMA with a scalar operand issues at half rate, with a literal
     * addend — v_fmaak — at full rate: tools/ubench/valu_rate3.hip, 1.90 against 1.18 ns per wave instruction and SIMD.  Measured and not
     * kept: the AND's mask from an SGPR instead of a literal, +1.1 us) */
    const float y = fmaf(r2, scaleTimesStride, 8388608.0F);
    /* (the 16-byte entries are the energy flavours' table: c_ewaldCorrTabSizeEnergy entries, masked with the power of two above it) */
    constexpr int c_maskEntries = (STRIDE == 16) ? 2048 : c_ewaldCorrTabSize;
    static_assert(c_ewaldCorrTabSizeEnergy <= 2048, "the energy table's address mask");
    return __builtin_bit_cast(unsigned, y) & static_cast<unsigned>((c_maskEntries * STRIDE - 1) & ~(STRIDE - 1));
}

/* ---- non-perturbed atom pair (nbnxm_cuda_kernel.cuh:518-645) ---------------------------------- */

/* Energy steps of the headline flavours (reaction field or analytical Ewald, LJ cut-off from the type table or a combination rule,
 * one cut-off): nbPairEnergy below. */
template<int ELEC, bool TWIN, int VDW, bool ENERGY>
constexpr bool c_energyHeadlineBlock = ENERGY && !TWIN && (VDW == VDK_CUT || VDW == VDK_COMB_GEOM || VDW == VDK_COMB_LB)
                                       && (ELEC == ELK_RF || ELEC == ELK_EWALD_ANA);

/* The sums of an energy step that nbPairEnergy feeds; NbEnergySums::finish turns them into E_lj and E_el once per wave (or piece). */
struct NbEnergySums
{
    float ljTimes12 = 0.0F; /* sum of 12 E_lj without the potential shift: c12 r^-12 - 2 c6 r^-6 (c6 = 6 C6, c12 = 12 C12) */
    float c12Masked = 0.0F; /* sum of c12 over the interacting pairs within the cut-off: times cpot_rep / 12 = the repulsion shift */
    float c6Masked  = 0.0F; /* ... of c6: times -cpot_disp / 6 = the dispersion shift */
    float el        = 0.0F; /* E_el without the Ewald potential shift */
    float qqMasked  = 0.0F; /* sum of q q over the interacting pairs within the cut-off: times -sh_ewald */
};

/* One pair of an energy step, with the sums it feeds (nbnxm_cuda_kernel.cuh:533-536,602-607,633-640 are the terms).  43 -> 38 vector
 * instructions per executed pair block against the form that returned F/r, E_lj and E_el for the caller to add up:
 *   - the exclusion bit masks 1/r ONCE (and the constant 1.0): r^-2, r^-6 and everything built from them is then zero for an excluded
 *     pair, where there were three masked results;
 *   - E_lj is summed as 12 E = c12 r^-12 - 2 c6 r^-6 = r^2 (F_lj / r) - c6 r^-6: the force's own product plus one FMA; the two
 *     potential shifts leave the pair — the sums of c12 and of c6 over the interacting pairs, one FMA each, times their constants
 *     once per wave (before: both shifts and both 1/12, 1/6 factors per pair, six instructions);
 *   - the Ewald potential shift leaves the pair the same way: sum of q q (1/r masked - beta V) here, the sum of q q over the interacting
 *     pairs times sh_ewald once per wave;
 *   - no r^2 clamp: the kernel's sum of squares starts from c_r2Floor, so 1/r is finite, and an excluded pair's 1/r is masked.
 * (With 16-byte table entries {c6, c12, 12 sh, 2 c6} the block is one instruction shorter still; not done: it doubles the LJ table in LDS,
 * which is what bounds the number of atom types at full occupancy.) */
template<int ELEC>
NB_DEVINL void nbPairEnergy(const NBParamGpu& nbp, const float r2, const int intMask, const float qq, const float c6, const float c12,
                            const float ewaldTabScaleV, float& F_invr, NbEnergySums& sums)
{
    [[maybe_unused]] float4 t = make_float4(0.0F, 0.0F, 0.0F, 0.0F);
    if constexpr (ELEC == ELK_EWALD_ANA)
    {
        /* {intercept, slope} of beta^3 F and of beta V in r^2, one ds_read_b128; the table sits at LDS address 0 */
        typedef __attribute__((address_space(3))) const float LdsFloat;
        LdsFloat* tab = reinterpret_cast<LdsFloat*>(static_cast<uintptr_t>(ewaldTabAddress<16>(r2, ewaldTabScaleV)));
        t.x           = tab[0];
        t.y           = tab[1];
        t.z           = tab[2];
        t.w           = tab[3];
    }
    float inv_r   = __frsqrt_rn(r2);
    float int_bit = 1.0F;
    asm("v_and_b32 %0, %1, %2" : "=v"(inv_r) : "v"(intMask), "v"(inv_r));
    asm("v_and_b32 %0, %1, %2" : "=v"(int_bit) : "v"(intMask), "v"(int_bit));
    const float inv_r2 = inv_r * inv_r;
    const float inv_r6 = inv_r2 * inv_r2 * inv_r2;
    const float lj     = fmaf(c12, inv_r6, -c6) * inv_r6; /* r^2 (F_lj / r) = c12 r^-12 - c6 r^-6 */
    const float nm     = fmaf(qq, inv_r, lj) * inv_r2;
    sums.ljTimes12     = fmaf(-c6, inv_r6, sums.ljTimes12 + lj);
    sums.c12Masked     = fmaf(c12, int_bit, sums.c12Masked);
    sums.c6Masked      = fmaf(c6, int_bit, sums.c6Masked);
    if constexpr (ELEC == ELK_RF)
    {
        F_invr = fmaf(qq, -nbp.two_k_rf, nm);
        /* (the reaction-field constant belongs to excluded pairs too: not masked) */
        sums.el = fmaf(qq, inv_r + fmaf(0.5F * nbp.two_k_rf, r2, -nbp.c_rf), sums.el);
    }
    else
    {
        F_invr        = fmaf(qq, fmaf(t.y, r2, t.x), nm);
        sums.el       = fmaf(qq, inv_r - fmaf(t.w, r2, t.z), sums.el);
        sums.qqMasked = fmaf(qq, int_bit, sums.qqMasked);
    }
}

template<int ELEC, bool TWIN, int VDW, bool ENERGY, bool EXCL_FORCES, bool HAS_EXCL = true, bool CORR_TABLE = true>
NB_DEVINL void nbPair(const NBParamGpu& nbp,
                      const float2*     ewaldCorrLds, /* analytical Ewald: NBParamGpu::ewaldCorrTab in LDS */
                      float             r2,
                      int               intMask, /* all ones if the pair interacts, 0 if it is excluded */
                      float             qq, /* epsfac q_i q_j */
                      float             c6,
                      float             c12,
                      float&            F_invr,
                      float&            E_lj,
                      float&            E_el,
                      [[maybe_unused]] float c6grid = 0.0F, /* LJ-PME flavours: C6 of the grid part for this pair */
                      [[maybe_unused]] float ewaldTabScaleV = 0.0F /* cluster kernel, analytical Ewald: NBParamGpu::ewaldCorrTabScale8 (16 on
                                                                    * energy steps) in a VECTOR register, set once per kernel */)
{
    /* (Round 4, advisor finding: NON-excluded pairs at zero distance exist as well — the reference parks all filler atoms of its grid on
     * ONE point (atomdata.cpp:148-184) and lists filler-filler pairs, with zero charge and zero LJ parameters.  Without the clamp such a
     * pair is 0 x inf = NaN, which the AND with an all-ones mask keeps.  The cluster kernel therefore starts its sum of squares from
     * c_r2Floor = 1e-12 nm^2: below half an ulp of any r^2 > 2e-5 nm^2, so no real pair sees it, and at r = 0 it makes 1/r = 1e6,
     * r^-6 = 1e36 finite, times the fillers' zero parameters = 0.  tests/test_gpu_parity.py::test_filler_atoms_on_one_point.) */
    /* r^2 >= c_nbnxnMinDistanceSquared (pairlist.h:166) exists so that EXCLUDED pairs at zero distance — an atom's pair with itself,
     * a shell on its core — do not turn the sums into NaN.  The force-only one-mask block below removes everything an exclusion
     * removes with a bit-wise AND, which gives +0 whatever the masked value is (infinity and NaN included), its Ewald term is finite
     * at r = 0 (reaction field, analytical Ewald) and the force is that scalar times r = 0: the clamp — a half-rate v_max with a
     * literal — is not needed there
     * (96k box 57.7 -> 56.3 us, 1M atoms 0.467 -> 0.458 ms).  Two NON-excluded atoms within 0.6 pm overflow r^-12 in either form.
     * tests/test_gpu_parity.py::test_excluded_atoms_on_top_of_each_other.  -DNBNXM_KEEP_R2_CLAMP restores it everywhere. */
#ifndef NBNXM_KEEP_R2_CLAMP
    /* (not the r-indexed Ewald table, whose index is r^2 / r, and not LJ-PME, whose grid term sits outside the mask: both would be 0 x inf) */
    constexpr bool c_oneMaskForceBlock = !ENERGY && EXCL_FORCES && HAS_EXCL && CORR_TABLE && (ELEC == ELK_RF || ELEC == ELK_EWALD_ANA)
                                         && VDW != VDK_EWALD_GEOM && VDW != VDK_EWALD_LB;
#else
    constexpr bool c_oneMaskForceBlock = false;
#endif
    if constexpr (!c_oneMaskForceBlock) { r2 = fmaxf(r2, c_nbnxnMinDistanceSquared); }
    /* force-only analytical Ewald: the correction's table read goes out before anything else — its LDS round trip is the longest
     * latency of the block, and left to the scheduler it is issued behind the multiplies that wait for the reciprocal square root */
    [[maybe_unused]] float2 tEarly = make_float2(0.0F, 0.0F);
    constexpr bool          c_earlyTableRead = c_oneMaskForceBlock && ELEC == ELK_EWALD_ANA;
    if constexpr (c_earlyTableRead)
    {
        typedef __attribute__((address_space(3))) const float LdsFloatE;
        /* the table sits at LDS address 0 (nbnxmKernel checks it): no base to add */
        LdsFloatE* tab = reinterpret_cast<LdsFloatE*>(static_cast<uintptr_t>(ewaldTabAddress<8>(r2, ewaldTabScaleV)));
        tEarly.x       = tab[0];
        tEarly.y       = tab[1];
#ifndef NBNXM_NO_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    const float inv_r  = __frsqrt_rn(r2);
    const float inv_r2 = inv_r * inv_r;
    float       inv_r6 = inv_r2 * inv_r2 * inv_r2;
    /* The exclusion bit is applied as a bit mask on the float (x & ~0 = x, x & 0 = +0): one full-rate v_and
     * per masked term instead of the compare + select + multiply chain (half-rate ops on gfx950).
     * HAS_EXCL == false: the caller guarantees that every pair interacts, no masking at all. */
    auto masked = [intMask](float v) { return __builtin_bit_cast(float, __builtin_bit_cast(int, v) & intMask); };
    [[maybe_unused]] const float int_bit = HAS_EXCL ? masked(1.0F) : 1.0F;
    constexpr bool MASK_FORCES = EXCL_FORCES && HAS_EXCL;
    if constexpr (ENERGY && (VDW == VDK_CUT || VDW == VDK_COMB_GEOM || VDW == VDK_COMB_LB) && !TWIN && HAS_EXCL
                  && (ELEC == ELK_RF || (ELEC == ELK_EWALD_ANA && CORR_TABLE)))
    {
        /* Energy steps of the headline flavours: the force as in the block below, and each energy with ONE mask,
         *   E_lj = mask(c12 (r^-12 + cpot12) / 12 - c6 (r^-6 + cpot6) / 6),   E_el = q q (mask(1/r - shift) - beta V((beta r)^2)),
         * with the potential correction from the same table entry as the force correction (one ds_read_b128). */
        [[maybe_unused]] float4 t  = make_float4(0.0F, 0.0F, 0.0F, 0.0F);
        [[maybe_unused]] float  xs = 0.0F;
        if constexpr (ELEC == ELK_EWALD_ANA)
        {
            typedef __attribute__((address_space(3))) const float LdsFloat;
            LdsFloat* tab = reinterpret_cast<LdsFloat*>(static_cast<uintptr_t>(ewaldTabAddress<16>(r2, ewaldTabScaleV)));
            t.x           = tab[0];
            t.y           = tab[1];
            t.z           = tab[2];
            t.w           = tab[3];
            (void)ewaldCorrLds;
        }
        const float lj = fmaf(c12, inv_r6, -c6) * inv_r6;
        float       nm = fmaf(qq, inv_r, lj) * inv_r2;
        asm("v_and_b32 %0, %1, %2" : "=v"(nm) : "v"(intMask), "v"(nm));
        const float e12 = fmaf(inv_r6, inv_r6, nbp.repulsion_shift.cpot) * c12;
        const float e6  = (inv_r6 + nbp.dispersion_shift.cpot) * c6;
        float       elj = fmaf(e6, -c_oneSixth, e12 * c_oneTwelfth);
        asm("v_and_b32 %0, %1, %2" : "=v"(elj) : "v"(intMask), "v"(elj));
        E_lj = elj;
        if constexpr (ELEC == ELK_RF)
        {
            F_invr    = fmaf(qq, -nbp.two_k_rf, nm);
            float eel = inv_r;
            asm("v_and_b32 %0, %1, %2" : "=v"(eel) : "v"(intMask), "v"(eel));
            E_el = qq * (eel + fmaf(0.5F * nbp.two_k_rf, r2, -nbp.c_rf));
        }
        else
        {
            F_invr    = fmaf(qq, fmaf(t.y, r2, t.x), nm); /* both corrections as {intercept, slope} in r^2 */
            float eel = inv_r - nbp.sh_ewald;
            asm("v_and_b32 %0, %1, %2" : "=v"(eel) : "v"(intMask), "v"(eel));
            E_el = qq * (eel - fmaf(t.w, r2, t.z));
        }
        return;
    }
    if constexpr (!ENERGY
                  && (VDW == VDK_CUT || VDW == VDK_COMB_GEOM || VDW == VDK_COMB_LB || VDW == VDK_FSWITCH || VDW == VDK_PSWITCH || VDW == VDK_EWALD_GEOM
                      || VDW == VDK_EWALD_LB)
                  && MASK_FORCES
                  && (ELEC == ELK_RF || ((ELEC == ELK_EWALD_ANA || ELEC == ELK_EWALD_TAB) && CORR_TABLE)))
    {
        /* The force-only flavours of the headline configurations: ONE mask for everything an exclusion removes,
         *   F/r = mask((q q / r + (c12 r^-6 - c6) r^-6) / r^2) + q q corr,
         * instead of masking r^-6 and r^-3 separately.  The AND is written as asm: left to the compiler, a single masked value
         * becomes v_and + v_cmp + v_cndmask (three instructions, two of them half rate, plus the VCC hazard). */
        [[maybe_unused]] float2 t = make_float2(0.0F, 0.0F);
        [[maybe_unused]] float  xs = 0.0F;
        typedef __attribute__((address_space(3))) const float LdsFloat;
        if constexpr (ELEC == ELK_EWALD_ANA)
        {
            if constexpr (c_earlyTableRead) { t = tEarly; }
            else
            {
                /* the table sits at LDS address 0 (nbnxmKernel checks it): no base to add */
                LdsFloat* tab = reinterpret_cast<LdsFloat*>(static_cast<uintptr_t>(ewaldTabAddress<8>(r2, ewaldTabScaleV)));
                t.x           = tab[0];
                t.y           = tab[1];
            }
            (void)ewaldCorrLds;
        }
        if constexpr (ELEC == ELK_EWALD_TAB)
        {
            /* the reference's table, indexed by r (nbnxm_cuda_kernel_utils.cuh:448-459), staged into LDS from address 0 by the kernel */
            xs                 = (r2 * inv_r) * nbp.coulomb_tab_scale;
            const unsigned idx = static_cast<unsigned>(xs);
            LdsFloat*      tab = reinterpret_cast<LdsFloat*>(static_cast<uintptr_t>(idx * 4U));
            t.x                = tab[0];
            t.y                = tab[1];
            (void)ewaldCorrLds;
        }
        float lj = fmaf(c12, inv_r6, -c6) * inv_r6; /* r^2 times the plain LJ F/r */
        /* The switch flavours: every instruction below takes at most ONE scalar (kernel argument) operand.  An FMA with two of them
         * needs a copy of one in a VGPR, which the compiler hoists out of the loops: that is what pushed these flavours over the
         * 96 VGPRs of 5 waves per SIMD. */
        if constexpr (VDW == VDK_FSWITCH)
        {
            /* force switch (nbnxm_cuda_kernel_utils.cuh:180-216), times r^2 like lj (the term's 1/r becomes r):
             * c12 (r2 + r3 rs) - c6 (d2 + d3 rs) = (c12 r2 - c6 d2) + (c12 r3 - c6 d3) rs */
            const float r  = r2 * inv_r;
            const float rs = fmaxf(r - nbp.rvdw_switch, 0.0F);
            const float a  = fmaf(-c6, nbp.dispersion_shift.c2, c12 * nbp.repulsion_shift.c2);
            const float b  = fmaf(-c6, nbp.dispersion_shift.c3, c12 * nbp.repulsion_shift.c3);
            lj             = fmaf(fmaf(b, rs, a) * (rs * rs), r, lj);
        }
        if constexpr (VDW == VDK_PSWITCH)
        {
            /* potential switch (:247-271): F/r = F_lj/r sw - E_lj dsw / r, times r^2; the exclusion mask below covers E_lj too.
             * sw = 1 + (c3 + (c4 + c5 rs) rs) rs^3,  dsw = (3 c3 + (4 c4 + 5 c5 rs) rs) rs^2 with 4 c4 + 5 c5 rs = 4 (c4 + c5 rs) + c5 rs */
            const float r   = r2 * inv_r;
            const float rs  = fmaxf(r - nbp.rvdw_switch, 0.0F);
            const float ea  = fmaf(inv_r6, inv_r6, nbp.repulsion_shift.cpot) * c12;
            const float eb  = (inv_r6 + nbp.dispersion_shift.cpot) * c6;
            const float e   = fmaf(eb, -c_oneSixth, ea * c_oneTwelfth);
            const float rs2 = rs * rs;
            const float u   = rs * nbp.vdw_switch.c5;
            const float v   = u + nbp.vdw_switch.c4;
            const float sw  = fmaf(fmaf(v, rs, nbp.vdw_switch.c3), rs2 * rs, 1.0F);
            const float dsw = fmaf(fmaf(v, 4.0F, u), rs, nbp.vdwSwitch3c3) * rs2;
            lj              = fmaf(-r * e, dsw, lj * sw);
        }
        if constexpr (TWIN) { lj = (r2 < nbp.rvdw_sq) ? lj : 0.0F; } /* rvdw < rcoulomb (what PME tuning leaves) */
        float nm = fmaf(qq, inv_r, lj) * inv_r2;
        asm("v_and_b32 %0, %1, %2" : "=v"(nm) : "v"(intMask), "v"(nm));
        if constexpr (ELEC == ELK_RF) { F_invr = fmaf(qq, -nbp.two_k_rf, nm); }
        else if constexpr (ELEC == ELK_EWALD_ANA)
        {
            /* (the table entry is not touched before the masked part is done: its wait then sits behind the whole LJ / Coulomb chain) */
            if constexpr (c_earlyTableRead) { asm volatile("" : "+v"(t.x), "+v"(t.y) : "v"(nm)); }
            F_invr = fmaf(qq, fmaf(t.y, r2, t.x), nm); /* {intercept, slope} in r^2: no fraction needed */
        }
        else
        {
            const float fr = __builtin_amdgcn_fractf(xs);
            F_invr         = fmaf(-qq * inv_r, fmaf(fr, t.y, fmaf(-fr, t.x, t.x)), nm);
        }
        if constexpr (VDW == VDK_EWALD_GEOM || VDW == VDK_EWALD_LB)
        {
            /* real-space part of the LJ-PME grid term (nbnxm_cuda_kernel_utils.cuh:231-330): outside the exclusion mask — an excluded
             * pair within the cut-off keeps the correction for what the grid adds */
            const float cr2     = nbp.ljEwaldCoeff2 * r2;
            const float expmcr2 = __expf(-cr2);
            const float poly    = fmaf(fmaf(0.5F, cr2, 1.0F), cr2, 1.0F);
            float       grid    = c6grid * (inv_r6 - expmcr2 * fmaf(inv_r6, poly, nbp.ljEwaldCoeff6_6));
            if constexpr (TWIN) { grid = (r2 < nbp.rvdw_sq) ? grid : 0.0F; }
            F_invr = fmaf(grid, inv_r2, F_invr);
        }
        return;
    }
    if constexpr (MASK_FORCES) { inv_r6 = masked(inv_r6); }
    const float inv_r3m = MASK_FORCES ? masked(inv_r2 * inv_r) : inv_r2 * inv_r; /* masked 1/r^3 */

    F_invr       = inv_r6 * (c12 * inv_r6 - c6) * inv_r2;
    float E_lj_p = 0.0F;
    if constexpr (ENERGY || VDW == VDK_PSWITCH)
    {
        E_lj_p = int_bit
                 * (c12 * (inv_r6 * inv_r6 + nbp.repulsion_shift.cpot) * c_oneTwelfth
                    - c6 * (inv_r6 + nbp.dispersion_shift.cpot) * c_oneSixth);
    }
    if constexpr (VDW == VDK_FSWITCH)
    {
        const float r  = r2 * inv_r;
        float       rs = r - nbp.rvdw_switch;
        rs             = rs >= 0.0F ? rs : 0.0F;
        F_invr += (-c6 * (nbp.dispersion_shift.c2 + nbp.dispersion_shift.c3 * rs)
                   + c12 * (nbp.repulsion_shift.c2 + nbp.repulsion_shift.c3 * rs))
                  * rs * rs * inv_r;
        if constexpr (ENERGY)
        {
            E_lj_p += (c6 * (nbp.dispersion_shift.c2 * (1.0F / 3.0F) + nbp.dispersion_shift.c3 * 0.25F * rs)
                       - c12 * (nbp.repulsion_shift.c2 * (1.0F / 3.0F) + nbp.repulsion_shift.c3 * 0.25F * rs))
                      * rs * rs * rs;
        }
    }
    if constexpr (VDW == VDK_EWALD_GEOM || VDW == VDK_EWALD_LB)
    {
        /* real-space part of the LJ-PME grid term (nbnxm_cuda_kernel_utils.cuh:231-330, calculate_lj_ewald_comb_*_F / _F_E):
         * not masked by the exclusion bit — an excluded pair within the cut-off keeps the correction for what the grid adds */
        const float inv_r6_nm = inv_r2 * inv_r2 * inv_r2;
        const float lje2      = nbp.ewaldcoeff_lj * nbp.ewaldcoeff_lj;
        const float lje6_6    = lje2 * lje2 * lje2 * c_oneSixth;
        const float cr2       = lje2 * r2;
        const float expmcr2   = __expf(-cr2);
        const float poly      = 1.0F + cr2 + 0.5F * cr2 * cr2;
        F_invr += c6grid * (inv_r6_nm - expmcr2 * (inv_r6_nm * poly + lje6_6)) * inv_r2;
        if constexpr (ENERGY)
        {
            E_lj_p += c_oneSixth * c6grid * (inv_r6_nm * (1.0F - expmcr2 * poly) + nbp.sh_lj_ewald * int_bit);
        }
    }
    if constexpr (VDW == VDK_PSWITCH)
    {
        const float r  = r2 * inv_r;
        float       rs = r - nbp.rvdw_switch;
        rs             = rs >= 0.0F ? rs : 0.0F;
        const float sw = 1.0F + (nbp.vdw_switch.c3 + (nbp.vdw_switch.c4 + nbp.vdw_switch.c5 * rs) * rs) * rs * rs * rs;
        const float dsw = (3.0F * nbp.vdw_switch.c3 + (4.0F * nbp.vdw_switch.c4 + 5.0F * nbp.vdw_switch.c5 * rs) * rs) * rs * rs;
        F_invr = F_invr * sw - inv_r * E_lj_p * dsw;
        E_lj_p *= sw;
    }
    if constexpr (TWIN)
    {
        const float inRange = (r2 < nbp.rvdw_sq) ? 1.0F : 0.0F;
        F_invr *= inRange;
        E_lj_p *= inRange;
    }
    if constexpr (ENERGY) { E_lj = E_lj_p; }

    if constexpr (ELEC == ELK_CUT)
    {
        F_invr += qq * inv_r3m;
        if constexpr (ENERGY) { E_el = qq * (int_bit * inv_r - nbp.c_rf); }
    }
    else if constexpr (ELEC == ELK_RF)
    {
        F_invr += qq * (inv_r3m - nbp.two_k_rf);
        if constexpr (ENERGY) { E_el = qq * (int_bit * inv_r + 0.5F * nbp.two_k_rf * r2 - nbp.c_rf); }
    }
    else
    {
        const float            beta  = nbp.ewald_beta;
        [[maybe_unused]] float corrV = 0.0F; /* beta V((beta r)^2) from the table (energy flavours of the cluster kernel) */
        if constexpr (ELEC == ELK_EWALD_ANA)
        {
            if constexpr (CORR_TABLE)
            {
                /* beta^3 F((beta r)^2) by linear interpolation in the LDS table; only pairs within rcoulomb get here,
                 * which is what bounds the index */
                const unsigned char* tabBase = reinterpret_cast<const unsigned char*>(ewaldCorrLds);
                if constexpr (ENERGY)
                {
                    /* the energy flavours' table carries the potential correction too */
                    const float4 t = *reinterpret_cast<const float4*>(tabBase + ewaldTabAddress<16>(r2, ewaldTabScaleV));
                    F_invr += qq * (inv_r3m + fmaf(t.y, r2, t.x));
                    corrV = fmaf(t.w, r2, t.z);
                }
                else
                {
                    /* {intercept, slope} of the entry's line in r^2 */
                    const float2 t = *reinterpret_cast<const float2*>(tabBase + ewaldTabAddress<8>(r2, ewaldTabScaleV));
                    F_invr += qq * (inv_r3m + fmaf(t.y, r2, t.x));
                }
            }
            else
            {
                /* the rational form (callers without the table in LDS: the perturbed-cluster-pair kernel) */
                const float beta2 = beta * beta;
                F_invr += qq * (inv_r3m + pmeCorrF(beta2 * r2) * beta2 * beta);
            }
        }
        else
        {
            if constexpr (CORR_TABLE)
            {
                const float    xs  = (r2 * inv_r) * nbp.coulomb_tab_scale;
                const unsigned idx = static_cast<unsigned>(xs);
                const float*   tab = reinterpret_cast<const float*>(ewaldCorrLds);
                const float    fr  = __builtin_amdgcn_fractf(xs);
                const float    d0  = tab[idx];
                F_invr += qq * (inv_r3m - fmaf(fr, tab[idx + 1], fmaf(-fr, d0, d0)) * inv_r);
            }
            else { F_invr += qq * (inv_r3m - interpolateCoulombForceR(nbp, r2 * inv_r) * inv_r); }
        }
        if constexpr (ENERGY && ELEC == ELK_EWALD_ANA)
        {
            /* erf(beta r)/r = beta V(beta^2 r^2): branch-free, shares z^2 with the force correction (the libm
             * erff costs two divergent branches per pair) */
            if constexpr (CORR_TABLE) { E_el = qq * (int_bit * (inv_r - nbp.sh_ewald) - corrV); }
            else { E_el = qq * (int_bit * (inv_r - nbp.sh_ewald) - beta * pmeCorrV(beta * beta * r2)); }
        }
        else if constexpr (ENERGY)
        {
            if constexpr (CORR_TABLE)
            {
                /* cluster kernel, tabulated flavours: erf(beta r)/r = beta V((beta r)^2) from the potential table at LDS address 0 */
                typedef __attribute__((address_space(3))) const float LdsFloat;
                /* (the potential half of the 16-byte entries, staged as 8-byte entries: same spans, half the address) */
                LdsFloat* tabV = reinterpret_cast<LdsFloat*>(static_cast<uintptr_t>(ewaldTabAddress<16>(r2, ewaldTabScaleV) >> 1));
                E_el = qq * (int_bit * (inv_r - nbp.sh_ewald) - fmaf(tabV[1], r2, tabV[0]));
            }
            else { E_el = qq * (inv_r * (int_bit - erff(r2 * inv_r * beta)) - int_bit * nbp.sh_ewald); }
        }
    }
}

/* ---- perturbed atom pair ------------------------------------------------------------------------
 * Semantics: the CPU kernel nb_free_energy_kernel<> (gmxlib/nonbonded/nb_free_energy.cpp:723-1136),
 * Beutler or Gapsys soft-core or none, RF/cut-off or Ewald, LJ cut-off with optional potential switch or LJ-PME
 * (grid correction :121-163, :1103-1136 — the reference's own GPU kernels evaluate plain shifted LJ there);
 * per-interaction cut-offs on the soft-core radii (:804-812,880-890) and the r^-6 cap (:907) as on the
 * CPU, which is where the reference's own CUDA kernel deviates.
 */
struct FepLambda
{
    float LFC[2], LFV[2];       /* lambda factors, state A/B            :420-427 */
    float scLFC[2], scLFV[2];   /* soft-core lambda factors             :437-449 */
    float scDLFC[2], scDLFV[2]; /* soft-core dV/dl factors                       */
    bool  differ;               /* scLambdasOrAlphasDiffer              :1405-1419 */
};

NB_DEVINL FepLambda makeFepLambda(float lambdaCoul, float lambdaVdw, int lamPower, float alphaCoul, float alphaVdw)
{
    FepLambda L;
    L.LFC[0] = 1.0F - lambdaCoul;
    L.LFC[1] = lambdaCoul;
    L.LFV[0] = 1.0F - lambdaVdw;
    L.LFV[1] = lambdaVdw;
    const float lp = static_cast<float>(lamPower);
#pragma unroll
    for (int k = 0; k < 2; k++)
    {
        const float dlf = (k == 0) ? -1.0F : 1.0F;
        const float oc = 1.0F - L.LFC[k], ov = 1.0F - L.LFV[k];
        L.scLFC[k]  = (lamPower == 2) ? oc * oc : oc;
        L.scLFV[k]  = (lamPower == 2) ? ov * ov : ov;
        L.scDLFC[k] = dlf * lp * c_oneSixth * ((lamPower == 2) ? oc : 1.0F);
        L.scDLFV[k] = dlf * lp * c_oneSixth * ((lamPower == 2) ? ov : 1.0F);
    }
    L.differ = !((alphaCoul == 0.0F && alphaVdw == 0.0F) || (lambdaCoul == lambdaVdw && alphaCoul == alphaVdw));
    return L;
}

/* Gapsys soft-core (gmxlib/nonbonded/nb_softcore.h:44-279; the reference's GPU kernels have Beutler only, its CPU kernel both):
 * inside a lambda-dependent radius rQ the Coulomb / LJ interaction of one end state is replaced by its quadratic expansion around rQ.
 * force / potential are overwritten, dvdl is added to, exactly where the CPU kernel does it. */
template<bool EWALD, bool ENERGY>
NB_DEVINL void gapsysCoulomb(float qq, float facel, float r, float rCutoff, float lambdaFac, float dLambdaFac, float alphaEff,
                             float twoKrf, float potentialShift, float& force, float& potential, float& dvdl)
{
    if (!(lambdaFac < 1.0F && 0.0F < alphaEff && facel != 0.0F)) { return; }
    float rQ = sqrtf(cbrtf(1.0F - lambdaFac)) * (1.0F + fabsf(qq / facel)) * alphaEff;
    const bool withinCutoff = (rQ <= rCutoff);
    rQ                      = fminf(rQ, rCutoff);
    if (!(r < rQ)) { return; }
    const float rInvQ    = 1.0F / rQ;
    const float constFac = qq * rInvQ;
    const float linFac   = constFac * r * rInvQ;
    const float quadrFac = linFac * r * rInvQ;
    float       fq       = -2.0F * quadrFac + 3.0F * linFac;
    float       vq       = quadrFac - 3.0F * (linFac - constFac);
    if constexpr (EWALD) { vq -= qq * potentialShift; }
    else
    {
        fq -= qq * twoKrf * r * r;
        vq += qq * (0.5F * twoKrf * r * r - potentialShift);
    }
    force     = fq;
    potential = vq;
    if constexpr (ENERGY)
    {
        if (withinCutoff) { dvdl += dLambdaFac * 0.5F * (lambdaFac / (1.0F - lambdaFac)) * (quadrFac - 2.0F * linFac + constFac); }
    }
}

template<bool ENERGY>
NB_DEVINL void gapsysLJ(float c6, float c12, float r, float rsq, float lambdaFac, float dLambdaFac, float sigma6, float alphaEff,
                        float repulsionShift, float dispersionShift, float& force, float& potential, float& dvdl)
{
    if (!(lambdaFac < 1.0F && 0.0F < alphaEff)) { return; }
    const float lambdaFacRev = 1.0F - lambdaFac;
    const float rQ           = sqrtf(cbrtf((26.0F / 7.0F) * sigma6 * lambdaFacRev)) * alphaEff;
    if (!(r < rQ)) { return; }
    const float c6s = c6 * c_oneSixth, c12s = c12 * c_oneTwelfth;
    const float rInvQ = 1.0F / rQ;
    float       i6    = rInvQ * rInvQ * rInvQ;
    i6                = i6 * i6;
    const float i7 = i6 * rInvQ, i8 = i7 * rInvQ;
    const float rInv14C = c12s * i7 * i7 * rsq, rInv13C = c12s * i7 * i6 * r, rInv12C = c12s * i6 * i6;
    const float rInv8C = i8 * c6s * rsq, rInv7C = i7 * c6s * r, rInv6C = i6 * c6s;
    const float quadrFac  = 156.0F * rInv14C - 42.0F * rInv8C;
    const float linearFac = 168.0F * rInv13C - 48.0F * rInv7C;
    const float constFac  = 91.0F * rInv12C - 28.0F * rInv6C;
    force     = -quadrFac + linearFac;
    potential = 0.5F * quadrFac - linearFac + constFac + (c12s * repulsionShift - c6s * dispersionShift);
    if constexpr (ENERGY)
    {
        dvdl += dLambdaFac * 28.0F * (lambdaFac / lambdaFacRev)
                * ((6.5F * rInv14C - rInv8C) - (13.0F * rInv13C - 2.0F * rInv7C) + (6.5F * rInv12C - rInv6C));
    }
}

/* erf(beta r) / r, the reciprocal-space part an Ewald pair's energy has to give back (nb_free_energy.cpp:1056-1101), with its r -> 0
 * limit; r and 1 / r of the clamped distance as fepPair computes them */
NB_DEVINL float fepEwaldPotentialLr(const float beta, const float r, const float rInv)
{
    return (beta * r > 1.0e-4F) ? erff(beta * r) * rInv : 2.0F * beta * c_oneOverSqrtPi;
}

/* Returns false when the pair is skipped (beyond the cut-off and not an exclusion, :665-678). */
template<int ELEC, bool PSWITCH, bool FORCE, bool ENERGY>
NB_DEVINL bool fepPair(const NBParamGpu& nbp,
                       const FepLambda&  L,
                       float             r2raw,
                       bool              included,
                       bool              iEqJ,
                       const float (&qq)[2],
                       const float (&c6)[2],
                       const float (&c12)[2],
                       float&            fscal,
                       float&            eLJ,
                       float&            eEl,
                       float&            dvdlLJ,
                       float&            dvdlEl,
                       const float (&c6grid)[2] /* LJ-PME: grid C6 of the two states (ljGridC6AB) */,
                       /* Ewald, ENERGY: erf(beta r) / r of this pair if the caller has it already (fepEwaldPotentialLr) — it does not depend
                        * on lambda, and the foreign-lambda loops evaluate the pair a dozen times —; negative: compute it here */
                       const float vLrKnown = -1.0F)
{
    const bool  ljPme  = isLjPme(nbp);
    const float rcMax2 = fmaxf(nbp.rcoulomb_sq, nbp.rvdw_sq);
    if (included && !(r2raw < rcMax2)) { return false; }

    /* dispatchKernel :1315-1364: Beutler without alphas, Gapsys without linearisation points = no soft-core */
    const bool gapsys      = (nbp.softcoreType == NBNXM_SOFTCORE_GAPSYS);
    const bool useSoftCore = !gapsys && (nbp.alpha_coul != 0.0F || nbp.alpha_vdw != 0.0F);
    const bool useGapsys   = gapsys && (nbp.gapsysLinpointCoul != 0.0F || nbp.gapsysLinpointVdw != 0.0F);
    const float r2          = fmaxf(r2raw, c_nbnxnMinDistanceSquared);
    const float rInv        = __frsqrt_rn(r2);
    const float r           = r2 * rInv;
    float       fs          = 0.0F;

    if (included)
    {
        float rp, rpm2;
        if (useSoftCore)
        {
            rpm2 = r2 * r2;
            rp   = rpm2 * r2;
        }
        else
        {
            rpm2 = rInv * rInv;
            rp   = 1.0F;
        }
        float sigma6[2], gapsysSigma6[2];
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            if (c6[k] > 0.0F && c12[k] > 0.0F)
            {
                gapsysSigma6[k] = 0.5F * c12[k] / c6[k];
                sigma6[k]       = fmaxf(gapsysSigma6[k], nbp.sc_sigma6_min);
            }
            else
            {
                sigma6[k]       = nbp.sc_sigma6;
                gapsysSigma6[k] = nbp.gapsysSigma6Vdw;
            }
        }
        const bool  hardCore   = (c12[0] > 0.0F && c12[1] > 0.0F);
        const float alphaVEff  = hardCore ? 0.0F : nbp.alpha_vdw;
        const float alphaCEff  = hardCore ? 0.0F : nbp.alpha_coul;
        const float gapsysLinV = hardCore ? 0.0F : nbp.gapsysLinpointVdw;
        const float gapsysLinC = hardCore ? 0.0F : nbp.gapsysLinpointCoul;

        float fC[2] = { 0.0F, 0.0F }, fV[2] = { 0.0F, 0.0F }, vC[2] = { 0.0F, 0.0F }, vV[2] = { 0.0F, 0.0F };
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            if (qq[k] != 0.0F || c6[k] != 0.0F || c12[k] != 0.0F)
            {
                float rPInvC, rInvC, rC, rPInvV, rInvV, rV;
                if (useSoftCore)
                {
                    rPInvC          = 1.0F / (alphaCEff * L.scLFC[k] * sigma6[k] + rp);
                    const float r2C = rcbrtf(rPInvC);
                    rInvC           = __frsqrt_rn(r2C);
                    rC              = r2C * rInvC;
                    if (L.differ)
                    {
                        rPInvV          = 1.0F / (alphaVEff * L.scLFV[k] * sigma6[k] + rp);
                        const float r2V = rcbrtf(rPInvV);
                        rInvV           = __frsqrt_rn(r2V);
                        rV              = r2V * rInvV;
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
                    rPInvC = 1.0F;
                    rInvC  = rInv;
                    rC     = r;
                    rPInvV = 1.0F;
                    rInvV  = rInv;
                    rV     = r;
                }

                const bool doElec = (qq[k] != 0.0F)
                                    && ((ELEC >= ELK_EWALD_ANA) ? (r2 < nbp.rcoulomb_sq) : (rC < nbp.rcoulomb));
                if (doElec)
                {
                    if constexpr (ELEC >= ELK_EWALD_ANA)
                    {
                        vC[k] = qq[k] * (rInvC - nbp.sh_ewald);
                        fC[k] = qq[k] * rInvC;
                        if (useGapsys)
                        {
                            gapsysCoulomb<true, ENERGY>(qq[k], nbp.epsfac, rC, nbp.rcoulomb, L.LFC[k], (k == 0) ? -1.0F : 1.0F, gapsysLinC, 0.0F,
                                                        nbp.sh_ewald, fC[k], vC[k], dvdlEl);
                        }
                    }
                    else
                    {
                        /* plain cut-off is reaction-field with k_rf = 0 (:377-386) */
                        vC[k] = qq[k] * (rInvC + 0.5F * nbp.two_k_rf * rC * rC - nbp.c_rf);
                        fC[k] = qq[k] * (rInvC - nbp.two_k_rf * rC * rC);
                        if (useGapsys)
                        {
                            gapsysCoulomb<false, ENERGY>(qq[k], nbp.epsfac, rC, nbp.rcoulomb, L.LFC[k], (k == 0) ? -1.0F : 1.0F, gapsysLinC,
                                                         nbp.two_k_rf, nbp.c_rf, fC[k], vC[k], dvdlEl);
                        }
                    }
                }
                const bool doVdw = (c6[k] != 0.0F || c12[k] != 0.0F) && ((ljPme ? r : rV) < nbp.rvdw); /* :880-890 */
                if (doVdw)
                {
                    float rInv6;
                    if (useSoftCore) { rInv6 = rPInvV; }
                    else
                    {
                        rInv6 = rInvV * rInvV;
                        rInv6 = rInv6 * rInv6 * rInv6;
                    }
                    rInv6           = fminf(rInv6, c_maxRInvSix);
                    const float v6  = c6[k] * rInv6;
                    const float v12 = c12[k] * rInv6 * rInv6;
                    vV[k]           = (v12 + c12[k] * nbp.repulsion_shift.cpot) * c_oneTwelfth
                            - (v6 + c6[k] * nbp.dispersion_shift.cpot) * c_oneSixth;
                    fV[k] = v12 - v6;
                    if (useGapsys)
                    {
                        gapsysLJ<ENERGY>(c6[k], c12[k], r, r2, L.LFV[k], (k == 0) ? -1.0F : 1.0F, gapsysSigma6[k], gapsysLinV,
                                         nbp.repulsion_shift.cpot, nbp.dispersion_shift.cpot, fV[k], vV[k], dvdlLJ);
                    }
                    if (ljPme) { vV[k] += c6grid[k] * nbp.sh_lj_ewald * c_oneSixth; } /* :937-944 */
                    if constexpr (PSWITCH)
                    {
                        float d        = rV - nbp.rvdw_switch;
                        d              = (0.0F < d) ? d : 0.0F;
                        const float d2 = d * d;
                        const float sw = 1.0F + d2 * d * (nbp.vdw_switch.c3 + d * (nbp.vdw_switch.c4 + d * nbp.vdw_switch.c5));
                        const float dsw = d2 * (3.0F * nbp.vdw_switch.c3 + d * (4.0F * nbp.vdw_switch.c4 + d * 5.0F * nbp.vdw_switch.c5));
                        fV[k] = fV[k] * sw - rV * vV[k] * dsw;
                        vV[k] = vV[k] * sw;
                    }
                }
                fC[k] *= rPInvC;
                fV[k] *= rPInvV;
            }
        }
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            const float dlf = (k == 0) ? -1.0F : 1.0F;
            if constexpr (FORCE)
            {
                fs += (L.LFC[k] * fC[k] + L.LFV[k] * fV[k]) * rpm2;
            }
            if constexpr (ENERGY)
            {
                eEl += L.LFC[k] * vC[k];
                eLJ += L.LFV[k] * vV[k];
                dvdlEl += vC[k] * dlf;
                dvdlLJ += vV[k] * dlf;
                if (useSoftCore)
                {
                    dvdlEl += L.LFC[k] * alphaCEff * L.scDLFC[k] * fC[k] * sigma6[k];
                    dvdlLJ += L.LFV[k] * alphaVEff * L.scDLFV[k] * fV[k] * sigma6[k];
                }
            }
        }
    }
    else if constexpr (ELEC < ELK_EWALD_ANA)
    {
        /* excluded pair, cut-off / reaction-field (:1023-1054) */
        float VV = 0.5F * nbp.two_k_rf * r2 - nbp.c_rf;
        if (iEqJ) { VV *= 0.5F; }
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            const float dlf = (k == 0) ? -1.0F : 1.0F;
            if constexpr (FORCE) { fs += L.LFC[k] * qq[k] * (-nbp.two_k_rf); }
            if constexpr (ENERGY)
            {
                eEl += L.LFC[k] * qq[k] * VV;
                dvdlEl += dlf * qq[k] * VV;
            }
        }
    }

    if constexpr (ELEC >= ELK_EWALD_ANA)
    {
        /* remove the reciprocal-space part (:1056-1101) */
        if (!included || r2 < nbp.rcoulomb_sq)
        {
            const float beta = nbp.ewald_beta;
            float       f_lr = 0.0F;
            if constexpr (FORCE)
            {
                /* analytical in the tabulated flavours too: the CPU free-energy kernel this path is checked against has no table
                 * (nb_free_energy.cpp:1064-1068, pmeForceCorrection); the reference's CUDA kernel interpolates its table here */
                f_lr = -pmeCorrF(beta * beta * r2) * beta * beta * beta;
            }
            float v_lr = 0.0F;
            if constexpr (ENERGY)
            {
                /* erf(beta r)/r, with its r -> 0 limit for the self pair */
                v_lr = (vLrKnown >= 0.0F) ? vLrKnown : fepEwaldPotentialLr(beta, r, rInv);
                if (iEqJ) { v_lr *= 0.5F; }
            }
#pragma unroll
            for (int k = 0; k < 2; k++)
            {
                const float dlf = (k == 0) ? -1.0F : 1.0F;
                if constexpr (FORCE) { fs -= L.LFC[k] * qq[k] * f_lr; }
                if constexpr (ENERGY)
                {
                    eEl -= L.LFC[k] * qq[k] * v_lr;
                    dvdlEl -= dlf * qq[k] * v_lr;
                }
            }
        }
    }
    if (ljPme && (!included || r < nbp.rvdw))
    {
        /* remove the grid part (:1103-1136, ewaldLennardJonesGridSubtract :121-163) */
        const float lje2    = nbp.ewaldcoeff_lj * nbp.ewaldcoeff_lj;
        const float lje6_6  = lje2 * lje2 * lje2 * c_oneSixth;
        const float rInvSq  = rInv * rInv;
        const float rInvSix = rInvSq * rInvSq * rInvSq;
        const float x       = lje2 * r2;
        const float expNegX = __expf(-x);
        const float poly    = 1.0F + x + 0.5F * x * x;
        /* below (8 eps)^(1/6) the series: the closed form cancels */
        const float term = (x < 0.09921257F) ? lje6_6 * (1.0F + x * (-0.75F + 0.3F * x)) : rInvSix * (1.0F - expNegX * poly);
        const float f_lr = (term - expNegX * lje6_6) * rInvSq;
        const float v_lr = (iEqJ ? 0.5F * lje6_6 : term) * c_oneSixth;
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            const float dlf = (k == 0) ? -1.0F : 1.0F;
            if constexpr (FORCE) { fs += L.LFV[k] * c6grid[k] * f_lr; }
            if constexpr (ENERGY)
            {
                eLJ += L.LFV[k] * c6grid[k] * v_lr;
                dvdlLJ += dlf * c6grid[k] * v_lr;
            }
        }
    }
    fscal = fs;
    return true;
}

#endif
