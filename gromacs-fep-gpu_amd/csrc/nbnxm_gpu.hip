/*
 * C ABI of the MI355X Nbnxm FEP path (include/nbnxm_hip.h): data management, kernel launch,
 * copy-back and task completion.  Host-side control flow follows the reference:
 *   nbnxm/nbnxm_gpu_data_mgmt.cpp   gpu_init :538-628, initNbparam :421-489, cuda_copy_fepparams :491-536,
 *                                   gpu_init_pairlist :667-759, gpu_init_feppairlist :761-871,
 *                                   gpu_init_atomdata :873-1045, gpu_clear_outputs :1047-1070,
 *                                   gpu_launch_cpyback :1117-1303, gpu_free :1540-1654
 *   nbnxm/cuda/nbnxm_cuda.cu        gpu_launch_kernel :642-858, gpu_launch_kernel_pruneonly :873-994
 *   nbnxm/gpu_common.h              gpu_reduce_staged_outputs :139-168, gpu_reduce_staged_foreign_term :178-191,
 *                                   gpu_try_finish_task :293-385, gpu_wait_finish_task :405-435
 * Deliberate deviations from the reference's quirks are marked "A.4".
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "nbnxm_gpu_internal.h"
#include "nbnxm_kernels.h"
#include "nbnxm_work_partition.h"

namespace nbnxm_hip
{
static thread_local std::string g_lastError;
void setLastError(const char* msg)
{
    g_lastError = msg ? msg : "";
}
} // namespace nbnxm_hip


/* Clears the force array and the block of scalar outputs (energies, dV/dl, foreign terms, shift forces)
 * in one launch; replaces the 5-10 separate memsets of gpu_clear_outputs (nbnxm_gpu_data_mgmt.cpp:1047-1070). */
__global__ void nbnxmClearOutputsKernel(float4* __restrict__ f4, int numFloat4, float* __restrict__ tail, int numTail,
                                        float* __restrict__ scalars, int numScalars, float* __restrict__ fshift, int numFshift,
                                        float* __restrict__ windowSlots, int numWindowFloats)
{
    const int    gid    = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
    const int    stride = static_cast<int>(gridDim.x * blockDim.x);
    const float4 zero   = make_float4(0.0F, 0.0F, 0.0F, 0.0F);
    for (int i = gid; i < numFloat4; i += stride) { f4[i] = zero; }
    if (gid < numTail) { tail[gid] = 0.0F; }
    for (int i = gid; i < numScalars; i += stride) { scalars[i] = 0.0F; }
    for (int i = gid; i < numFshift; i += stride) { fshift[i] = 0.0F; }
    for (int i = gid; i < numWindowFloats; i += stride) { windowSlots[i] = 0.0F; }
}

/* Shape check of an uploaded pair list ON the device (gpu_init_pairlist): a j-cluster or an exclusion index outside its array would
 * fault in the kernels.  One thread per packed group; a bad value is replaced by 0 — nothing can fault — and reported through a flag
 * in mapped host memory, which the host looks at when it next launches or finishes on the object (a fatal error then).  On the host the
 * same check means reading the whole list once more — 0.12 ms for the 96k box's 60 k groups, cold from DRAM — in a call that otherwise
 * only queues DMAs. */
__global__ void nbnxmValidateListKernel(nbnxn_cj_packed_t* __restrict__ cjPacked, const int ncjPacked, const int numClusters, const int nexcl,
                                        int* __restrict__ errorFlag)
{
    const int j = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= ncjPacked) { return; }
    nbnxn_cj_packed_t g   = cjPacked[j];
    int               bad = 0;
    for (int m = 0; m < c_jGroupSize; m++)
    {
        if (g.cj[m] < 0 || g.cj[m] >= numClusters)
        {
            g.cj[m] = 0;
            bad |= 1;
        }
    }
    for (auto& im : g.imei)
    {
        if (im.excl_ind < 0 || im.excl_ind >= nexcl)
        {
            im.excl_ind = 0;
            bad |= 2;
        }
    }
    if (bad != 0)
    {
        cjPacked[j] = g;
        atomicOr(errorFlag, bad);
        __threadfence_system();
    }
}

namespace
{

/* the flag of nbnxmValidateListKernel: 1 = j-cluster outside the atom range, 2 = exclusion index out of range */
void checkListErrorFlag(const NbnxmGpu* nb)
{
    if (nb->h_listError == nullptr) { return; }
    const int e = *static_cast<volatile const int*>(nb->h_listError);
    NBNXM_ASSERT((e & 1) == 0, "pair list: j-cluster outside the atom range");
    NBNXM_ASSERT((e & 2) == 0, "pair list: exclusion index out of range");
}

void setCutoffParameters(NBParamGpu* nbp, const nbnxm_interaction_params_t* ic)
{
    /* set_cutoff_parameters, nbnxm_gpu_data_mgmt.cpp:201-223 */
    nbp->ewald_beta        = ic->ewaldcoeff_q;
    nbp->sh_ewald          = ic->sh_ewald;
    nbp->epsfac            = ic->epsfac;
    nbp->two_k_rf          = 2.0F * ic->k_rf;
    nbp->c_rf              = ic->c_rf;
    nbp->rvdw_sq           = ic->rvdw * ic->rvdw;
    nbp->rcoulomb_sq       = ic->rcoulomb * ic->rcoulomb;
    nbp->rcoulomb          = ic->rcoulomb;
    nbp->rvdw              = ic->rvdw;
    nbp->rlistOuter_sq     = ic->rlistOuter * ic->rlistOuter;
    nbp->rlistInner_sq     = ic->rlistInner * ic->rlistInner;
    nbp->useDynamicPruning = ic->useDynamicPruning != 0;
    nbp->sh_lj_ewald       = ic->sh_lj_ewald;
    nbp->ewaldcoeff_lj     = ic->ewaldcoeff_lj;
    nbp->rvdw_switch       = ic->rvdw_switch;
    nbp->dispersion_shift  = ic->dispersion_shift;
    nbp->repulsion_shift   = ic->repulsion_shift;
    nbp->vdw_switch        = ic->vdw_switch;
    nbp->vdwSwitch3c3      = 3.0F * ic->vdw_switch.c3;
    {
        const double c2      = static_cast<double>(ic->ewaldcoeff_lj) * ic->ewaldcoeff_lj;
        nbp->ljEwaldCoeff2   = static_cast<float>(c2);
        nbp->ljEwaldCoeff6_6 = static_cast<float>(c2 * c2 * c2 / 6.0);
    }
}

/* The electrostatics kernel type that runs for the caller's pick.  The reference picks the TABULATED Ewald kernels by default on AMD
 * devices (nbnxm_gpu_data_mgmt.cpp:120-145; analytical on NVIDIA) — a speed decision, both evaluate the same real-space Ewald term.
 * Here the analytical flavours are the fast ones (their correction is one LDS read and one FMA, nbnxm_device_helpers.h: 0.0515 against
 * 0.0624 ms per force step on the 96k box), so a tabulated pick runs them whenever beta r_c is inside the range their table and the
 * perturbed-pair fit cover; NBNXM_HIP_KEEP_TAB_KERNELS=1 keeps the caller's pick (the reference's r-indexed table staged in LDS). */
int kernelElecType(const NbnxmGpu* nb, const nbnxm_interaction_params_t* ic)
{
    const bool tabulated  = (ic->elecType == NBNXM_ELEC_EWALD_TAB || ic->elecType == NBNXM_ELEC_EWALD_TAB_TWIN);
    const bool fitCovers  = ic->ewaldcoeff_q * ic->ewaldcoeff_q * ic->rcoulomb * ic->rcoulomb <= 12.0F;
    if (tabulated && fitCovers && !nb->keepTabulatedKernels)
    {
        return ic->elecType == NBNXM_ELEC_EWALD_TAB ? NBNXM_ELEC_EWALD_ANA : NBNXM_ELEC_EWALD_ANA_TWIN;
    }
    return ic->elecType;
}

void uploadCoulombTable(NbnxmGpu* nb, const nbnxm_interaction_params_t* ic)
{
    NBParamGpu* nbp = nb->nbparam;
    if (ic->coulomb_tab != nullptr && ic->coulomb_tab_size > 0)
    {
        if (ic->coulomb_tab_size > nb->coulomb_tab_n)
        {
            freeDeviceBuffer(&nbp->coulomb_tab);
            allocateDeviceBuffer(&nbp->coulomb_tab, ic->coulomb_tab_size);
            nb->coulomb_tab_n = ic->coulomb_tab_size;
        }
        copyToDeviceBuffer(&nbp->coulomb_tab, ic->coulomb_tab, 0, ic->coulomb_tab_size, nb->deviceStreams[0].stream, false);
        nbp->coulomb_tab_scale = ic->coulomb_tab_scale;
        nbp->coulombTabSize    = ic->coulomb_tab_size;
    }
    const bool tabulated = (nbp->elecType == NBNXM_ELEC_EWALD_TAB || nbp->elecType == NBNXM_ELEC_EWALD_TAB_TWIN);
    NBNXM_ASSERT(!tabulated || nbp->coulomb_tab != nullptr, "tabulated Ewald kernel selected without a force table");
}

/* beta^3 F((beta r)^2), F(x) = (2/sqrt(pi) z exp(-z^2) - erf z) / z^3 with x = z^2 (definition as gmx::pmeForceCorrection,
 * simd/simd_math.h:1560-1650), tabulated for the cluster kernel: see NBParamGpu::ewaldCorrTab */
void uploadEwaldCorrectionTable(NbnxmGpu* nb)
{
    NBParamGpu* nbp = nb->nbparam;
    /* (the tabulated flavours use the potential part on energy steps) */
    if (nbp->elecType != NBNXM_ELEC_EWALD_ANA && nbp->elecType != NBNXM_ELEC_EWALD_ANA_TWIN && nbp->elecType != NBNXM_ELEC_EWALD_TAB
        && nbp->elecType != NBNXM_ELEC_EWALD_TAB_TWIN)
    {
        return;
    }
    const double beta = nbp->ewald_beta;
    const double xMax = beta * beta * nbp->rcoulomb_sq * (1.0 + 1.0e-5);
    const int    n    = c_ewaldCorrTabSize;
    auto         F    = [](double x) {
        if (x < 1.0e-2)
        {
            /* series: 2/sqrt(pi) sum_k (-1)^k 2k x^(k-1) / ((2k+1) k!) , k >= 1 */
            double s = 0, xp = 1, kf = 1;
            for (int k = 1; k < 10; k++)
            {
                kf *= k;
                const double term = xp * 2.0 * k / ((2.0 * k + 1.0) * kf);
                s += (k & 1) ? -term : term;
                xp *= x;
            }
            return 2.0 / std::sqrt(M_PI) * s;
        }
        const double z = std::sqrt(x);
        return (2.0 / std::sqrt(M_PI) * z * std::exp(-x) - std::erf(z)) / (x * z);
    };
    /* V(x) = erf(z) / z */
    auto V = [](double x) {
        if (x < 1.0e-2)
        {
            /* series: 2/sqrt(pi) sum_k (-1)^k x^k / ((2k+1) k!) */
            double s = 0, xp = 1, kf = 1;
            for (int k = 0; k < 10; k++)
            {
                if (k > 0) { kf *= k; }
                const double term = xp / ((2.0 * k + 1.0) * kf);
                s += (k & 1) ? -term : term;
                xp *= x;
            }
            return 2.0 / std::sqrt(M_PI) * s;
        }
        const double z = std::sqrt(x);
        return std::erf(z) / z;
    };
    nb->h_ewaldCorrTab.resize(n);
    nb->h_ewaldCorrTabFV.resize(n);
    const double b3 = beta * beta * beta;
    /* entry k: the line through the function at (k - 1/16) h and (k + 15/16) h in u = r^2, h = u_max / (n - 1): the kernel rounds
     * 8 r^2 / h to an integer and drops its low three bits (ewaldTabAddress), which selects exactly that span; entry 0 reaches to
     * -h/16 (F and V are analytic at 0: the series) and entry n - 1 covers the cut-off.
     * Force flavours {intercept, slope} of beta^3 F, energy flavours the same for beta^3 F and beta V: value = a + b r^2, one FMA with
     * the r^2 the pair block already holds. */
    const double uMax = xMax / (beta * beta), du = uMax / (n - 1);
    const int    nE   = c_ewaldCorrTabSizeEnergy;
    const double duE  = uMax / (nE - 1); /* the energy flavours' table: fewer, wider intervals (c_ewaldCorrTabSizeEnergy) */
    /* (the 16-byte entries of the energy flavours: 16 r^2 / h rounded, four bits dropped: spans from (k - 1/32) h) */
    auto line = [&](auto&& fn, double scale, int k, double eighth, double du, double& a, double& b) {
        const double uL = (k - eighth) * du, uR = (k + 1.0 - eighth) * du;
        const double y0 = scale * fn(beta * beta * uL), y1 = scale * fn(beta * beta * uR);
        b = (y1 - y0) / du;
        a = y0 - b * uL;
    };
    for (int k = 0; k < n; k++)
    {
        double a, b, aV, bV;
        line(F, b3, k, 1.0 / 16.0, du, a, b);
        nb->h_ewaldCorrTab.data[k] = make_float2(static_cast<float>(a), static_cast<float>(b));
        /* (entries nE .. n - 1 of the energy table are never addressed: r^2 stays below the cut-off; filled with the last interval) */
        const int kE = std::min(k, nE - 1);
        line(F, b3, kE, 1.0 / 32.0, duE, a, b);
        line(V, beta, kE, 1.0 / 32.0, duE, aV, bV);
        nb->h_ewaldCorrTabFV.data[k] = make_float4(static_cast<float>(a), static_cast<float>(b), static_cast<float>(aV), static_cast<float>(bV));
    }
    if (nbp->ewaldCorrTab == nullptr)
    {
        allocateDeviceBuffer(&nbp->ewaldCorrTab, n);
        allocateDeviceBuffer(&nbp->ewaldCorrTabFV, n);
    }
    copyToDeviceBuffer(&nbp->ewaldCorrTab, nb->h_ewaldCorrTab.data, 0, n, nb->deviceStreams[0].stream, true);
    copyToDeviceBuffer(&nbp->ewaldCorrTabFV, nb->h_ewaldCorrTabFV.data, 0, n, nb->deviceStreams[0].stream, true);
    nbp->ewaldCorrTabScale8  = static_cast<float>(8.0 / du);
    nbp->ewaldCorrTabScale16 = static_cast<float>(16.0 / duE);
}

bool canSkipNonbondedWork(const NbnxmGpu& nb, int iloc)
{
    return iloc == NBNXM_NONLOCAL && nb.plist[iloc]->nsci == 0;
}

void initFeplist(gpu_feplist* l)
{
    std::memset(l, 0, sizeof(*l));
    l->maxnri = l->maxnshift = l->maxnjidx = l->maxnrj = l->maxnexcl = -1;
}

void accumulateTimings(NbnxmGpu* nb, int iloc)
{
    if (!nb->bDoTime) { return; }
    InteractionTimers& t = nb->timers[iloc];
    t.nb_k.accumulate();
    t.fep_k.accumulate();
    t.prune_k.accumulate();
    nb->timings.nb_k_ms       = nb->timers[0].nb_k.totalMs + nb->timers[1].nb_k.totalMs;
    nb->timings.nb_k_count    = nb->timers[0].nb_k.count + nb->timers[1].nb_k.count;
    nb->timings.fep_k_ms      = nb->timers[0].fep_k.totalMs + nb->timers[1].fep_k.totalMs;
    nb->timings.fep_k_count   = nb->timers[0].fep_k.count + nb->timers[1].fep_k.count;
    nb->timings.prune_k_ms    = nb->timers[0].prune_k.totalMs + nb->timers[1].prune_k.totalMs;
    nb->timings.prune_k_count = nb->timers[0].prune_k.count + nb->timers[1].prune_k.count;
}

/* Workgroup shape of the cluster-pair kernel (host arithmetic only).  Every workgroup holds its own copy of the tables in LDS.  The
 * default is one workgroup of 4 waves (one per SIMD) per wave slot; the LJ table of a force field with many atom types (8 numTypes^2
 * bytes: 32 KB at 64 types) makes that many copies overflow the CU's 160 KB, and the dispatcher would silently keep fewer waves resident.
 * Then fewer, larger workgroups share a copy — 8 waves (two workgroups per CU) or 16 (one) — at 4 waves per SIMD: measured on MI355X
 * the force kernel loses 4 % from 5 to 4 waves per SIMD, but 2x from 5 to 2. */
struct NbLaunchShape
{
    int wavesPerBlock, wavesPerSimd, ldsBytes;
};
static NbLaunchShape chooseNbLaunchShape(int elecType, int vdwType, bool energy, int numTypes, int coulombTabSize, int defaultWavesPerBlock)
{
    const bool ljEwald        = (vdwType == NBNXM_VDW_EWALD_GEOM || vdwType == NBNXM_VDW_EWALD_LB);
    const bool useTable       = (vdwType == NBNXM_VDW_CUT || vdwType == NBNXM_VDW_FSWITCH || vdwType == NBNXM_VDW_PSWITCH || ljEwald);
    const bool ewaldCorrTable = (elecType == NBNXM_ELEC_EWALD_ANA || elecType == NBNXM_ELEC_EWALD_ANA_TWIN);
    const bool ewaldRTable    = (elecType == NBNXM_ELEC_EWALD_TAB || elecType == NBNXM_ELEC_EWALD_TAB_TWIN);
    /* analytical: {F, step} or, on energy steps, {F, step, V, step}; tabulated: the caller's r-indexed force table, on energy steps behind
     * the {V, step} part of the correction table */
    const int  ewaldTableBytes = ewaldCorrTable ? (energy ? c_ewaldCorrTabSizeEnergy * static_cast<int>(sizeof(float4)) : c_ewaldCorrTabSize * static_cast<int>(sizeof(float2)))
                                                : (ewaldRTable ? coulombTabLdsBytes(coulombTabSize)
                                                                         + (energy ? c_ewaldCorrTabSizeEnergy * static_cast<int>(sizeof(float2)) : 0)
                                                               : 0);
    const int  compiledWavesPerSimd = nbKernelWavesPerEu(vdwType, energy, false);
    NbLaunchShape shape{ 0, 0, 0 };
    for (const int w : { defaultWavesPerBlock, 2 * c_nbWavesPerBlock, 4 * c_nbWavesPerBlock })
    {
        const int bytes    = nbLdsBytes(numTypes, useTable, ljEwald, ewaldTableBytes, w);
        const int lds      = (bytes + c_ldsAllocGranularity - 1) / c_ldsAllocGranularity * c_ldsAllocGranularity;
        const int resident = std::min(c_simdsPerCu * compiledWavesPerSimd / w, c_ldsBytesPerCu / lds) * w / c_simdsPerCu;
        if (resident > shape.wavesPerSimd) { shape = NbLaunchShape{ w, resident, bytes }; }
    }
    return shape;
}

} // namespace

/* Which sets of work ranges one call launches and whether the trailing workgroups (perturbed cluster pairs, a pending rolling-prune
 * part, the clear of the spare force buffer) ride with it.  Host arithmetic only, so that the C ABI can expose it and the CPU tests
 * can walk through every case (tests/test_launch_plan.py).
 *   launchPart 0: a plain nbnxm_gpu_launch_kernel: everything, both sets back to back when the list is partitioned in two;
 *              1: nbnxm_gpu_launch_kernel_part(.., 1): the first set only; the whole launch, tail included, when there is one set;
 *              2: nbnxm_gpu_launch_kernel_part(.., 2): the second set with the tail; NOTHING when there is one set.
 * A plan that launches nothing has withTail == false: no state of the step (pending prune part, spare-buffer flag) may be consumed
 * by a call that queues no kernel (that was the bug fixed in c622397). */
struct NbLaunchPlan
{
    int  firstSet;  /* sets firstSet .. firstSet + numSets - 1 are launched, one kernel each */
    int  numSets;   /* 0: nothing to launch */
    int  setRanges; /* ranges (= waves of the main part) per set; offset of set k in the range arrays: k * setRanges */
    bool withTail;  /* the trailing workgroups ride with the LAST launched set */
};
static NbLaunchPlan planNbLaunch(int launchPart, int workParts, int numRanges)
{
    NbLaunchPlan plan;
    /* two sets exist only when the partition made them (workParts 2 always yields an even count; the test is a guard) */
    const bool twoSets = (workParts == 2 && numRanges >= 2 && numRanges % 2 == 0);
    plan.setRanges     = twoSets ? numRanges / 2 : numRanges;
    if (numRanges <= 0)
    {
        plan.firstSet = 0;
        plan.numSets  = 0;
    }
    else if (!twoSets)
    {
        plan.firstSet = 0;
        plan.numSets  = (launchPart == 2) ? 0 : 1; /* a list that is not partitioned in two ran completely with the first part */
    }
    else
    {
        plan.firstSet = (launchPart == 2) ? 1 : 0;
        plan.numSets  = (launchPart == 0) ? 2 : 1;
    }
    const int lastSet = plan.firstSet + plan.numSets - 1;
    plan.withTail     = plan.numSets > 0 && (!twoSets || lastSet == 1);
    return plan;
}

extern "C" {

void nbnxm_hip_query_launch_shape(int elecType, int vdwType, int computeEnergy, int numTypes, int coulombTabSize, int* wavesPerWorkgroup,
                                  int* wavesPerSimd, int* ldsBytesPerWorkgroup)
{
    const NbLaunchShape shape = chooseNbLaunchShape(elecType, vdwType, computeEnergy != 0, numTypes, coulombTabSize, c_nbWavesPerBlock);
    *wavesPerWorkgroup    = shape.wavesPerBlock;
    *wavesPerSimd         = shape.wavesPerSimd;
    *ldsBytesPerWorkgroup = shape.ldsBytes;
}

void nbnxm_hip_query_launch_plan(int launchPart, int workParts, int numRanges, int* firstSet, int* numSets, int* setRanges, int* withTail)
{
    const NbLaunchPlan plan = planNbLaunch(launchPart, workParts, numRanges);
    *firstSet               = plan.firstSet;
    *numSets                = plan.numSets;
    *setRanges              = plan.setRanges;
    *withTail               = plan.withTail ? 1 : 0;
}

int nbnxm_hip_abi_version(void)
{
    return 1;
}

const char* nbnxm_hip_last_error(void)
{
    return g_lastError.c_str();
}

/* the kernels' small outputs live in outputsBlock[which]: [scalar-output block | shift-force block] */
static void pointOutputsAt(NbnxmGpu* nb, int which)
{
    NBAtomDataGpu* ad   = nb->atdat;
    const int      n1   = nb->n_lambda + 1;
    nb->outputsActive   = which;
    nb->scalarOutputs   = nb->outputsBlock[which];
    ad->eLJ             = nb->scalarOutputs + 0;
    ad->eElec           = nb->scalarOutputs + 1;
    ad->dvdlLJ          = nb->scalarOutputs + 2;
    ad->dvdlElec        = nb->scalarOutputs + 3;
    ad->eLJForeign      = nb->scalarOutputs + 4;
    ad->eElecForeign    = ad->eLJForeign + n1;
    ad->dvdlLJForeign   = ad->eElecForeign + n1;
    ad->dvdlElecForeign = ad->dvdlLJForeign + n1;
    ad->energySlots     = nb->scalarOutputs + nb->slotOffset;
    ad->foreignSlots    = nb->scalarOutputs + nb->foreignSlotOffset;
    ad->fShift          = reinterpret_cast<decltype(ad->fShift)>(nb->scalarOutputs + nb->numScalarOutputs);
}

NbnxmGpu* nbnxm_gpu_init(const nbnxm_interaction_params_t* ic, int numTypes, const float* nbfp,
                         const float* nbfp_comb, int bLocalAndNonlocal, int bFEP, int n_lambda,
                         void* localStream, void* nonLocalStream)
{
    NBNXM_ASSERT(ic != nullptr && nbfp != nullptr && numTypes > 0, "gpu_init needs interaction parameters and the nbfp table");
    int numDevices = 0;
    if (hipGetDeviceCount(&numDevices) != hipSuccess || numDevices == 0)
    {
        fatal(__FILE__, __LINE__, "nbnxm_gpu_init", "no HIP device available: this path has no CPU fallback");
    }
    auto* nb           = new NbnxmGpu;
    nb->bUseTwoStreams = bLocalAndNonlocal != 0;
    nb->n_lambda       = n_lambda;
    nb->atdat          = new NBAtomDataGpu;
    nb->nbparam        = new NBParamGpu;
    std::memset(nb->atdat, 0, sizeof(NBAtomDataGpu));
    std::memset(nb->nbparam, 0, sizeof(NBParamGpu));
    for (int i = 0; i < (nb->bUseTwoStreams ? 2 : 1); i++)
    {
        nb->plist[i] = new gpu_plist;
        std::memset(nb->plist[i], 0, sizeof(gpu_plist));
        nb->plist[i]->sci_nalloc = nb->plist[i]->cjPacked_nalloc = nb->plist[i]->imask_nalloc = nb->plist[i]->excl_nalloc = -1;
        nb->plist[i]->sciSorted_nalloc = nb->plist[i]->groupWeight_nalloc = nb->plist[i]->weightBlockSum_nalloc = -1;
        for (int p = 0; p < c_numWorkPartitions; p++) { nb->plist[i]->work_nalloc[p] = -1; }
        nb->feplist[i] = new gpu_feplist;
        initFeplist(nb->feplist[i]);
    }
    nb->deviceStreams[0].init(localStream);
    {
        int device = 0;
        hipDeviceProp_t prop;
        NBNXM_HIP_CHECK(hipGetDevice(&device));
        NBNXM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
        nb->numSimds = prop.multiProcessorCount * 4; /* CDNA: 4 SIMDs per CU */
        if (const char* env = diagnosticsEnv("NBNXM_HIP_NUM_WORK_RANGES")) { nb->numWorkRangesOverride = std::atoi(env); }
        if (const char* env = diagnosticsEnv("NBNXM_HIP_NUM_WORK_RANGES4")) { nb->numWorkRangesEnergy = std::atoi(env); }
        if (const char* env = diagnosticsEnv("NBNXM_HIP_MIN_GROUPS_PER_WAVE")) { nb->minGroupsPerWave = std::max(1, std::atoi(env)); }
        if (const char* env = diagnosticsEnv("NBNXM_HIP_WORK_WEIGHTS"))
        {
            if (std::sscanf(env, "%d,%d,%d", &nb->workWeightsOverride[0], &nb->workWeightsOverride[1], &nb->workWeightsOverride[2]) != 3)
            {
                nb->workWeightsOverride[0] = -1;
            }
        }
        if (const char* env = diagnosticsEnv("NBNXM_HIP_CLASS_SHARES_SHORT"))
        {
            /* (experiment: shares of the age classes of a SHORT list's partition, three or four ranges per SIMD, oldest first) */
            if (std::sscanf(env, "%d,%d,%d,%d", &nb->waveClassShareShort[0], &nb->waveClassShareShort[1], &nb->waveClassShareShort[2], &nb->waveClassShareShort[3]) < 3)
            {
                nb->waveClassShareShort[0] = 0;
            }
        }
        /* NBNXM_HIP_CLASS_SHARES4 / 5 = "s0,s1,.." in 1/1024 of an average range, oldest wave of a SIMD first (renormalised) */
        for (int p = 0; p < 2; p++)
        {
            const char* env = diagnosticsEnv(p == 0 ? "NBNXM_HIP_CLASS_SHARES4" : "NBNXM_HIP_CLASS_SHARES5");
            if (env == nullptr) { continue; }
            int       v[5] = { 0, 0, 0, 0, 0 };
            const int got = std::sscanf(env, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]);
            long long sum = 0;
            for (int k = 0; k < 4 + p; k++) { sum += v[k]; }
            if (got >= 4 + p && sum > 0)
            {
                int acc = 0;
                for (int k = 0; k < 4 + p; k++)
                {
                    nb->waveClassShare[p][k] = std::max(1, static_cast<int>(static_cast<long long>(v[k]) * 1024 * (4 + p) / sum));
                    acc += nb->waveClassShare[p][k];
                }
                nb->waveClassShare[p][4 + p - 1] += 1024 * (4 + p) - acc;
                nb->waveClassShareFixed[p] = true;
            }
        }
    }
    if (const char* env = diagnosticsEnv("NBNXM_HIP_FEP_CONCURRENT"))
    {
        nb->fepConcurrent      = (std::atoi(env) != 0);
        nb->fepConcurrentFused = (std::atoi(env) == 2);
        nb->fepBehindFused     = (std::atoi(env) == 3);
    }
    if (const char* env = diagnosticsEnv("NBNXM_HIP_F_DOUBLE_BUFFER"))
    {
        nb->fDoubleBuffer = (std::atoi(env) != 0);
    }
    if (const char* env = diagnosticsEnv("NBNXM_HIP_PRUNE_MERGED"))
    {
        nb->pruneMerged = (std::atoi(env) != 0);
    }
    if (const char* env = diagnosticsEnv("NBNXM_HIP_FEP_MERGED"))
    {
        nb->fepMergedFused = (std::atoi(env) != 0);
    }
    if (const char* env = diagnosticsEnv("NBNXM_HIP_FEP_LIST_MERGED"))
    {
        nb->fepListMerged = (std::atoi(env) != 0);
    }
    if (const char* env = diagnosticsEnv("NBNXM_HIP_ENERGY_TAIL"))
    {
        nb->energyTail = std::max(0, std::min(c_energyTailCompiled, std::atoi(env)));
    }
    if (bFEP && nb->fepConcurrent)
    {
        for (int i = 0; i < (nb->bUseTwoStreams ? 2 : 1); i++)
        {
            int lo = 0, hi = 0;
            NBNXM_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            /* few, long-latency waves: give them priority over the throughput kernel they overlap with */
            NBNXM_HIP_CHECK(hipStreamCreateWithPriority(&nb->fepStreams[i].stream, hipStreamNonBlocking, hi));
            nb->fepStreams[i].owned = true;
            NBNXM_HIP_CHECK(hipEventCreateWithFlags(&nb->fepFork[i], hipEventDisableTiming));
            NBNXM_HIP_CHECK(hipEventCreateWithFlags(&nb->fepJoin[i], hipEventDisableTiming));
        }
    }
    if (nb->bUseTwoStreams)
    {
        nb->deviceStreams[1].init(nonLocalStream, true);
        NBNXM_HIP_CHECK(hipEventCreateWithFlags(&nb->nonlocal_done, hipEventDisableTiming));
        NBNXM_HIP_CHECK(hipEventCreateWithFlags(&nb->misc_ops_and_local_H2D_done, hipEventDisableTiming));
        NBNXM_HIP_CHECK(hipEventCreateWithFlags(&nb->nonlocalKernelDone, hipEventDisableTiming));
    }
    for (auto& t : nb->timers)
    {
        t.nb_k.init();
        t.fep_k.init();
        t.prune_k.init();
    }
    hipDeviceProp_t prop;
    int             dev = 0;
    NBNXM_HIP_CHECK(hipGetDevice(&dev));
    NBNXM_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    nb->numCUs = prop.multiProcessorCount;
    if (const char* env = diagnosticsEnv("NBNXM_HIP_WAVES_PER_BLOCK"))
    {
        const int w = std::atoi(env);
        NBNXM_ASSERT(w == 1 || w == 2 || w == 4, "NBNXM_HIP_WAVES_PER_BLOCK must be 1, 2 or 4");
        nb->nbWavesPerBlock = w;
    }

    if (const char* env = diagnosticsEnv("NBNXM_HIP_KEEP_COMB_KERNELS")) { nb->keepCombinationKernels = (std::atoi(env) != 0); }
    if (const char* env = diagnosticsEnv("NBNXM_HIP_KEEP_TAB_KERNELS")) { nb->keepTabulatedKernels = (std::atoi(env) != 0); }
    nb->debugLaunchShape = (diagnosticsEnv("NBNXM_HIP_DEBUG_LAUNCH_SHAPE") != nullptr);

    /* pinned staging (gpu_init :573-583) */
    auto pinned = [](float** p, size_t n) {
        NBNXM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(p), n * sizeof(float), hipHostMallocDefault));
        std::memset(*p, 0, n * sizeof(float));
    };
    /* scalar-output block, device and pinned mirror:
     * [eLJ, eElec, dvdlLJ, dvdlElec, eLJForeign[n+1], eElecForeign[n+1], dvdlLJForeign[n+1], dvdlElecForeign[n+1],
     *  pad to a 128-byte line, energySlots[c_numEnergySlots][c_energySlotStride]] */
    nb->numHeadScalars   = 4 + 4 * (n_lambda + 1);
    nb->slotOffset       = (nb->numHeadScalars + c_energySlotStride - 1) / c_energySlotStride * c_energySlotStride;
    nb->foreignSlotOffset = nb->slotOffset + c_numEnergySlots * c_energySlotStride;
    nb->foreignSlotStride = (4 * (n_lambda + 1) + c_energySlotStride - 1) / c_energySlotStride * c_energySlotStride;
    nb->numScalarOutputs  = nb->foreignSlotOffset + c_numForeignSlots * nb->foreignSlotStride;
    pinned(&nb->nbst.scalars, nb->numScalarOutputs);
    pinned(&nb->nbst.fShift, c_fshiftBlockFloats);
    nb->nbst.eLJ             = nb->nbst.scalars + 0;
    nb->nbst.eElec           = nb->nbst.scalars + 1;
    nb->nbst.dvdlLJ          = nb->nbst.scalars + 2;
    nb->nbst.dvdlElec        = nb->nbst.scalars + 3;
    nb->nbst.eLJForeign      = nb->nbst.scalars + 4;
    nb->nbst.eElecForeign    = nb->nbst.eLJForeign + (n_lambda + 1);
    nb->nbst.dvdlLJForeign   = nb->nbst.eElecForeign + (n_lambda + 1);
    nb->nbst.dvdlElecForeign = nb->nbst.dvdlLJForeign + (n_lambda + 1);
    nb->nbst.energySlots     = nb->nbst.scalars + nb->slotOffset;
    nb->nbst.foreignSlots    = nb->nbst.scalars + nb->foreignSlotOffset;

    /* initNbparam :421-489 */
    NBParamGpu* nbp = nb->nbparam;
    NBNXM_ASSERT(ic->elecType >= 0 && ic->elecType < NBNXM_ELEC_COUNT, "unknown electrostatics kernel type");
    nb->callerParams             = *ic;
    nb->callerParams.coulomb_tab = nullptr; /* (kept for the kernel pick only, see nbnxm_gpu_set_kernel_routing) */
    nbp->elecType   = kernelElecType(nb, ic);
    nbp->vdwType    = ic->vdwType;
    nbp->bFEP       = bFEP != 0;
    setCutoffParameters(nbp, ic);
    NBNXM_ASSERT(ic->elecType >= 0 && ic->elecType < NBNXM_ELEC_COUNT && ic->vdwType >= 0 && ic->vdwType < NBNXM_VDW_COUNT,
                 "unknown electrostatics / VdW kernel type");
    hipStream_t s = nb->deviceStreams[0].stream;
    allocateDeviceBuffer(&nbp->nbfp, static_cast<size_t>(numTypes) * numTypes);
    copyToDeviceBuffer(&nbp->nbfp, reinterpret_cast<const float2*>(nbfp), 0, static_cast<size_t>(numTypes) * numTypes, s, false);
    nb->nbfp_n = numTypes * numTypes;
    if (nbfp_comb != nullptr)
    {
        allocateDeviceBuffer(&nbp->nbfp_comb, numTypes);
        copyToDeviceBuffer(&nbp->nbfp_comb, reinterpret_cast<const float2*>(nbfp_comb), 0, numTypes, s, false);
        nb->nbfp_comb_n = numTypes;
    }
    uploadCoulombTable(nb, ic);
    uploadEwaldCorrectionTable(nb);
    if (nbp->elecType == NBNXM_ELEC_EWALD_ANA || nbp->elecType == NBNXM_ELEC_EWALD_ANA_TWIN)
    {
        /* domain of the fitted analytical correction (pme_corr_coeffs.h): (beta r)^2 <= 12 */
        NBNXM_ASSERT(ic->ewaldcoeff_q * ic->ewaldcoeff_q * ic->rcoulomb * ic->rcoulomb <= 12.0F,
                     "beta*rc > 3.46: outside the analytical Ewald correction's fitted range, use the tabulated kernel type");
    }
    allocateDeviceBuffer(&nbp->allLambdaCoul, std::max(1, n_lambda));
    allocateDeviceBuffer(&nbp->allLambdaVdw, std::max(1, n_lambda));

    /* initAtomdataFirst :297-339 */
    NBAtomDataGpu* ad = nb->atdat;
    ad->numTypes      = numTypes;
    allocateDeviceBuffer(&ad->shiftVec, c_numShiftVectors);
    /* all scalar outputs live in one block (layout above) with the shift forces behind them — primary shift forces + the cluster kernel's
     * accumulator slots (c_fshiftBlockFloats floats) —, so that one kernel clears them; two copies (NbnxmGpu::outputsBlock) */
    for (int k = 0; k < 2; k++)
    {
        allocateDeviceBuffer(&nb->outputsBlock[k], nb->numScalarOutputs + c_fshiftBlockFloats);
        clearDeviceBufferAsync(&nb->outputsBlock[k], 0, nb->numScalarOutputs + c_fshiftBlockFloats, s);
    }
    pointOutputsAt(nb, 0);
    nb->outputsSpareCleared = true;
    ad->foreignSlotStride = nb->foreignSlotStride;
    ad->shiftVecUploaded = false;
    NBNXM_HIP_CHECK(hipStreamSynchronize(s));
    return nb;
}

void nbnxm_gpu_free(NbnxmGpu* nb)
{
    if (nb == nullptr) { return; }
    for (auto& st : nb->deviceStreams)
    {
        if (st.stream) { (void)hipStreamSynchronize(st.stream); }
    }
    for (hipEvent_t& e : nb->slowCountReady)
    {
        if (e != nullptr) { (void)hipEventDestroy(e); }
        e = nullptr;
    }
    if (nb->h_listError != nullptr) { (void)hipHostFree(nb->h_listError); }
    nb->h_listError = nullptr;
    if (nb->listStagingFree != nullptr) { (void)hipEventDestroy(nb->listStagingFree); }
    nb->listStagingFree = nullptr;
    NBAtomDataGpu* ad  = nb->atdat;
    NBParamGpu*    nbp = nb->nbparam;
    freeDeviceBuffer(&ad->xq);
    freeDeviceBuffer(&ad->q4);
    freeDeviceBuffer(&ad->f);
    freeDeviceBuffer(&nb->fSpare);
    freeDeviceBuffer(&nb->outputsBlock[0]);
    freeDeviceBuffer(&nb->outputsBlock[1]);
    nb->scalarOutputs = nullptr;
    ad->fShift = nullptr; /* (part of outputsBlock) */
    freeDeviceBuffer(&ad->atomTypes);
    freeDeviceBuffer(&ad->ljComb);
    freeDeviceBuffer(&ad->atomTypes4);
    freeDeviceBuffer(&ad->shiftVec);
    freeDeviceBuffer(&ad->fepBits);
    freeDeviceBuffer(&nbp->nbfp);
    freeDeviceBuffer(&nbp->nbfp_comb);
    freeDeviceBuffer(&nbp->coulomb_tab);
    freeDeviceBuffer(&nbp->ewaldCorrTab);
    freeDeviceBuffer(&nbp->ewaldCorrTabFV);
    freeDeviceBuffer(&nb->atomIndices);
    freeDeviceBuffer(&nb->cell);
    freeDeviceBuffer(&nbp->allLambdaCoul);
    freeDeviceBuffer(const_cast<float2**>(&nbp->windowLambda));
    freeDeviceBuffer(&nb->atdat->windowSlots);
    freeDeviceBuffer(&nbp->allLambdaVdw);
    for (int i = 0; i < 2; i++)
    {
        if (nb->plist[i])
        {
            freeDeviceBuffer(&nb->plist[i]->sci);
            freeDeviceBuffer(&nb->plist[i]->cjPacked);
            freeDeviceBuffer(&nb->plist[i]->imask);
            freeDeviceBuffer(&nb->plist[i]->excl);
            freeDeviceBuffer(&nb->plist[i]->sciSorted);
            freeDeviceBuffer(&nb->plist[i]->groupWeight);
            freeDeviceBuffer(&nb->plist[i]->groupSlowMask);
            freeDeviceBuffer(&nb->plist[i]->slowPairs);
            freeDeviceBuffer(&nb->plist[i]->slowCount);
            freeDeviceBuffer(&nb->plist[i]->weightBlockSum);
            for (int p = 0; p < c_numWorkPartitions; p++)
            {
                freeDeviceBuffer(&nb->plist[i]->workRangeStart[p]);
                freeDeviceBuffer(&nb->plist[i]->workFirstSci[p]);
                freeDeviceBuffer(&nb->plist[i]->workDesc[p]);
                freeDeviceBuffer(&nb->plist[i]->workShare[p]);
                freeDeviceBuffer(&nb->plist[i]->workShareCum[p]);
            }
            delete nb->plist[i];
        }
        if (nb->feplist[i])
        {
            freeDeviceBuffer(&nb->feplist[i]->iinr);
            freeDeviceBuffer(&nb->feplist[i]->shift);
            freeDeviceBuffer(&nb->feplist[i]->jindex);
            freeDeviceBuffer(&nb->feplist[i]->jjnr);
            freeDeviceBuffer(&nb->feplist[i]->excl_fep);
            freeDeviceBuffer(&nb->feplist[i]->pairEntry);
            freeDeviceBuffer(&nb->feplist[i]->clItem);
            freeDeviceBuffer(&nb->feplist[i]->clListed);
            freeDeviceBuffer(&nb->feplist[i]->clIncl);
            delete nb->feplist[i];
        }
    }
    auto unpin = [](float* p) {
        if (p) { (void)hipHostFree(p); }
    };
    unpin(nb->nbst.scalars);
    unpin(nb->nbst.fShift);
    for (auto& t : nb->timers)
    {
        t.nb_k.destroy();
        t.fep_k.destroy();
        t.prune_k.destroy();
    }
    if (nb->nonlocal_done) { (void)hipEventDestroy(nb->nonlocal_done); }
    if (nb->misc_ops_and_local_H2D_done) { (void)hipEventDestroy(nb->misc_ops_and_local_H2D_done); }
    if (nb->nonlocalKernelDone) { (void)hipEventDestroy(nb->nonlocalKernelDone); }
    for (int i = 0; i < 2; i++)
    {
        if (nb->fepStreams[i].stream) { (void)hipStreamSynchronize(nb->fepStreams[i].stream); }
        nb->fepStreams[i].destroy();
        if (nb->fepFork[i]) { (void)hipEventDestroy(nb->fepFork[i]); }
        if (nb->fepJoin[i]) { (void)hipEventDestroy(nb->fepJoin[i]); }
    }
    nb->deviceStreams[0].destroy();
    nb->deviceStreams[1].destroy();
    delete nb->atdat;
    delete nb->nbparam;
    delete nb;
}

void nbnxm_gpu_set_softcore(NbnxmGpu* nb, int softcoreType, float gapsysScaleLinpointVdW, float gapsysScaleLinpointCoul,
                            float gapsysSigma6VdW)
{
    NBNXM_ASSERT(softcoreType == NBNXM_SOFTCORE_BEUTLER || softcoreType == NBNXM_SOFTCORE_GAPSYS, "unknown soft-core type");
    NBParamGpu* nbp         = nb->nbparam;
    nbp->softcoreType       = softcoreType;
    nbp->gapsysLinpointVdw  = gapsysScaleLinpointVdW;
    nbp->gapsysLinpointCoul = gapsysScaleLinpointCoul;
    nbp->gapsysSigma6Vdw    = gapsysSigma6VdW;
}

void nbnxm_gpu_copy_fepparams(NbnxmGpu* nb, int bFEP, float alpha_coul, float alpha_vdw,
                              int lam_power, float sc_sigma6_def, float sc_sigma6_min,
                              float lambda_q, float lambda_v, int n_lambda,
                              const double* all_lambda_coul, const double* all_lambda_vdw)
{
    if (!bFEP) { return; }
    NBNXM_ASSERT(n_lambda == nb->n_lambda, "n_lambda differs from the value given to gpu_init");
    NBNXM_ASSERT(lam_power == 1 || lam_power == 2, "sc-power must be 1 or 2");
    NBParamGpu* nbp    = nb->nbparam;
    nbp->bFEP          = true;
    nbp->alpha_coul    = alpha_coul;
    nbp->alpha_vdw     = alpha_vdw;
    nbp->lam_power     = lam_power;
    nbp->sc_sigma6     = sc_sigma6_def;
    nbp->sc_sigma6_min = sc_sigma6_min;
    nbp->lambda_q      = lambda_q;
    nbp->lambda_v      = lambda_v;
    if (n_lambda > 0)
    {
        NBNXM_ASSERT(all_lambda_coul && all_lambda_vdw, "foreign lambda arrays missing");
        std::vector<float> c(n_lambda), v(n_lambda);
        for (int i = 0; i < n_lambda; i++)
        {
            c[i] = static_cast<float>(all_lambda_coul[i]);
            v[i] = static_cast<float>(all_lambda_vdw[i]);
        }
        copyToDeviceBuffer(&nbp->allLambdaCoul, c.data(), 0, n_lambda, nb->deviceStreams[0].stream, false);
        copyToDeviceBuffer(&nbp->allLambdaVdw, v.data(), 0, n_lambda, nb->deviceStreams[0].stream, false);
    }
}

void nbnxm_gpu_set_kernel_routing(NbnxmGpu* nb, int keepTabulatedKernels, int keepCombinationKernels)
{
    nb->keepTabulatedKernels   = (keepTabulatedKernels != 0);
    nb->keepCombinationKernels = (keepCombinationKernels != 0);
    nb->nbparam->elecType      = kernelElecType(nb, &nb->callerParams);
}

void nbnxm_gpu_pme_loadbal_update_param(NbnxmGpu* nb, const nbnxm_interaction_params_t* ic)
{
    nb->callerParams            = *ic;
    nb->callerParams.coulomb_tab = nullptr; /* (the copy is kept for the kernel pick only: the caller's table pointer does not outlive the call) */
    nb->nbparam->elecType = kernelElecType(nb, ic);
    setCutoffParameters(nb->nbparam, ic);
    uploadCoulombTable(nb, ic);
    uploadEwaldCorrectionTable(nb);
}

void nbnxm_gpu_init_atomdata(NbnxmGpu* nb, int numAtoms, int numAtomsLocal, const int* type,
                             const float* lj_comb, const float* qA, const float* qB,
                             const int* typeA, const int* typeB, const float* lj_combA,
                             const float* lj_combB)
{
    NBAtomDataGpu* ad = nb->atdat;
    hipStream_t    s  = nb->deviceStreams[0].stream;
    NBNXM_ASSERT(numAtoms % c_clSize == 0, "the atom count must be a multiple of the cluster size");
    const bool useComb = (nb->nbparam->vdwType == NBNXM_VDW_CUT_COMB_GEOM || nb->nbparam->vdwType == NBNXM_VDW_CUT_COMB_LB);
    NBNXM_ASSERT(type != nullptr, "atom types missing");
    NBNXM_ASSERT(!useComb || lj_comb != nullptr, "combination-rule kernel selected without per-atom LJ parameters");

    if (numAtoms > ad->numAtomsAlloc)
    {
        const int nalloc = static_cast<int>(numAtoms * 1.2) + 1024;
        freeDeviceBuffer(&ad->f);
        freeDeviceBuffer(&ad->xq);
        freeDeviceBuffer(&ad->atomTypes);
        freeDeviceBuffer(&ad->ljComb);
        freeDeviceBuffer(&ad->q4);
        freeDeviceBuffer(&ad->atomTypes4);
        allocateDeviceBuffer(&ad->f, nalloc);
        allocateDeviceBuffer(&ad->xq, nalloc);
        allocateDeviceBuffer(&ad->atomTypes, nalloc);
        allocateDeviceBuffer(&ad->ljComb, nalloc);
        allocateDeviceBuffer(&ad->q4, nalloc);
        allocateDeviceBuffer(&ad->atomTypes4, nalloc);
        ad->numAtomsAlloc = nalloc;
        clearDeviceBufferAsync(&ad->f, 0, nalloc, s); /* first use: no stale forces */
        freeDeviceBuffer(&nb->fSpare);
        nb->fSpareAlloc   = 0;
        nb->fSpareCleared = false;
    }
    ad->numAtoms      = numAtoms;
    ad->numAtomsLocal = numAtomsLocal;
    nb->fSpareCleared = false; /* the spare force buffer was zeroed for the previous atom count */

    nb->h_atomTypes.resize(numAtoms);
    std::memcpy(nb->h_atomTypes.data, type, sizeof(int) * numAtoms);
    copyToDeviceBuffer(&ad->atomTypes, nb->h_atomTypes.data, 0, numAtoms, s, true);
    if (lj_comb != nullptr)
    {
        nb->h_ljComb.resize(numAtoms);
        std::memcpy(nb->h_ljComb.data, lj_comb, sizeof(float2) * numAtoms);
        copyToDeviceBuffer(&ad->ljComb, nb->h_ljComb.data, 0, numAtoms, s, true);
    }
    if (nb->nbparam->bFEP)
    {
        /* FEP part, :990-1036: pack qA/qB -> float4, typeA/typeB -> int4 */
        NBNXM_ASSERT(qA && qB && typeA && typeB, "FEP needs the A/B charges and types");
        nb->h_q4.resize(numAtoms);
        nb->h_atomTypes4.resize(numAtoms);
        /* (range check as one min / max reduction, packing as branch-free loops: the compiler vectorises both — per-atom asserts
         * inside the packing loop were most of this call's 0.2 ms at 96k atoms) */
        int tLo = 0, tHi = 0;
        for (int i = 0; i < numAtoms; i++)
        {
            tLo = std::min(tLo, std::min(typeA[i], typeB[i]));
            tHi = std::max(tHi, std::max(typeA[i], typeB[i]));
        }
        NBNXM_ASSERT(tLo >= 0 && tHi < ad->numTypes, "atom type out of range");
        float4* __restrict__ q4h = nb->h_q4.data;
        int4* __restrict__   t4h = nb->h_atomTypes4.data;
        for (int i = 0; i < numAtoms; i++) { q4h[i] = make_float4(qA[i], qB[i], 0.0F, 0.0F); }
        for (int i = 0; i < numAtoms; i++) { t4h[i] = make_int4(typeA[i], typeB[i], 0, 0); }
        copyToDeviceBuffer(&ad->q4, nb->h_q4.data, 0, numAtoms, s, true);
        copyToDeviceBuffer(&ad->atomTypes4, nb->h_atomTypes4.data, 0, numAtoms, s, true);
        /* lj_combA / lj_combB (NBAtomDataGpu::ljComb4 of the reference, nbnxm_fep_cuda_kernel.cuh:357-375) are not uploaded: the
         * perturbed pairs of every flavour take c6 / c12 of the A and B state from the type-pair table (nbfp[typeA], nbfp[typeB]),
         * which holds the numbers the combination rule would produce, so no kernel here reads per-atom A/B LJ parameters */
        (void)lj_combA;
        (void)lj_combB;
    }
    int lo = 0, hi = 0;
    for (int i = 0; i < numAtoms; i++)
    {
        lo = std::min(lo, type[i]);
        hi = std::max(hi, type[i]);
    }
    NBNXM_ASSERT(lo >= 0 && hi < ad->numTypes, "atom type out of range");
}

static void uploadPairlist(NbnxmGpu* nb, int iloc, int na_c, int nsci, const nbnxn_sci_t* sci, int ncjPacked, const nbnxn_cj_packed_t* cjPacked,
                           int nexcl, const nbnxn_excl_t* excl);

void nbnxm_gpu_set_merged_localities(NbnxmGpu* nb, int merged)
{
    NBNXM_ASSERT(nb->bUseTwoStreams || !merged, "an object with one locality has nothing to merge");
    nb->mergedLocalities = merged != 0;
}

int nbnxm_gpu_get_merged_localities(const NbnxmGpu* nb)
{
    return nb->mergedLocalities ? 1 : 0;
}

void nbnxm_gpu_init_pairlist(NbnxmGpu* nb, int iloc, int na_c, int nsci, const nbnxn_sci_t* sci,
                             int ncjPacked, const nbnxn_cj_packed_t* cjPacked, int nexcl,
                             const nbnxn_excl_t* excl)
{
    NBNXM_ASSERT(iloc == NBNXM_LOCAL || (iloc == NBNXM_NONLOCAL && nb->bUseTwoStreams), "bad locality");
    if (!nb->mergedLocalities)
    {
        uploadPairlist(nb, iloc, na_c, nsci, sci, ncjPacked, cjPacked, nexcl, excl);
        return;
    }
    /* Merged localities (nbnxm_gpu_set_merged_localities): the two lists of a domain become ONE device list, local entries first,
     * launched once per step behind the coordinate halo.  A wave of the cluster kernel owns a balanced range of the list and the
     * launch has one wave per wave slot, so two launches pay the start and the drain of the machine twice (measured: 81 us for the
     * two kernels of a domain with 96k home and 46k halo atoms, 62 us for the same pairs as one list).  The caller keeps the
     * reference's call sequence — gpu_init_pairlist(Local), then (NonLocal), pairlist.cpp:4450-4452 —: the local call uploads the
     * local list (a domain without a halo is complete with it), the non-local call uploads both as one; the non-local device list
     * stays empty, so its launches, prunes and copy-backs are no-ops. */
    static const nbnxn_excl_t allOnes = [] {
        nbnxn_excl_t e;
        for (unsigned& w : e.pair) { w = 0xFFFFFFFFU; }
        return e;
    }();
    if (iloc == NBNXM_LOCAL)
    {
        nb->mergeLocalSci.assign(sci, sci + nsci);
        nb->mergeLocalCj.assign(cjPacked, cjPacked + ncjPacked);
        nb->mergeLocalExcl.assign(excl, excl + nexcl);
        uploadPairlist(nb, NBNXM_LOCAL, na_c, nsci, sci, ncjPacked, cjPacked, nexcl, excl);
        uploadPairlist(nb, NBNXM_NONLOCAL, na_c, 0, nullptr, 0, nullptr, 1, &allOnes);
        nb->numMergedLocalGroups = ncjPacked;
        nb->mergeLocalIsFresh    = true;
        return;
    }
    NBNXM_ASSERT(nb->mergeLocalIsFresh, "merged localities: gpu_init_pairlist(NonLocal) must follow gpu_init_pairlist(Local) of the same search");
    nb->mergeLocalIsFresh = false;
    const int nsciL = static_cast<int>(nb->mergeLocalSci.size()), ncjL = static_cast<int>(nb->mergeLocalCj.size()),
              nexclL = static_cast<int>(nb->mergeLocalExcl.size());
    std::vector<nbnxn_sci_t>       mSci(nb->mergeLocalSci);
    std::vector<nbnxn_cj_packed_t> mCj(nb->mergeLocalCj);
    std::vector<nbnxn_excl_t>      mExcl(nb->mergeLocalExcl);
    mSci.reserve(nsciL + nsci);
    mCj.reserve(ncjL + ncjPacked);
    mExcl.reserve(nexclL + nexcl);
    for (int i = 0; i < nsci; i++)
    {
        nbnxn_sci_t e = sci[i];
        e.cjPackedBegin += ncjL;
        e.cjPackedEnd += ncjL;
        mSci.push_back(e);
    }
    for (int j = 0; j < ncjPacked; j++)
    {
        nbnxn_cj_packed_t g = cjPacked[j];
        for (auto& im : g.imei) { im.excl_ind += nexclL; } /* entry 0 of either list is the all-ones mask: it stays one */
        mCj.push_back(g);
    }
    mExcl.insert(mExcl.end(), excl, excl + nexcl);
    uploadPairlist(nb, NBNXM_LOCAL, na_c, nsciL + nsci, mSci.data(), ncjL + ncjPacked, mCj.data(), nexclL + nexcl, mExcl.data());
    uploadPairlist(nb, NBNXM_NONLOCAL, na_c, 0, nullptr, 0, nullptr, 1, &allOnes);
    nb->numMergedLocalGroups = ncjL;
}

static void uploadPairlist(NbnxmGpu* nb, int iloc, int na_c, int nsci, const nbnxn_sci_t* sci, int ncjPacked, const nbnxn_cj_packed_t* cjPacked,
                           int nexcl, const nbnxn_excl_t* excl)
{
    NBNXM_ASSERT(na_c == c_clSize, "the pair list cluster size does not match the kernels (8)");
    gpu_plist*  d = nb->plist[iloc];
    hipStream_t s = nb->deviceStreams[iloc].stream;
    const int numAtoms = nb->atdat->numAtoms;
#ifdef NBNXM_HOST_UPLOAD_TIMING /* diagnostics: host microseconds by section of this call */
    std::vector<std::chrono::steady_clock::time_point> tp_{ std::chrono::steady_clock::now() };
#define NBNXM_TP tp_.push_back(std::chrono::steady_clock::now());
#else
#define NBNXM_TP
#endif
    d->na_c = na_c;
    reallocateDeviceBuffer(&d->sci, nsci, &d->nsci, &d->sci_nalloc);
    reallocateDeviceBuffer(&d->cjPacked, ncjPacked, &d->ncjPacked, &d->cjPacked_nalloc);
    reallocateDeviceBuffer(&d->imask, static_cast<size_t>(ncjPacked) * NBNXM_GPU_CLUSTERPAIR_SPLIT, &d->nimask, &d->imask_nalloc);
    reallocateDeviceBuffer(&d->excl, nexcl, &d->nexcl, &d->excl_nalloc);
    /* The three arrays go up as DMAs from page-locked memory.  A caller that keeps its lists in pinned memory (the reference does:
     * HostVector with the pinning allocator, pairlist.h) is read in place; other memory is staged through the object's pinned buffers,
     * the larger two first, each DMA queued as soon as its staging copy is done so that it runs beside the next one (4 MB of staging
     * copies are 0.25 ms of a 0.38 ms upload on the 96k box).  The staging buffers are per object, not per locality: the upload is
     * complete — the stream synchronised, below — before they can be reused; that also ends the caller's obligation to keep its
     * arrays, whichever path was taken. */
    auto upload = [s](auto* deviceBuffer, auto& staging, const auto* src, int n) {
        if (n == 0) { return; }
        if (isPinnedHostMemory(src)) { copyToDeviceBuffer(deviceBuffer, src, 0, n, s, true); }
        else
        {
            staging.resize(n);
            std::memcpy(staging.data, src, sizeof(*src) * n);
            copyToDeviceBuffer(deviceBuffer, staging.data, 0, n, s, true);
        }
    };
    /* (the staging buffers of the previous upload — the other locality's, or the last search step's sorted entries — are free?) */
    if (nb->listStagingBusy)
    {
        NBNXM_HIP_CHECK(hipEventSynchronize(nb->listStagingFree));
        nb->listStagingBusy = false;
    }
    NBNXM_TP
    const bool inPlace = isPinnedHostMemory(cjPacked) && isPinnedHostMemory(excl) && isPinnedHostMemory(sci);
    NBNXM_TP
    upload(&d->cjPacked, nb->h_cjPacked, cjPacked, ncjPacked);
    upload(&d->excl, nb->h_excl, excl, nexcl);
    upload(&d->sci, nb->h_sci, sci, nsci);
    NBNXM_TP
    /* shape checks, while the DMAs run: an out-of-range index in the list would fault on the device.  The i-entries here (a failed
     * check ends the process before anything is launched on the list), the 60 k groups on the device */
    for (int i = 0; i < nsci; i++)
    {
        NBNXM_ASSERT(sci[i].cjPackedBegin >= 0 && sci[i].cjPackedEnd <= ncjPacked && sci[i].cjPackedBegin <= sci[i].cjPackedEnd,
                     "sci entry points outside cjPacked");
        NBNXM_ASSERT(sci[i].sci >= 0 && (sci[i].sci + 1) * c_superClSize <= numAtoms, "sci entry outside the atom range");
        NBNXM_ASSERT((sci[i].shift & NBNXM_CI_SHIFT_MASK) < c_numShiftVectors, "shift index out of range");
    }
    NBNXM_TP
    NBNXM_TP

    /* the i-entries ordered by their j-group range, for the work partition (empty entries first among equals) */
    {
        int dummy = 0;
        reallocateDeviceBuffer(&d->sciSorted, nsci, &dummy, &d->sciSorted_nalloc);
        /* (worked on in ordinary memory and copied to the pinned staging buffer once: the passes below over page-locked memory took
         * 140 us for 6 k entries) */
        std::vector<nbnxn_sci_t>& w = nb->sciWorkHost;
        NBNXM_TP
        w.assign(sci, sci + nsci);
        NBNXM_TP
        auto byGroupRange = [](const nbnxn_sci_t& a, const nbnxn_sci_t& b) {
            return a.cjPackedBegin != b.cjPackedBegin ? a.cjPackedBegin < b.cjPackedBegin : a.cjPackedEnd < b.cjPackedEnd;
        };
        /* (a list builder appends j-groups entry by entry: the entries usually come in this order already) */
        if (!std::is_sorted(w.begin(), w.end(), byGroupRange))
        {
            /* (a producer that concatenates per-thread lists: sorted as 64-bit keys {begin, end, index} while the three fit 21 bits
             * each — 6 k entries in 35 instead of 110 us —, with the comparator beyond) */
            if (ncjPacked < (1 << 21) && nsci < (1 << 21))
            {
                std::vector<unsigned long long>& keys = nb->sciSortKeys;
                keys.resize(nsci);
                for (int i = 0; i < nsci; i++)
                {
                    keys[i] = (static_cast<unsigned long long>(sci[i].cjPackedBegin) << 42) | (static_cast<unsigned long long>(sci[i].cjPackedEnd) << 21)
                              | static_cast<unsigned long long>(i);
                }
                /* (sorted runs, one per thread of the producer: merged pairwise, O(n log runs), instead of a full sort) */
                std::vector<int>& runs = nb->sciSortRuns;
                runs.clear();
                runs.push_back(0);
                for (int i = 1; i < nsci; i++)
                {
                    if (keys[i] < keys[i - 1]) { runs.push_back(i); }
                }
                runs.push_back(nsci);
                if (runs.size() > 66) { std::sort(keys.begin(), keys.end()); }
                else
                {
                    for (size_t width = 1; width + 1 < runs.size(); width *= 2)
                    {
                        for (size_t r = 0; r + width + 1 <= runs.size() - 1; r += 2 * width)
                        {
                            const size_t last = std::min(r + 2 * width, runs.size() - 1);
                            std::inplace_merge(keys.begin() + runs[r], keys.begin() + runs[r + width], keys.begin() + runs[last]);
                        }
                    }
                }
                for (int i = 0; i < nsci; i++) { w[i] = sci[keys[i] & ((1ULL << 21) - 1ULL)]; }
            }
            else { std::sort(w.begin(), w.end(), byGroupRange); }
        }
        /* A list builder that balances for GPUs by i-entry count (pairlist.cpp:2283-2400) cuts the j-list of one (super-cluster,
         * shift) into consecutive entries.  The cluster kernel balances by wave-slot ranges that cut through entries, and every
         * entry start costs it an i-side staging and a force reduction (19,551 instead of 2,637 entries on the 96k box: 83.6 vs
         * 61.6 us), so its own entry list joins such pieces again; list pruning keeps working on the caller's entries (d->sci). */
        int nWork = 0, prevEnd = 0;
        for (int i = 0; i < nsci; i++)
        {
            const nbnxn_sci_t e = w[i];
            if (e.cjPackedBegin == e.cjPackedEnd) { continue; }
            NBNXM_ASSERT(e.cjPackedBegin >= prevEnd, "the j-group ranges of two sci entries overlap");
            prevEnd = e.cjPackedEnd;
            if (nWork > 0 && w[nWork - 1].sci == e.sci && w[nWork - 1].shift == e.shift && w[nWork - 1].cjPackedEnd == e.cjPackedBegin)
            {
                w[nWork - 1].cjPackedEnd = e.cjPackedEnd;
            }
            else { w[nWork++] = e; }
        }
        NBNXM_TP
        nb->h_sciSorted.resize(nWork);
        if (nWork) { std::memcpy(nb->h_sciSorted.data, w.data(), sizeof(nbnxn_sci_t) * nWork); }
        NBNXM_TP
        d->nsciWork = nWork;
        copyToDeviceBuffer(&d->sciSorted, nb->h_sciSorted.data, 0, nWork, s, true);
    }
    if (ncjPacked > 0)
    {
        /* the j-side of the check runs on the device, behind the copies (nbnxmValidateListKernel) */
        if (nb->h_listError == nullptr)
        {
            NBNXM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&nb->h_listError), sizeof(int), hipHostMallocMapped));
            *nb->h_listError = 0;
            NBNXM_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&nb->d_listError), nb->h_listError, 0));
        }
        constexpr int c_validateBlock = 256;
        hipLaunchKernelGGL(nbnxmValidateListKernel, dim3((ncjPacked + c_validateBlock - 1) / c_validateBlock), dim3(c_validateBlock), 0, s, d->cjPacked,
                           ncjPacked, numAtoms / c_clSize, nexcl, nb->d_listError);
        NBNXM_HIP_CHECK(hipGetLastError());
    }
    /* Lists read in place from page-locked memory: no wait — the copies run beside what the host does next in its search step, and
     * the caller keeps the arrays unchanged until the list's stream has passed them, as the reference's caller does
     * (nbnxm_gpu_data_mgmt.cpp:706-735, GpuApiCallBehavior::Async from its pinned HostVectors); the one staging buffer this path uses
     * (the sorted entries) is guarded by an event until the next upload.  Staged lists: the upload is complete on return, the caller's
     * arrays are free. */
    NBNXM_TP
#ifdef NBNXM_HOST_UPLOAD_TIMING
    {
        std::fprintf(stderr, "uploadPairlist host us:");
        for (size_t k = 1; k < tp_.size(); k++) { std::fprintf(stderr, " %.1f", std::chrono::duration<double, std::micro>(tp_[k] - tp_[k - 1]).count()); }
        std::fprintf(stderr, "  (realloc+wait | pinned? | 3 copies queued | sci checks | validate kernel queued | realloc sorted | assign | sort+join | to pinned | copy queued)\n");
    }
#endif
    if (inPlace)
    {
        if (nb->listStagingFree == nullptr) { NBNXM_HIP_CHECK(hipEventCreateWithFlags(&nb->listStagingFree, hipEventDisableTiming)); }
        NBNXM_HIP_CHECK(hipEventRecord(nb->listStagingFree, s));
        nb->listStagingBusy = true;
    }
    else { NBNXM_HIP_CHECK(hipStreamSynchronize(s)); }
    {
        /* the first-pass prune cuts every entry into waves of a few groups: 4 groups per wave unless that makes more than 64 waves per entry */
        int longest = 0;
        for (int i = 0; i < nsci; i++) { longest = std::max(longest, sci[i].cjPackedEnd - sci[i].cjPackedBegin); }
        constexpr int c_pruneGroupsPerWave = 4, c_maxPruneWavesPerEntry = 64;
        d->pruneGroupsPerWave = std::max(c_pruneGroupsPerWave, (longest + c_maxPruneWavesPerEntry - 1) / c_maxPruneWavesPerEntry);
        d->pruneWavesPerEntry = std::max(1, (longest + d->pruneGroupsPerWave - 1) / d->pruneGroupsPerWave);
    }
    d->workRangesDirty        = true;
    d->slowListDirty          = true;
    d->haveFreshList          = true;
    d->firstPruneDone         = false;
    d->rollingPruningNumParts = 0;
    d->rollingPruningPart     = 0;
    d->pendingPrunePart       = -1;
    d->pendingPruneEntries    = 0;
    d->pruneCallsSinceRebalance = 0;
    nb->haveWork[iloc]        = nsci > 0;
}

void nbnxm_gpu_init_feppairlist(NbnxmGpu* nb, int iloc, int nri, const int* iinr, const int* shift,
                                const int* jindex, int nrj, const int* jjnr, const int* excl_fep,
                                int numAtomIndices, const int* atomIndices)
{
    NBNXM_ASSERT(iloc == NBNXM_LOCAL || (iloc == NBNXM_NONLOCAL && nb->bUseTwoStreams), "bad locality");
    NBNXM_ASSERT(nb->nbparam->bFEP, "FEP pair list given to a non-FEP object");
    /* merged localities: the non-local launch is a no-op (its device list is empty), so an atom-pair list of that locality would never
     * run; the fused mode, which needs no atom-pair list, is the one that goes with merged localities */
    NBNXM_ASSERT(!(nb->mergedLocalities && iloc == NBNXM_NONLOCAL && nri > 0),
                 "merged localities take the perturbed pairs from the cluster list (nbnxm_gpu_set_fep_mode(nb, 1)): no non-local atom-pair list");
    gpu_feplist* d = nb->feplist[iloc];
    hipStream_t  s = nb->deviceStreams[iloc].stream;
    const int    numAtoms = nb->atdat->numAtoms;
    NBNXM_ASSERT(nri == 0 || (jindex[0] == 0 && jindex[nri] == nrj), "FEP list jindex does not cover jjnr");

#ifdef NBNXM_HOST_UPLOAD_TIMING
    std::vector<std::chrono::steady_clock::time_point> tq_{ std::chrono::steady_clock::now() };
#define NBNXM_TQ tq_.push_back(std::chrono::steady_clock::now());
#else
#define NBNXM_TQ
#endif
    /* topology id -> grid index (inverse of gridSet.atomIndices(), :766-790) */
    std::vector<int>& inverse = nb->fepInverse; /* (kept with the object: no allocation per search step) */
    inverse.clear();
    if (atomIndices != nullptr)
    {
        int maxId = -1;
        for (int g = 0; g < numAtomIndices; g++) { maxId = std::max(maxId, atomIndices[g]); }
        inverse.assign(maxId + 1, -1);
        for (int g = 0; g < numAtomIndices; g++)
        {
            if (atomIndices[g] >= 0) { inverse[atomIndices[g]] = g; }
        }
    }
    auto toGrid = [&](int a) -> int {
        int g = a;
        if (atomIndices != nullptr)
        {
            NBNXM_ASSERT(a >= 0 && a < static_cast<int>(inverse.size()) && inverse[a] >= 0, "FEP list atom is not on the grid");
            g = inverse[a];
        }
        NBNXM_ASSERT(g >= 0 && g < numAtoms, "FEP list atom outside the atom range");
        return g;
    };
    NBNXM_TQ
    nb->h_iinr.resize(nri);
    nb->h_shift.resize(nri);
    nb->h_jindex.resize(nri + 1);
    nb->h_jjnr.resize(nrj);
    nb->h_pairEntry.resize(nrj);
    nb->h_exclFep.resize(nrj);
    nb->h_jindex.data[0] = 0;
    for (int n = 0; n < nri; n++)
    {
        nb->h_iinr.data[n]       = toGrid(iinr[n]);
        nb->h_shift.data[n]      = shift[n];
        nb->h_jindex.data[n + 1] = jindex[n + 1];
        NBNXM_ASSERT(shift[n] >= 0 && shift[n] < c_numShiftVectors, "shift index out of range");
        NBNXM_ASSERT(jindex[n + 1] >= jindex[n], "jindex must not decrease");
        for (int k = jindex[n]; k < jindex[n + 1]; k++)
        {
            nb->h_jjnr.data[k]      = toGrid(jjnr[k]);
            nb->h_pairEntry.data[k] = n;
            nb->h_exclFep.data[k]   = (excl_fep != nullptr) ? excl_fep[k] : 1;
        }
    }
    NBNXM_TQ
    reallocateDeviceBuffer(&d->iinr, nri, &d->nri, &d->maxnri);
    reallocateDeviceBuffer(&d->shift, nri, &d->nshift, &d->maxnshift);
    reallocateDeviceBuffer(&d->jindex, nri + 1, &d->njidx, &d->maxnjidx);
    reallocateDeviceBuffer(&d->jjnr, nrj, &d->nrj, &d->maxnrj);
    reallocateDeviceBuffer(&d->excl_fep, nrj, &d->nexcl, &d->maxnexcl);
    int nPairEntry = 0;
    reallocateDeviceBuffer(&d->pairEntry, nrj, &nPairEntry, &d->pairEntry_nalloc);
    copyToDeviceBuffer(&d->iinr, nb->h_iinr.data, 0, nri, s, true);
    copyToDeviceBuffer(&d->shift, nb->h_shift.data, 0, nri, s, true);
    copyToDeviceBuffer(&d->jindex, nb->h_jindex.data, 0, nri + 1, s, true);
    copyToDeviceBuffer(&d->jjnr, nb->h_jjnr.data, 0, nrj, s, true);
    copyToDeviceBuffer(&d->excl_fep, nb->h_exclFep.data, 0, nrj, s, true);
    copyToDeviceBuffer(&d->pairEntry, nb->h_pairEntry.data, 0, nrj, s, true);

    NBNXM_TQ
    /* the same list regrouped by (i-cluster, j-cluster, shift): what the trailing workgroups of the cluster kernel evaluate, one wave per
     * item (gpu_feplist::clItem; fepListClusterItem) */
    {
        /* (no hashing: the pairs of one (i-cluster, shift) come in runs — the i-atoms of a cluster are consecutive i-entries — so a
         * table indexed by the j-cluster, valid for the current run only, finds the item; a run that comes back later simply opens
         * new items for its j-clusters, which is as good.  32.6 k pairs: 0.3 ms with std::unordered_map, the largest part of this call) */
        std::vector<int4>  items;
        std::vector<uint2> listed, incl;
        items.reserve(static_cast<size_t>(nrj) / 8 + 16);
        listed.reserve(items.capacity());
        incl.reserve(items.capacity());
        const int        numClusters = numAtoms / c_clSize + 1;
        std::vector<int> itemOfCj(numClusters, -1), runOfCj(numClusters, -1);
        int              run = -1;
        long long        runKey = -1;
        for (int n = 0; n < nri; n++)
        {
            const int       ai = nb->h_iinr.data[n], ci = ai / c_clSize, sh = nb->h_shift.data[n];
            const long long key = static_cast<long long>(ci) * c_numShiftVectors + sh;
            if (key != runKey)
            {
                runKey = key;
                run++;
            }
            for (int k = nb->h_jindex.data[n]; k < nb->h_jindex.data[n + 1]; k++)
            {
                const int aj = nb->h_jjnr.data[k], cj = aj / c_clSize;
                int       idx;
                if (runOfCj[cj] != run)
                {
                    idx          = static_cast<int>(items.size());
                    runOfCj[cj]  = run;
                    itemOfCj[cj] = idx;
                    items.push_back(make_int4(ci, cj, sh, 0));
                    listed.push_back(make_uint2(0U, 0U));
                    incl.push_back(make_uint2(0U, 0U));
                }
                else { idx = itemOfCj[cj]; }
                const unsigned bit = static_cast<unsigned>((aj % c_clSize) * c_clSize + ai % c_clSize); /* the lane tidxj * 8 + tidxi */
                unsigned&      lw  = (bit < 32U) ? listed[idx].x : listed[idx].y;
                unsigned&      iw  = (bit < 32U) ? incl[idx].x : incl[idx].y;
                NBNXM_ASSERT((lw & (1U << (bit & 31U))) == 0U, "the FEP list holds an atom pair twice");
                lw |= 1U << (bit & 31U);
                if (nb->h_exclFep.data[k] != 0) { iw |= 1U << (bit & 31U); }
            }
        }
        NBNXM_TQ
        const int numItems = static_cast<int>(items.size());
        if (numItems > d->clItem_nalloc)
        {
            freeDeviceBuffer(&d->clItem);
            freeDeviceBuffer(&d->clListed);
            freeDeviceBuffer(&d->clIncl);
            d->clItem_nalloc = static_cast<int>(numItems * 1.2) + 64;
            allocateDeviceBuffer(&d->clItem, d->clItem_nalloc);
            allocateDeviceBuffer(&d->clListed, d->clItem_nalloc);
            allocateDeviceBuffer(&d->clIncl, d->clItem_nalloc);
        }
        d->numClusterItems = numItems;
        if (numItems > 0)
        {
            /* (through the object's pinned buffers, on the list's stream: three blocking copies were 50 us of this call) */
            nb->h_clItem.resize(numItems);
            nb->h_clListed.resize(numItems);
            nb->h_clIncl.resize(numItems);
            std::memcpy(nb->h_clItem.data, items.data(), sizeof(int4) * numItems);
            std::memcpy(nb->h_clListed.data, listed.data(), sizeof(uint2) * numItems);
            std::memcpy(nb->h_clIncl.data, incl.data(), sizeof(uint2) * numItems);
            copyToDeviceBuffer(&d->clItem, nb->h_clItem.data, 0, numItems, s, true);
            copyToDeviceBuffer(&d->clListed, nb->h_clListed.data, 0, numItems, s, true);
            copyToDeviceBuffer(&d->clIncl, nb->h_clIncl.data, 0, numItems, s, true);
        }
    }
    NBNXM_TQ
    NBNXM_HIP_CHECK(hipStreamSynchronize(s));
    NBNXM_TQ
#ifdef NBNXM_HOST_UPLOAD_TIMING
    {
        std::fprintf(stderr, "init_feppairlist host us:");
        for (size_t k = 1; k < tq_.size(); k++) { std::fprintf(stderr, " %.1f", std::chrono::duration<double, std::micro>(tq_[k] - tq_[k - 1]).count()); }
        std::fprintf(stderr, "  (inverse map | remap | 6 copies queued | regroup | 3 copies queued | wait)\n");
    }
#endif
}

void nbnxm_gpu_init_fep_cluster_bits(NbnxmGpu* nb, int numClusters, const unsigned char* fepBits)
{
    NBAtomDataGpu* ad = nb->atdat;
    NBNXM_ASSERT(numClusters * c_clSize == ad->numAtoms, "fepBits must hold one byte per 8-atom cluster");
    NBNXM_ASSERT(numClusters % c_numClPerSupercl == 0, "cluster count must be a multiple of 8");
    if (numClusters > nb->fepBits_nalloc)
    {
        freeDeviceBuffer(&ad->fepBits);
        nb->fepBits_nalloc = static_cast<int>(numClusters * 1.2) + 64;
        allocateDeviceBuffer(&ad->fepBits, nb->fepBits_nalloc);
    }
    ad->numClusters = numClusters;
    nb->h_fepBits.resize(numClusters);
    std::memcpy(nb->h_fepBits.data, fepBits, numClusters);
    copyToDeviceBuffer(&ad->fepBits, nb->h_fepBits.data, 0, numClusters, nb->deviceStreams[0].stream, true);
    for (gpu_plist* pl : nb->plist)
    {
        if (pl)
        {
            pl->workRangesDirty = true;
            pl->slowListDirty   = true;
        }
    }
}

void nbnxm_gpu_set_fep_mode(NbnxmGpu* nb, int fused)
{
    NBNXM_ASSERT(!fused || nb->atdat->fepBits != nullptr, "fused FEP mode needs nbnxm_gpu_init_fep_cluster_bits first");
    const bool changed = (nb->fusedFep != (fused != 0));
    nb->fusedFep       = fused != 0;
    for (gpu_plist* pl : nb->plist)
    {
        if (pl)
        {
            pl->workRangesDirty = true;
            pl->slowListDirty   = true;
            /* the default shares of the ranges depend on the mode (updateWorkPartition); they survive everything else — in particular
             * the perturbed-atom bits of every search step, which used to drop them: two allocations and two blocking copies per search */
            if (changed) { pl->workShareCount[0] = pl->workShareCount[1] = -1; }
        }
    }
}

void nbnxm_gpu_upload_shiftvec(NbnxmGpu* nb, const float* shift_vec)
{
    nb->h_shiftVec.resize(3 * c_numShiftVectors);
    std::memcpy(nb->h_shiftVec.data, shift_vec, sizeof(float) * 3 * c_numShiftVectors);
    copyToDeviceBuffer(&nb->atdat->shiftVec, reinterpret_cast<const float3*>(nb->h_shiftVec.data), 0, c_numShiftVectors,
                       nb->deviceStreams[0].stream, true);
    nb->atdat->shiftVecUploaded = true;
}

void nbnxm_gpu_copy_xq_to_gpu(NbnxmGpu* nb, const float* xq, int atomLocality)
{
    NBAtomDataGpu* ad   = nb->atdat;
    const int      iloc = atomLocality;
    NBNXM_ASSERT(iloc == NBNXM_LOCAL || (iloc == NBNXM_NONLOCAL && nb->bUseTwoStreams), "bad locality");
    hipStream_t s = nb->deviceStreams[iloc].stream;
    /* local: atoms [0, numAtomsLocal), non-local: the rest (nbnxm_gpu.h:93-96) */
    const int begin = (iloc == NBNXM_LOCAL) ? 0 : ad->numAtomsLocal;
    const int count = (iloc == NBNXM_LOCAL) ? ad->numAtomsLocal : ad->numAtoms - ad->numAtomsLocal;
    if (count > 0)
    {
        NBNXM_HIP_CHECK(hipMemcpyAsync(ad->xq + begin, xq + 4 * static_cast<size_t>(begin), sizeof(float4) * count,
                                       hipMemcpyHostToDevice, s));
    }
    if (iloc == NBNXM_LOCAL && nb->bUseTwoStreams)
    {
        NBNXM_HIP_CHECK(hipEventRecord(nb->misc_ops_and_local_H2D_done, s));
    }
}

void nbnxm_gpu_clear_outputs(NbnxmGpu* nb, int computeVirial)
{
    NBAtomDataGpu* ad = nb->atdat;
    hipStream_t    s  = nb->deviceStreams[0].stream;
    /* A.4: the reference clears E / dV/dl only on virial steps; they are accumulated with atomics, so they are cleared on every
     * call that follows a launch which wrote them (one launch for everything; none at all when the forces come from the swap). */
    const int numFloats = 3 * ad->numAtoms;
    int       numFloat4 = numFloats / 4;
    int       numTail   = numFloats - 4 * numFloat4;
    if (nb->fDoubleBuffer && nb->fSpareCleared)
    {
        /* the last force-only kernel has zeroed the other force buffer in its trailing workgroups: swap */
        std::swap(ad->f, nb->fSpare);
        nb->fSpareCleared = false;
        numFloat4         = 0;
        numTail           = 0;
    }
    int numFshift = computeVirial ? c_fshiftBlockFloats : 0;
    if (nb->fDoubleBuffer && nb->outputsDoubleBuffer && nb->outputsSpareCleared && nb->numWindows == 0 && (nb->scalarsDirty || (computeVirial && nb->fshiftDirty)))
    {
        /* ... and the other copy of the scalar outputs and shift forces: the same swap (results of the last step have been copied back by
         * now: the copy-back and this call are on the same stream, or the caller has waited for the step) */
        pointOutputsAt(nb, 1 - nb->outputsActive);
        nb->outputsSpareCleared = false;
        nb->scalarsDirty        = false;
        nb->fshiftDirty         = false;
        numFshift               = 0;
    }
    const int numScalars = nb->scalarsDirty ? nb->numScalarOutputs : 0;
    const int numWindow  = nb->scalarsDirty ? nb->numWindows * ad->windowSlotStride : 0;
    nb->scalarsDirty     = false;
    if (numFshift > 0) { nb->fshiftDirty = false; }
    if (numFloat4 == 0 && numTail == 0 && numScalars == 0 && numWindow == 0 && numFshift == 0) { return; } /* nothing to clear */
    /* (sized by the largest of the arrays: after a swap of the force buffers only the few thousand scalars are left) */
    const int numMost = std::max(std::max(numFloat4, numScalars), std::max(numFshift, numWindow));
    const int nblock  = std::max(1, std::min(2048, (numMost + 255) / 256));
    hipLaunchKernelGGL(nbnxmClearOutputsKernel, dim3(nblock), dim3(256), 0, s, reinterpret_cast<float4*>(ad->f), numFloat4,
                       reinterpret_cast<float*>(ad->f) + 4 * static_cast<size_t>(numFloat4), numTail, nb->scalarOutputs, numScalars,
                       reinterpret_cast<float*>(ad->fShift), numFshift, ad->windowSlots, numWindow);
    NBNXM_HIP_CHECK(hipGetLastError());
}

/* uploads the shares of the ranges of partition p and their normalised running sum (synchronous: rare) */
static void setWorkShares(gpu_plist* d, int p, const float* share, int n, hipStream_t s)
{
    if (d->workShareCount[p] != n)
    {
        freeDeviceBuffer(&d->workShare[p]);
        freeDeviceBuffer(&d->workShareCum[p]);
        allocateDeviceBuffer(&d->workShare[p], n);
        allocateDeviceBuffer(&d->workShareCum[p], n + 1);
        d->workShareCount[p] = n;
    }
    double sum = 0;
    for (int r = 0; r < n; r++) { sum += share[r]; }
    std::vector<float> norm(n), cum(n + 1);
    double             run = 0;
    for (int r = 0; r < n; r++)
    {
        norm[r] = static_cast<float>(share[r] * n / sum);
        cum[r]  = static_cast<float>(run / sum);
        run += share[r];
    }
    cum[n] = 1.0F;
    NBNXM_HIP_CHECK(hipStreamSynchronize(s));
    NBNXM_HIP_CHECK(hipMemcpy(d->workShare[p], norm.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    NBNXM_HIP_CHECK(hipMemcpy(d->workShareCum[p], cum.data(), sizeof(float) * (n + 1), hipMemcpyHostToDevice));
    d->workRangesDirty = true;
}

/* the count of perturbed cluster pairs of the current list, if its copy has arrived (updateWorkPartition queued it) */
static void pickUpSlowCount(NbnxmGpu* nb, int iloc)
{
    gpu_plist* d = nb->plist[iloc];
    if (!d->slowCountPending) { return; }
    if (hipEventQuery(nb->slowCountReady[iloc]) != hipSuccess)
    {
        (void)hipGetLastError(); /* "not ready" is an answer, not an error for the launch checks that follow */
        return;
    }
    d->numSlowPairs     = nb->h_slowCount.data[2 * iloc];
    d->numSlowHeavy     = std::min(nb->h_slowCount.data[2 * iloc + 1], d->numSlowPairs);
    d->slowCountPending = false;
    d->slowCountKnown   = true;
    NBNXM_ASSERT(d->numSlowPairs <= d->slowPairs_nalloc,
                 "more perturbed cluster pairs than the fused mode provides for (use the atom-pair list mode for large perturbed regions)");
}

/* (Re)computes the work partition of a list on its stream; cheap (three launches over ncjPacked ints). */
static void updateWorkPartition(NbnxmGpu* nb, int iloc)
{
    gpu_plist*  d = nb->plist[iloc];
    hipStream_t s = nb->deviceStreams[iloc].stream;
    d->workRangesDirty = false;
#if defined(NBNXM_WAVE_TIMELINE) || defined(NBNXM_BLOCK_STATS)
    if (d->debugTimeline == nullptr)
    {
        /* (second half: four more time stamps per range wave, the steps of its prologue) */
        allocateDeviceBuffer(&d->debugTimeline, 12 * 16384);
        NBNXM_HIP_CHECK(hipMemset(d->debugTimeline, 0, sizeof(unsigned long long) * 12 * 16384));
    }
#endif
    if (d->nsci == 0 || d->ncjPacked == 0)
    {
        for (int p = 0; p < c_numWorkPartitions; p++) { d->numWorkRanges[p] = 0; }
        return;
    }
    const int numBlocks = (d->ncjPacked + c_workBlockSize - 1) / c_workBlockSize;
    int       dummy     = 0;
    const int oldAlloc = d->groupWeight_nalloc;
    reallocateDeviceBuffer(&d->groupWeight, d->ncjPacked, &dummy, &d->groupWeight_nalloc);
    if (d->groupWeight_nalloc != oldAlloc)
    {
        freeDeviceBuffer(&d->groupSlowMask);
        freeDeviceBuffer(&d->slowPairs);
        d->slowPairs_nalloc = std::max(8192, 2 * d->groupWeight_nalloc); /* a ligand-sized region has a few thousand */
        allocateDeviceBuffer(&d->groupSlowMask, d->groupWeight_nalloc);

        allocateDeviceBuffer(&d->slowPairs, d->slowPairs_nalloc);
        d->slowListDirty = true;
    }
    if (d->slowCount == nullptr)
    {
        allocateDeviceBuffer(&d->slowCount, 2);
        clearDeviceBufferAsync(&d->slowCount, 0, 2, s);
    }
    reallocateDeviceBuffer(&d->weightBlockSum, numBlocks + 1, &dummy, &d->weightBlockSum_nalloc);
    const bool fused     = nb->fusedFep && nb->nbparam->bFEP && nb->atdat->fepBits != nullptr;
    const bool buildSlow = fused && d->slowListDirty;
    if (buildSlow) { clearDeviceBufferAsync(&d->slowCount, 0, 1, s); }
    /* a list that has been through its first prune: the working masks are inner-pruned, the outer-pruned ones are in d->imask */
    const unsigned* outerMask = (!d->haveFreshList && d->firstPruneDone) ? d->imask : nullptr;
    /* the cost model (nbnxm_work_partition.h), in units of 1/8 cluster pair */
    NbWorkWeights weights = { c_weightPair, c_weightSlot, c_weightGroup, c_weightEmptyGroup, c_weightEntry };
    /* Short lists (the rule of the force flavour's partition below: under 4.5 packed groups per wave slot — boxes up to ~40k atoms, the
     * domains of a decomposed box): a wave has three or four groups and one to four pieces, it is bound by the LATENCY of its own
     * chain, not by its SIMD's issue rate, and the last waves of a launch are the ones with many groups and pieces.  Swept on the
     * 24k-atom box (tools/gpu_r4h.sh, slot / group / entry): 4 / 16 / 128 20.5 - 20.9 us, 0 / 48 / 200 and 4 / 48 / 200 18.5 - 18.8 us,
     * 0 / 64 / 260 19.0, 0 / 48 / 260 19.7; 12k atoms flat (16.8 - 17.0 us); 48k and 96k atoms — long lists — flat or best at 4 / 16 / 128. */
    if (2 * static_cast<long long>(d->ncjPacked) < 9LL * nb->numSimds * 5 && nb->numWorkRangesOverride <= 0)
    {
        weights.group = weights.emptyGroup = c_weightGroupShortList;
        weights.entry = c_weightEntryShortList;
    }
    if (nb->workWeightsOverride[0] >= 0)
    {
        weights.slot  = nb->workWeightsOverride[0];
        weights.group = weights.emptyGroup = nb->workWeightsOverride[1];
        weights.entry = nb->workWeightsOverride[2];
    }
    hipLaunchKernelGGL(nbnxmWorkWeightKernel, dim3(numBlocks), dim3(c_workBlockSize), 0, s, d->cjPacked, d->ncjPacked, d->sciSorted,
                       d->nsciWork, fused ? nb->atdat->fepBits : nullptr, buildSlow ? 1 : 0, outerMask, d->groupSlowMask, d->slowPairs,
                       d->slowPairs_nalloc, d->slowCount, d->groupWeight, d->weightBlockSum, weights);
    if (buildSlow && fused)
    {
        /* what the first pass has counted is the number of heavy pairs: kept beside the count (the kernels split those over several waves
         * on dH/dlambda steps) */
        NBNXM_HIP_CHECK(hipMemcpyAsync(d->slowCount + 1, d->slowCount, sizeof(int), hipMemcpyDeviceToDevice, s));
        /* the second pass of the slow-pair list: the light pairs behind the heavy ones (weights and masks: the same values again) */
        hipLaunchKernelGGL(nbnxmWorkWeightKernel, dim3(numBlocks), dim3(c_workBlockSize), 0, s, d->cjPacked, d->ncjPacked, d->sciSorted,
                           d->nsciWork, nb->atdat->fepBits, 2, outerMask, d->groupSlowMask, d->slowPairs, d->slowPairs_nalloc,
                           d->slowCount, d->groupWeight, d->weightBlockSum, weights);
    }
    hipLaunchKernelGGL(nbnxmWorkScanKernel, dim3(1), dim3(c_workBlockSize), 0, s, d->weightBlockSum, numBlocks);
    NBNXM_HIP_CHECK(hipGetLastError());
    if (buildSlow)
    {
        /* The host only needs the count to SIZE launches (the kernels stride over the device's count).  The first list of an object
         * waits for it; later lists queue the copy, size their first launches from the previous count plus a margin, and pick the
         * exact figure up when it has arrived (pickUpSlowCount): no host round trip in a search step. */
        if (nb->h_slowCount.size < 4) { nb->h_slowCount.resize(4); } /* per locality: the count, and how many of them are heavy */
        if (nb->slowCountReady[iloc] == nullptr) { NBNXM_HIP_CHECK(hipEventCreateWithFlags(&nb->slowCountReady[iloc], hipEventDisableTiming)); }
        NBNXM_HIP_CHECK(hipMemcpyAsync(nb->h_slowCount.data + 2 * iloc, d->slowCount, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        NBNXM_HIP_CHECK(hipEventRecord(nb->slowCountReady[iloc], s));
        d->slowCountPending = true;
        d->slowListDirty    = false;
        if (!d->slowCountKnown) { NBNXM_HIP_CHECK(hipStreamSynchronize(s)); }
        else { d->numSlowPairs = std::min(d->slowPairs_nalloc, std::max(256, d->numSlowPairs + d->numSlowPairs / 4 + 64)); }
        pickUpSlowCount(nb, iloc);
    }
    if (!fused)
    {
        d->numSlowPairs     = 0;
        d->numSlowHeavy     = 0;
        d->slowCountPending = false;
    }

    WorkPartitionOut out[c_numWorkPartitions];
    for (int p = 0; p < c_numWorkPartitions; p++)
    {
        const int pw    = (p == c_partitionForce) ? 1 : 0; /* which of the two sets of age-class shares */
        const int slots = nb->numSimds * workPartitionWaves(p);
        /* a launch in two parts (the local list of a decomposed run, force flavour): two sets of one range per wave slot, the first
         * set holding localPartFraction of the weight */
        const bool twoParts = (iloc == NBNXM_LOCAL && nb->localLaunchParts == 2 && nb->numWorkRangesOverride <= 0
                               && d->ncjPacked / nb->minGroupsPerWave >= 2 * slots);
        const int  want     = twoParts ? 2 * slots : slots;
        d->workParts[p]     = twoParts ? 2 : 1;
        d->numWorkRanges[p] = std::max(1, std::min(want, d->ncjPacked / nb->minGroupsPerWave));
        /* Short lists (under 4.5 packed groups per wave slot: boxes up to ~40k atoms, or the domains of a decomposed one): the
         * five-waves-per-SIMD partition cuts ranges of one to three groups, and every range pays a wave's prologue and an i-entry
         * start; four ranges per SIMD are faster there (12k atoms 0.0203 -> 0.0176 ms per force step, 24k 0.0221 -> 0.0208, 48k equal,
         * 96k 0.0510 -> 0.0556 the other way; tools/gpu_ranges_sizes.sh).  Equal shares (the age-class shares belong to five waves). */
        /* (the four-waves-per-SIMD partition of the energy flavours: three ranges per SIMD under 2.5 groups per slot — 12k atoms, energy
         * step 0.0270 -> 0.0232 ms; 24k atoms the other way, 0.0297 -> 0.0333) */
        const bool shortList = !twoParts && nb->numWorkRangesOverride <= 0
                               && ((p == c_partitionForce && 2 * static_cast<long long>(d->ncjPacked) < 9LL * slots)
                                   || (p != c_partitionForce && 2 * static_cast<long long>(d->ncjPacked) < 5LL * slots));
        if (shortList) { d->numWorkRanges[p] = std::max(1, std::min(nb->numSimds * (workPartitionWaves(p) - 1), d->ncjPacked / nb->minGroupsPerWave)); }
        /* (energy steps of a 12k-atom box — 7.5 k groups —: 2.5 ranges per SIMD, 0.0214 -> 0.0193 ms; dH/dlambda steps and the 3k box: flat) */
        if (shortList && p == c_partitionEnergy && nb->fusedFep && d->ncjPacked >= 5 * nb->numSimds && nb->numWorkRangesOverride <= 0)
        {
            d->numWorkRanges[p] = nb->numSimds * 5 / 2;
        }
        /* Energy flavours, up to 6 packed groups per wave slot (24k atoms: 3.7; 48k: 7.3): fewer ranges than wave slots.  The trailing
         * workgroups — perturbed pairs with their energies, on dH/dlambda steps at every foreign lambda — are a fifth to a half of such a
         * kernel, and with a range in every slot they only start when range waves retire; with 15 of 16 (dH/dlambda: 3 of 4) slots taken they
         * run beside the ranges from the start.  24k atoms: energy step 0.0247 -> 0.0231 ms, dH/dlambda step 0.0325 -> 0.0257 ms; at 96k
         * atoms and up fewer ranges only lose (the ranges' age classes need one range per slot). */
        const bool leaveSlots = !shortList && !twoParts && p != c_partitionForce && nb->fusedFep && nb->numWorkRangesOverride <= 0
                                && static_cast<long long>(d->ncjPacked) < 6LL * slots && d->numWorkRanges[p] == want;
        if (leaveSlots) { d->numWorkRanges[p] = (p == c_partitionDhdl) ? want * 3 / 4 : want * 15 / 16; }
        if (nb->numWorkRangesOverride > 0) { d->numWorkRanges[p] = std::min(nb->numWorkRangesOverride, d->ncjPacked); }
        if (p != c_partitionForce && nb->numWorkRangesEnergy > 0) { d->numWorkRanges[p] = std::max(1, std::min(nb->numWorkRangesEnergy, d->ncjPacked / nb->minGroupsPerWave)); }
        reallocateDeviceBuffer(&d->workRangeStart[p], d->numWorkRanges[p] + 1, &dummy, &d->work_nalloc[p]);
        out[p].numRanges = d->numWorkRanges[p];
        /* shares: only for the launch they are meant for, one wave per slot of every SIMD; they start from the age classes
         * (the waves of a SIMD are dispatched numRanges / classes apart) and survive new lists of the same size */
        const float fraction = twoParts ? nb->localPartFraction : 1.0F;
        /* how long the ranges are, in sixteenths between the short-range and the long-range shares of the age classes */
        const int groupsPerRange = d->ncjPacked / std::max(1, want);
        const int taper16        = nb->waveClassShareFixed[pw]
                                           ? 0
                                           : std::max(0, std::min(16, (groupsPerRange - c_shortRangeGroups) * 16 / (c_longRangeGroups - c_shortRangeGroups)));
        if (d->numWorkRanges[p] == want && !shortList
            && (d->workShareCount[p] != want || d->workPartFraction[p] != fraction || d->workShareTaper16[p] != taper16))
        {
            const int          classes = workPartitionWaves(p), perClass = slots / classes;
            std::vector<float> share(want);
            d->workShareTaper16[p] = taper16;
            /* the age-class shares were measured for the fused mode; with the atom-pair kernels running beside the cluster
             * kernel (split mode) equal shares are the better start (0.1037 vs 0.1055 ms per step) */
            for (int r = 0; r < want; r++)
            {
                const int   k       = std::min(classes - 1, (r % slots) / perClass);
                const float ofClass = nb->fusedFep ? (nb->waveClassShare[pw][k] * (16 - taper16) + nb->waveClassShareLong[pw][k] * taper16) / (16.0F * 1024.0F) : 1.0F;
                share[r]            = ofClass * (twoParts ? (r < slots ? fraction : 1.0F - fraction) : 1.0F);
            }
            setWorkShares(d, p, share.data(), want, s);
            d->workPartFraction[p] = fraction;
        }
        out[p].shareCum = (d->numWorkRanges[p] == want && !shortList) ? d->workShareCum[p] : nullptr;
        /* short lists, force flavours: one range fewer per SIMD, with age-class shares of their own (the energy flavours' three ranges: equal) */
        if (shortList && p == c_partitionForce && nb->fusedFep && nb->waveClassShareShort[0] > 0 && d->numWorkRanges[p] == nb->numSimds * (workPartitionWaves(p) - 1))
        {
            const int n = d->numWorkRanges[p], classes = workPartitionWaves(p) - 1, perClass = n / classes;
            if (d->workShareCount[p] != n)
            {
                std::vector<float> share(n);
                for (int r = 0; r < n; r++) { share[r] = nb->waveClassShareShort[std::min(classes - 1, r / perClass)] / 1024.0F; }
                setWorkShares(d, p, share.data(), n, s);
                d->workShareTaper16[p] = -1;
            }
            out[p].shareCum = d->workShareCum[p];
        }
    }
    /* workFirstSci shares work_nalloc with workRangeStart: reallocate when that one grew */
    for (int p = 0; p < c_numWorkPartitions; p++)
    {
        if (d->workFirstSciAlloc[p] < d->work_nalloc[p])
        {
            freeDeviceBuffer(&d->workFirstSci[p]);
            allocateDeviceBuffer(&d->workFirstSci[p], d->work_nalloc[p]);
            d->workFirstSciAlloc[p] = d->work_nalloc[p];
        }
        out[p].rangeStart = d->workRangeStart[p];
        out[p].firstSci   = d->workFirstSci[p];
    }
    hipLaunchKernelGGL(nbnxmWorkRangesKernel, dim3(numBlocks), dim3(c_workBlockSize), 0, s, d->groupWeight, d->weightBlockSum,
                       d->ncjPacked, numBlocks, d->sciSorted, d->nsciWork, out[0], out[1], out[2]);
    NBNXM_HIP_CHECK(hipGetLastError());
    /* the start record of every range (NbWorkDesc): one scalar load at the top of a wave instead of four dependent round trips */
    for (int p = 0; p < c_numWorkPartitions; p++)
    {
        reallocateDeviceBuffer(&d->workDesc[p], d->numWorkRanges[p], &dummy, &d->workDesc_nalloc[p]);
        hipLaunchKernelGGL(nbnxmWorkDescKernel, dim3((d->numWorkRanges[p] + c_workBlockSize - 1) / c_workBlockSize), dim3(c_workBlockSize), 0, s,
                           d->workRangeStart[p], d->workFirstSci[p], d->numWorkRanges[p], d->sciSorted, d->nsciWork, d->cjPacked, d->workDesc[p]);
    }
    NBNXM_HIP_CHECK(hipGetLastError());
    d->workRangesDirty = false; /* (setWorkShares above marks the ranges dirty: they have just been computed with the new shares) */
}

/* a deferred rolling-prune part in its own kernel, now */
static void flushPendingPrune(NbnxmGpu* nb, int iloc)
{
    gpu_plist* plist = nb->plist[iloc];
    if (plist->pendingPrunePart < 0) { return; }
    hipStream_t        s = nb->deviceStreams[iloc].stream;
    InteractionTimers& t = nb->timers[iloc];
    if (nb->bDoTime) { t.prune_k.openTimingRegion(s); }
    hipLaunchKernelGGL(selectPruneKernel(false), dim3(plist->pendingPruneEntries), dim3(c_waveSize), 0, s, *nb->atdat, *nb->nbparam, *plist,
                       plist->rollingPruningNumParts, plist->pendingPrunePart);
    NBNXM_HIP_CHECK(hipGetLastError());
    if (nb->bDoTime) { t.prune_k.closeTimingRegion(s); }
    plist->pendingPrunePart = -1;
}

void nbnxm_gpu_launch_kernel_pruneonly(NbnxmGpu* nb, int iloc, int numParts)
{
    gpu_plist*  plist = nb->plist[iloc];
    hipStream_t s     = nb->deviceStreams[iloc].stream;
    if (plist->haveFreshList)
    {
        NBNXM_ASSERT(numParts == 1, "with first pruning we expect 1 part");
        plist->rollingPruningNumParts = 0;
    }
    else
    {
        if (plist->rollingPruningNumParts == 0) { plist->rollingPruningNumParts = numParts; }
        else { NBNXM_ASSERT(numParts == plist->rollingPruningNumParts, "the number of rolling-prune parts may not change between searches"); }
    }
    const int part         = plist->haveFreshList ? 0 : plist->rollingPruningPart;
    const int numSciInPart = (plist->nsci - part + numParts - 1) / numParts;
    if (numSciInPart <= 0)
    {
        plist->haveFreshList = false;
        return;
    }
    InteractionTimers& t = nb->timers[iloc];
    flushPendingPrune(nb, iloc); /* an earlier part that no force-only launch has picked up */
    if (!plist->haveFreshList && nb->pruneMerged)
    {
        /* rolling pruning: the part runs in trailing workgroups of the next force-only cluster kernel (nbnxm_kernel_impl.h),
         * or in its own kernel ahead of any other flavour (nbnxm_gpu_launch_kernel) */
        plist->pendingPrunePart    = part;
        plist->pendingPruneEntries = numSciInPart;
    }
    else
    {
        if (nb->bDoTime) { t.prune_k.openTimingRegion(s); }
        const PruneKernelPtr kernel = selectPruneKernel(plist->haveFreshList);
        /* (first pass: several waves per i-entry, see nbnxmPruneKernel) */
        const int wavesPerEntry = plist->haveFreshList ? std::max(1, plist->pruneWavesPerEntry) : 1;
        hipLaunchKernelGGL(kernel, dim3(numSciInPart, wavesPerEntry), dim3(c_waveSize), 0, s, *nb->atdat, *nb->nbparam, *plist, numParts, part);
        NBNXM_HIP_CHECK(hipGetLastError());
        if (nb->bDoTime) { t.prune_k.closeTimingRegion(s); }
    }
    if (plist->haveFreshList)
    {
        plist->haveFreshList   = false;
        plist->firstPruneDone  = true;
        plist->workRangesDirty = true; /* the masks changed: re-balance */
        t.didPrune             = true;
    }
    else
    {
        plist->rollingPruningPart = (part + 1) % numParts;
        /* a rolling pass adds cluster pairs to a part of the masks, few and evenly spread: re-balance at the end of a full cycle,
         * and not more often than every c_rebalanceMinPruneCalls steps (the three partition kernels cost ~30 us together) */
        constexpr int c_rebalanceMinPruneCalls = 32;
        plist->pruneCallsSinceRebalance++;
        if (plist->rollingPruningPart == 0 && plist->pruneCallsSinceRebalance >= c_rebalanceMinPruneCalls)
        {
            plist->workRangesDirty          = true;
            plist->pruneCallsSinceRebalance = 0;
        }
        t.didRollingPrune = true;
    }
}

void nbnxm_gpu_launch_kernel(NbnxmGpu* nb, const nbnxm_step_workload_t* stepWork, int iloc)
{
    checkListErrorFlag(nb);
    NBNXM_ASSERT(iloc == NBNXM_LOCAL || (iloc == NBNXM_NONLOCAL && nb->bUseTwoStreams), "bad locality");
    NBAtomDataGpu*     adat  = nb->atdat;
    NBParamGpu*        nbp   = nb->nbparam;
    gpu_plist*         plist = nb->plist[iloc];
    hipStream_t        s     = nb->deviceStreams[iloc].stream;
    InteractionTimers& t     = nb->timers[iloc];
    NBNXM_ASSERT(adat->shiftVecUploaded, "shift vectors have not been uploaded");
    if (stepWork->computeEnergy || stepWork->computeDhdl) { nb->scalarsDirty = true; } /* see nbnxm_gpu_clear_outputs */
    if (stepWork->computeVirial) { nb->fshiftDirty = true; }
    /* nbnxm_gpu_launch_kernel_part: 1 = everything up to and including the first set of ranges (or the whole launch when the list is
     * not partitioned in two), 2 = the second set of ranges with the trailing workgroups (or nothing) */
    const int launchPart = nb->launchPartNow;
    NBNXM_ASSERT(launchPart != 2 || !plist->workRangesDirty, "the second part of a launch needs the partition its first part ran on");

    if (canSkipNonbondedWork(*nb, iloc))
    {
        plist->haveFreshList = false;
        return;
    }
    if (iloc == NBNXM_NONLOCAL)
    {
        /* the non-local kernel must see the local H2D and the output clearing (nbnxm_cuda.cu:625-641) */
        NBNXM_HIP_CHECK(hipStreamWaitEvent(s, nb->misc_ops_and_local_H2D_done, 0));
    }
    const bool secondPartOnly = (launchPart == 2); /* the perturbed-pair kernels, pruning and the partition belong to the first part */
    /* A.4: the reference returns before the FEP launch when the normal list is empty; here the
     * perturbed pairs are evaluated regardless.
     * The atom-pair kernels are few, latency-bound waves: they go first, on the locality's FEP stream, and
     * overlap with the cluster-pair kernel (the reference queues them behind it, nbnxm_cuda.cu:762-857). */
    const bool fused = nb->fusedFep && nbp->bFEP;
    /* foreign-lambda energies are wanted on dH/dl steps of soft-core runs (nbnxm_cuda.cu:817-856).  In fused mode
     * nbnxmFepClusterKernel accumulates them itself: no atom-pair list is needed at all. */
    const bool wantForeign = nbp->bFEP && nb->n_lambda > 0 && stepWork->computeDhdl
                             && ((nbp->softcoreType == NBNXM_SOFTCORE_GAPSYS) ? (nbp->gapsysLinpointCoul != 0.0F || nbp->gapsysLinpointVdw != 0.0F)
                                                                               : (nbp->alpha_coul != 0.0F || nbp->alpha_vdw != 0.0F));
    bool       fepForked = false;
    /* Atom-pair list (reference shape): its force / energy kernel rides in trailing workgroups of the cluster kernel whenever that
     * kernel runs and carries a tail for this flavour (NBNXM_HIP_FEP_LIST_MERGED=0: always a kernel of its own, on the FEP stream):
     * 0.0844 -> 0.0566 ms per force step of the 96k box in this mode — no second kernel competing for the wave slots, no second
     * stream, no fork / join events.  The foreign-lambda kernel of dH/dl steps stays a kernel of its own. */
    const bool energyStep    = stepWork->computeEnergy != 0;
    const bool tailCarriesFep = !energyStep || nb->energyTail >= 2;
    const bool mergeFepList  = nbp->bFEP && !fused && nb->fepListMerged && tailCarriesFep && plist->nsci > 0 && nb->feplist[iloc]->nrj > 0;
    if (nbp->bFEP && !fused && !secondPartOnly)
    {
        gpu_feplist* feplist   = nb->feplist[iloc];
        const bool   doForce   = !mergeFepList;
        const bool   doForeign = wantForeign;
        if (feplist->nri > 0 && feplist->nrj > 0 && (doForce || doForeign))
        {
            hipStream_t fs = s;
            if (nb->fepConcurrent && nb->fepStreams[iloc].stream != nullptr && plist->nsci > 0)
            {
                fs = nb->fepStreams[iloc].stream;
                NBNXM_HIP_CHECK(hipEventRecord(nb->fepFork[iloc], s));
                NBNXM_HIP_CHECK(hipStreamWaitEvent(fs, nb->fepFork[iloc], 0));
                fepForked = true;
            }
            const int nblock = (feplist->nrj + 255) / 256;
            if (nb->bDoTime) { t.fep_k.openTimingRegion(fs); }
            if (doForce)
            {
                const FepKernelPtr k = selectFepKernel(nbp->elecType, nbp->vdwType, stepWork->computeEnergy != 0);
                NBNXM_ASSERT(k != nullptr, "no FEP kernel for this electrostatics type");
                hipLaunchKernelGGL(k, dim3(nblock), dim3(256), 0, fs, *adat, *nbp, *feplist, stepWork->computeVirial);
                NBNXM_HIP_CHECK(hipGetLastError());
            }
            if (doForeign)
            {
                const FepKernelPtr k = selectFepForeignKernel(nbp->elecType, nbp->vdwType);
                NBNXM_ASSERT(k != nullptr, "no foreign-lambda kernel for this electrostatics type");
                const int foreignLds = (256 / c_waveSize) * 4 * (nb->n_lambda + 1) * static_cast<int>(sizeof(float));
                hipLaunchKernelGGL(k, dim3(nblock), dim3(256), foreignLds, fs, *adat, *nbp, *feplist, nb->n_lambda);
                NBNXM_HIP_CHECK(hipGetLastError());
            }
            if (nb->bDoTime) { t.fep_k.closeTimingRegion(fs); }
            if (fepForked) { NBNXM_HIP_CHECK(hipEventRecord(nb->fepJoin[iloc], fs)); }
        }
    }

    if (nbp->useDynamicPruning && plist->haveFreshList)
    {
        nbnxm_gpu_launch_kernel_pruneonly(nb, iloc, 1);
    }
    /* fused mode: the list of perturbed cluster pairs is built by the partition pass — behind the first prune, from the OUTER masks it
     * leaves in plist->imask (a superset of what any later rolling prune keeps), in the same pass that weighs the inner-pruned
     * working masks for the ranges: one partition pass per search step (it ran twice, once ahead of the prune for this list) */
    if (fused && plist->slowListDirty && plist->nsci > 0) { updateWorkPartition(nb, iloc); }
    pickUpSlowCount(nb, iloc);

    if (plist->nsci > 0)
    {
        const bool        energyFlavour = stepWork->computeEnergy != 0;
        /* Combination-rule flavours: the reference has them to save the table lookup.  Here the lookup is one LDS read and the
         * combination kernels pay for their 16 per-atom parameter registers with the fifth wave per SIMD, so while five table copies
         * fit the LDS (up to 28 types) the force-only step takes the table kernel: 0.0655 / 0.0695 -> 0.062 ms.  The table holds what
         * the rule gives (nbnxn_atomdata_t builds both from the same per-type parameters). */
        int vdwTypeKernel = nbp->vdwType;
        if ((nbp->vdwType == NBNXM_VDW_CUT_COMB_GEOM || nbp->vdwType == NBNXM_VDW_CUT_COMB_LB) && !energyFlavour && !nb->keepCombinationKernels
            && adat->numTypes <= c_maxTypesAtFullOccupancy)
        {
            vdwTypeKernel = NBNXM_VDW_CUT;
        }
        const NbKernelPtr kernel        = selectNbKernel(nbp->elecType, vdwTypeKernel, energyFlavour, fused);
        if (kernel == nullptr)
        {
            fatal(__FILE__, __LINE__, "nbnxm_gpu_launch_kernel", "no kernel for this electrostatics / VdW combination");
        }
        /* which trailing workgroups this flavour carries (the force flavours: all of them) */
        const bool tailPruneAndClear = !energyFlavour || nb->energyTail >= 1;
        const bool tailFep           = !energyFlavour || nb->energyTail >= 2;
        if (!tailPruneAndClear) { flushPendingPrune(nb, iloc); }
        if (plist->workRangesDirty) { updateWorkPartition(nb, iloc); }

        /* force-only steps: the perturbed cluster pairs ride in trailing workgroups of the cluster kernel (nbnxm_kernel_impl.h) */
        /* (foreign-lambda energies ride only with an energy flavour; a dH/dl step without energies keeps the kernel of its own) */
        const bool mergeFep = fused && plist->numSlowPairs > 0 && nb->fepMergedFused && tailFep && (!wantForeign || energyFlavour);
        const bool fepKernelOfItsOwn = fused && plist->numSlowPairs > 0 && !mergeFep && !secondPartOnly;
        /* NBNXM_HIP_FEP_CONCURRENT=3 (diagnostics): that kernel on the second stream BEHIND the cluster kernel's launch — its workgroups
         * are four waves with a table of a few hundred bytes and start in the first wave slots the ranges' waves free, where a
         * trailing workgroup of the cluster kernel needs a whole workgroup's slots and LDS */
        const bool fepBehindClusterKernel = fepKernelOfItsOwn && nb->fepBehindFused && nb->fepStreams[iloc].stream != nullptr;
        auto launchFepClusterKernel = [&]() {
            /* (energy / dH/dl steps, or NBNXM_HIP_FEP_MERGED=0; force-only steps: trailing workgroups of the cluster kernel, below)
             * the cluster pairs that touch a perturbed atom: a few thousand short latency-bound waves, ~10 us on the 96k
             * box.  On the same stream, ahead of the cluster kernel: measured on MI355X a second stream does not help here
             * (the cluster kernel fills every wave slot, so the other kernel's waves only start when those retire, and
             * the fork / join events cost more than the overlap gains: 0.101 vs 0.099 ms per step); the option stays
             * for experiments (NBNXM_HIP_FEP_CONCURRENT=2). */
            const FepClusterKernelPtr fk = selectFepClusterKernel(nbp->elecType, nbp->vdwType, stepWork->computeEnergy != 0 || wantForeign, wantForeign);
            NBNXM_ASSERT(fk != nullptr, "no perturbed-cluster-pair kernel for this electrostatics / VdW combination");
            hipStream_t fs = s;
            if (fepBehindClusterKernel)
            {
                fs = nb->fepStreams[iloc].stream;
                NBNXM_HIP_CHECK(hipStreamWaitEvent(fs, nb->fepFork[iloc], 0));
                fepForked = true;
            }
            else if (nb->fepConcurrentFused && nb->fepStreams[iloc].stream != nullptr)
            {
                fs = nb->fepStreams[iloc].stream;
                NBNXM_HIP_CHECK(hipEventRecord(nb->fepFork[iloc], s));
                NBNXM_HIP_CHECK(hipStreamWaitEvent(fs, nb->fepFork[iloc], 0));
                fepForked = true;
            }
            const bool fepUseTable = (nbp->vdwType == NBNXM_VDW_CUT || nbp->vdwType == NBNXM_VDW_FSWITCH || nbp->vdwType == NBNXM_VDW_PSWITCH
                                      || nbp->vdwType == NBNXM_VDW_EWALD_GEOM || nbp->vdwType == NBNXM_VDW_EWALD_LB);
            /* the LJ table of the workgroup; dH/dlambda steps: and a scratch area per wave (fepClusterPair's foreign-lambda terms) */
            const int  fepLds      = (fepUseTable ? ((adat->numTypes * adat->numTypes * 8 + 15) & ~15) : 0)
                               + (wantForeign ? c_fepClusterWavesPerBlockDef * c_fepForeignLdsBytes : 0);
            NBNXM_ASSERT(fepLds <= 160 * 1024, "too many atom types: the LJ parameter table does not fit the 160 KB LDS");
            if (fepLds > 64 * 1024)
            {
                NBNXM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, fepLds));
            }
            if (nb->bDoTime) { t.fep_k.openTimingRegion(fs); }
            /* one wave per perturbed cluster pair; dH/dlambda steps: c_fepForeignHeavyChunks per heavy one */
            const int numFepWaves = plist->numSlowPairs + (wantForeign ? plist->numSlowHeavy * (c_fepForeignHeavyChunks - 1) : 0);
            hipLaunchKernelGGL(fk, dim3((numFepWaves + c_fepClusterWavesPerBlockDef - 1) / c_fepClusterWavesPerBlockDef),
                               dim3(c_fepClusterWavesPerBlockDef * c_waveSize), fepLds, fs, *adat, *nbp, *plist, stepWork->computeVirial,
                               plist->sciSorted, plist->cjPacked, plist->excl, adat->xq, adat->atomTypes, adat->ljComb,
                               reinterpret_cast<const unsigned*>(adat->fepBits), wantForeign ? nb->n_lambda : -1);
            NBNXM_HIP_CHECK(hipGetLastError());
            if (nb->bDoTime) { t.fep_k.closeTimingRegion(fs); }
            if (fepForked) { NBNXM_HIP_CHECK(hipEventRecord(nb->fepJoin[iloc], fs)); }
        };
        if (fepBehindClusterKernel)
        {
            /* the fork is recorded AHEAD of the cluster kernel: the perturbed pairs depend on what came before it only */
            NBNXM_HIP_CHECK(hipEventRecord(nb->fepFork[iloc], s));
        }
        else if (fepKernelOfItsOwn) { launchFepClusterKernel(); }
        if (nb->bDoTime) { t.nb_k.openTimingRegion(s); }
        /* The LJ table lives in LDS (up to ~140 types in the 160 KB; large tables cost occupancy) */
        const bool ljEwald = (nbp->vdwType == NBNXM_VDW_EWALD_GEOM || nbp->vdwType == NBNXM_VDW_EWALD_LB);
        NBNXM_ASSERT(!ljEwald || nbp->nbfp_comb != nullptr, "LJ-PME kernel selected without the grid parameters (nbfp_comb)");
        const bool ewaldRTable = (nbp->elecType == NBNXM_ELEC_EWALD_TAB || nbp->elecType == NBNXM_ELEC_EWALD_TAB_TWIN);
        NBNXM_ASSERT(!ewaldRTable || nbp->coulombTabSize <= c_coulombTabMaxLds, "the Ewald force table is too large for the LDS (16384 entries)");
        const NbLaunchShape shape = chooseNbLaunchShape(nbp->elecType, vdwTypeKernel, energyFlavour, adat->numTypes, nbp->coulombTabSize,
                                                        nb->nbWavesPerBlock);
        NBNXM_ASSERT(shape.wavesPerSimd >= 1, "too many atom types: the LJ parameter table does not fit the 160 KB LDS");
        const int wavesPerBlock = shape.wavesPerBlock, wavesPerSimd = shape.wavesPerSimd, ldsBytes = shape.ldsBytes;
        int       wavesPerBlockLog2 = 0;
        while ((1 << wavesPerBlockLog2) < wavesPerBlock) { wavesPerBlockLog2++; }
        NBNXM_ASSERT((1 << wavesPerBlockLog2) == wavesPerBlock, "the cluster kernel's workgroups are 1, 2, 4, 8 or 16 waves");
        if (nb->debugLaunchShape)
        {
            std::fprintf(stderr, "nbnxm_hip: cluster kernel launch shape: %d types, %d waves per workgroup, %d waves per SIMD, %d LDS bytes per workgroup\n",
                         adat->numTypes, wavesPerBlock, wavesPerSimd, ldsBytes);
            nb->debugLaunchShape = false;
        }
        if (ldsBytes > 64 * 1024)
        {
            NBNXM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes));
        }
        /* one wave per resident wave slot, each with its share of the list (see updateWorkPartition) */
        /* (fewer than 4 resident waves per SIMD — 118 to 131 types — run the 4-wave partition in rounds) */
        /* (dH/dlambda steps of the energy flavours have a partition of their own: c_partitionDhdl) */
        const int p         = (wavesPerSimd >= 5) ? c_partitionForce : ((wantForeign && energyFlavour) ? c_partitionDhdl : c_partitionEnergy);
        const int numRanges = plist->numWorkRanges[p];
        NBNXM_ASSERT(numRanges > 0, "work partition missing");
        /* two sets of ranges (workParts 2): one launch per set; the trailing workgroups ride with the second.  A caller that does
         * not ask for a part gets both launches back to back. */
        const NbLaunchPlan plan     = planNbLaunch(launchPart, plist->workParts[p], numRanges);
        const int          firstSet = plan.firstSet, lastSet = plan.firstSet + plan.numSets - 1, setRanges = plan.setRanges;
        const bool         withTail = plan.withTail; /* false whenever nothing is launched: no state is consumed then */
        NBNXM_ASSERT(plan.numSets == 0 || (firstSet + plan.numSets) * setRanges <= numRanges, "launch plan beyond the range arrays");
        const int mergedFepItems = !withTail ? 0
                                   : mergeFep ? plist->numSlowPairs + ((wantForeign && energyFlavour) ? plist->numSlowHeavy * (c_fepForeignHeavyChunks - 1) : 0)
                                   : mergeFepList ? nb->feplist[iloc]->numClusterItems
                                                  : 0;
        const int pruneEntries   = (plist->pendingPrunePart >= 0 && withTail) ? plist->pendingPruneEntries : 0;
        const int prunePart      = std::max(plist->pendingPrunePart, 0);
        if (withTail) { plist->pendingPrunePart = -1; }
        /* the force flavour zeroes the spare force buffer for the next step (nbnxm_gpu_clear_outputs swaps) */
        int clearNumFloat4 = 0;
        /* Only the LOCAL launch does it: with two localities both kernels add to the buffer in use, and the buffer being zeroed
         * is the one the previous step's kernels wrote — the local stream has to be behind the previous NON-LOCAL kernel too,
         * whatever the caller's copy-back / reduction schedule was (nonlocalKernelDone). */
        if (nb->fDoubleBuffer && tailPruneAndClear && (3 * adat->numAtoms) % 4 == 0 && iloc == NBNXM_LOCAL && withTail)
        {
            if (nb->bUseTwoStreams && nb->nonlocalKernelRecorded)
            {
                NBNXM_HIP_CHECK(hipStreamWaitEvent(s, nb->nonlocalKernelDone, 0));
            }
            if (nb->fSpareAlloc < adat->numAtomsAlloc)
            {
                freeDeviceBuffer(&nb->fSpare);
                allocateDeviceBuffer(&nb->fSpare, adat->numAtomsAlloc);
                nb->fSpareAlloc = adat->numAtomsAlloc;
            }
            clearNumFloat4    = 3 * adat->numAtoms / 4;
            nb->fSpareCleared = true;
        }
        const int clearChunk     = wavesPerBlock * c_waveSize * static_cast<int>(c_clearFloat4PerThread);
        /* ... and the spare copy of the scalar outputs and shift forces, whenever it is not clean (after a swap: the previous step's) */
        float* clearB          = nullptr;
        int    clearNumFloat4B = 0;
        if (nb->fDoubleBuffer && nb->outputsDoubleBuffer && tailPruneAndClear && iloc == NBNXM_LOCAL && withTail && !nb->outputsSpareCleared
            && nb->numWindows == 0)
        {
            if (nb->bUseTwoStreams && nb->nonlocalKernelRecorded) { NBNXM_HIP_CHECK(hipStreamWaitEvent(s, nb->nonlocalKernelDone, 0)); }
            clearB                  = nb->outputsBlock[1 - nb->outputsActive];
            clearNumFloat4B         = (nb->numScalarOutputs + c_fshiftBlockFloats) / 4;
            nb->outputsSpareCleared = true;
        }
        for (int set = firstSet; set <= lastSet; set++)
        {
            const bool tail      = withTail && set == lastSet;
            const int  numBlocks = (setRanges + wavesPerBlock - 1) / wavesPerBlock
                                  + (tail ? (mergedFepItems + wavesPerBlock - 1) / wavesPerBlock + (pruneEntries + wavesPerBlock - 1) / wavesPerBlock
                                                    + (clearNumFloat4 + clearChunk - 1) / clearChunk + (clearNumFloat4B + clearChunk - 1) / clearChunk
                                          : 0);
            NBNXM_ASSERT(numBlocks > 0, "empty cluster-kernel launch");
#ifdef NBNXM_HOST_LAUNCH_TIMING /* diagnostics: host microseconds inside hipLaunchKernelGGL of the cluster kernel */
            static double s_us = 0; static long s_n = 0;
            const auto t0_ = std::chrono::steady_clock::now();
#endif
            hipLaunchKernelGGL(kernel, dim3(numBlocks), dim3(wavesPerBlock * c_waveSize), ldsBytes, s,
                               plist->workDesc[p] + set * setRanges, setRanges, wavesPerBlockLog2,
                               *adat, *nbp, *plist, stepWork->computeVirial, plist->sciSorted, plist->cjPacked, plist->excl, adat->xq,
                               adat->atomTypes, adat->ljComb, reinterpret_cast<const unsigned*>(adat->fepBits), plist->groupSlowMask,
                               tail ? mergedFepItems : 0, std::max(plist->rollingPruningNumParts, 1), prunePart, tail ? pruneEntries : 0,
                               reinterpret_cast<float4*>(nb->fSpare), tail ? clearNumFloat4 : 0, reinterpret_cast<float4*>(clearB),
                               tail ? clearNumFloat4B : 0, (wantForeign && energyFlavour) ? nb->n_lambda : -1,
                               *nb->feplist[iloc]);
#ifdef NBNXM_HOST_LAUNCH_TIMING
            s_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0_).count();
            if (++s_n % 200 == 0) { std::fprintf(stderr, "cluster kernel launch call: %.2f us on the host (mean of %ld)\n", s_us / s_n, s_n); }
#endif
            NBNXM_HIP_CHECK(hipGetLastError());
        }
        if (nb->bDoTime) { t.nb_k.closeTimingRegion(s); }
        if (fepBehindClusterKernel) { launchFepClusterKernel(); }
        if (iloc == NBNXM_NONLOCAL && nb->fDoubleBuffer)
        {
            NBNXM_HIP_CHECK(hipEventRecord(nb->nonlocalKernelDone, s));
            nb->nonlocalKernelRecorded = true;
        }
    }
    plist->haveFreshList = false;

    if (fepForked) { NBNXM_HIP_CHECK(hipStreamWaitEvent(s, nb->fepJoin[iloc], 0)); }
}

void nbnxm_gpu_set_local_launch_parts(NbnxmGpu* nb, int numParts, float firstPartFraction)
{
    NBNXM_ASSERT((numParts == 1 || numParts == 2) && firstPartFraction > 0.0F && firstPartFraction < 1.0F, "one or two parts, the first one a fraction of the work");
    if (nb->localLaunchParts == numParts && nb->localPartFraction == firstPartFraction) { return; }
    nb->localLaunchParts  = numParts;
    nb->localPartFraction = firstPartFraction;
    if (nb->plist[NBNXM_LOCAL] != nullptr) { nb->plist[NBNXM_LOCAL]->workRangesDirty = true; }
}

void nbnxm_gpu_launch_kernel_part(NbnxmGpu* nb, const nbnxm_step_workload_t* stepWork, int iloc, int part)
{
    NBNXM_ASSERT(part == 1 || part == 2, "part 1 or 2");
    nb->launchPartNow = part;
    nbnxm_gpu_launch_kernel(nb, stepWork, iloc);
    nb->launchPartNow = 0;
}

void nbnxm_gpu_launch_cpyback(NbnxmGpu* nb, float* f_out, const nbnxm_step_workload_t* stepWork,
                              int atomLocality)
{
    NBAtomDataGpu* ad   = nb->atdat;
    const int      iloc = atomLocality;
    NBNXM_ASSERT(iloc == NBNXM_LOCAL || (iloc == NBNXM_NONLOCAL && nb->bUseTwoStreams), "bad locality");
    hipStream_t s = nb->deviceStreams[iloc].stream;
    if (iloc == NBNXM_NONLOCAL && !nb->haveWork[iloc]) { return; }
    /* merged localities: the ONE kernel of the step, launched for the local locality, has computed the halo atoms' forces too and the
     * non-local device list is empty (the return above), so the local copy-back brings ALL atoms' forces — the halo rows the caller's
     * force halo sends included.  (Round 3 copied the home rows only with this call sequence: the halo forces never reached f_out.) */
    const bool all   = (iloc == NBNXM_LOCAL && nb->mergedLocalities);
    const int  begin = (iloc == NBNXM_LOCAL) ? 0 : ad->numAtomsLocal;
    const int  count = (iloc == NBNXM_LOCAL) ? (all ? ad->numAtoms : ad->numAtomsLocal) : ad->numAtoms - ad->numAtomsLocal;

    if (iloc == NBNXM_LOCAL && nb->bUseTwoStreams)
    {
        /* local forces also receive non-local kernel contributions (:1150-1160) */
        NBNXM_HIP_CHECK(hipStreamWaitEvent(s, nb->nonlocal_done, 0));
    }
    if (!stepWork->useGpuFBufferOps && count > 0)
    {
        NBNXM_ASSERT(f_out != nullptr, "force output buffer missing");
        NBNXM_HIP_CHECK(hipMemcpyAsync(f_out + 3 * static_cast<size_t>(begin), reinterpret_cast<float*>(ad->f) + 3 * static_cast<size_t>(begin),
                                       sizeof(float) * 3 * count, hipMemcpyDeviceToHost, s));
    }
    if (iloc == NBNXM_NONLOCAL)
    {
        NBNXM_HIP_CHECK(hipEventRecord(nb->nonlocal_done, s));
    }
    if (iloc == NBNXM_LOCAL)
    {
        if (stepWork->computeVirial)
        {
            NBNXM_HIP_CHECK(hipMemcpyAsync(nb->nbst.fShift, ad->fShift, sizeof(float) * c_fshiftBlockFloats, hipMemcpyDeviceToHost, s));
        }
        /* one copy of the scalar-output block instead of the reference's 4 + 4 small ones (:1263-1294):
         * energy steps need the accumulator slots too, dH/dl-only steps just the head */
        const bool wantForeign = nb->n_lambda > 0 && stepWork->computeDhdl;
        if (stepWork->computeEnergy || wantForeign)
        {
            /* (dH/dl steps bring the foreign-lambda slots, which lie behind the energy slots) */
            const int n = wantForeign ? nb->numScalarOutputs : (stepWork->computeEnergy ? nb->foreignSlotOffset : nb->numHeadScalars);
            NBNXM_HIP_CHECK(hipMemcpyAsync(nb->nbst.scalars, nb->scalarOutputs, sizeof(float) * n, hipMemcpyDeviceToHost, s));
            if (nb->numWindows > 0)
            {
                NBNXM_HIP_CHECK(hipMemcpyAsync(nb->h_windowSlots.data, ad->windowSlots,
                                               sizeof(float) * static_cast<size_t>(nb->numWindows) * ad->windowSlotStride, hipMemcpyDeviceToHost, s));
            }
        }
    }
}

static int finishTask(NbnxmGpu* nb, const nbnxm_step_workload_t* stepWork, int atomLocality, int haveSoftCore,
                      nbnxm_enerdata_t* enerd, float* shiftForces, bool wait)
{
    const int iloc = atomLocality;
    NBNXM_ASSERT(iloc == NBNXM_LOCAL || (iloc == NBNXM_NONLOCAL && nb->bUseTwoStreams), "bad locality");
    if (iloc == NBNXM_LOCAL || nb->haveWork[iloc])
    {
        if (!wait)
        {
            if (!nb->deviceStreams[iloc].completed()) { return 0; }
        }
        else { nb->deviceStreams[iloc].synchronize(); }
        checkListErrorFlag(nb); /* (the list check runs on the device: gpu_init_pairlist) */
        /* the count of perturbed cluster pairs of a new list is picked up without a wait (updateWorkPartition): the stream has just been
         * synchronised, so it has arrived, and a list with more of them than the fused mode provides for ends the run HERE — before the
         * caller consumes the outputs of a step whose kernels evaluated a truncated list (advisor finding of round 3) */
        if (wait || nb->deviceStreams[iloc].completed()) { pickUpSlowCount(nb, iloc); }
        accumulateTimings(nb, iloc);
        if (iloc == NBNXM_LOCAL)
        {
            /* gpu_reduce_staged_outputs, gpu_common.h:139-168 */
            if (stepWork->computeEnergy)
            {
                NBNXM_ASSERT(enerd != nullptr, "energy step without an energy accumulator");
                /* staged scalars (atom-pair kernels) + the cluster-pair kernel's accumulator slots */
                double sum[4] = { *nb->nbst.eLJ, *nb->nbst.eElec, *nb->nbst.dvdlLJ, *nb->nbst.dvdlElec };
                for (int k = 0; k < c_numEnergySlots; k++)
                {
                    for (int c = 0; c < 4; c++) { sum[c] += nb->nbst.energySlots[k * c_energySlotStride + c]; }
                }
                enerd->e_lj += sum[0];
                enerd->e_el += sum[1];
                double* dvdl = haveSoftCore ? enerd->dvdl_nonlin : enerd->dvdl_lin; /* gpu_common.h:420-428 */
                dvdl[0] += sum[3];
                dvdl[1] += sum[2];
            }
            if (nb->numWindows > 0 && (stepWork->computeEnergy || (nb->n_lambda > 0 && stepWork->computeDhdl)))
            {
                /* batched lambda windows: every window's own sums (nbnxm_gpu_get_window_energies); the caller's accumulators get
                 * the sum over the windows, like everything else of such an object */
                const NBAtomDataGpu* ad = nb->atdat;
                const int            n1 = nb->n_lambda + 1, per = 4 + 4 * n1;
                for (int w = 0; w < nb->numWindows; w++)
                {
                    double*      sums = nb->windowSums.data() + static_cast<size_t>(w) * per;
                    const float* base = nb->h_windowSlots.data + static_cast<size_t>(w) * ad->windowSlotStride;
                    std::fill(sums, sums + per, 0.0);
                    if (stepWork->computeEnergy)
                    {
                        for (int k = 0; k < c_numEnergySlots; k++)
                        {
                            for (int c = 0; c < 4; c++) { sums[c] += base[k * c_energySlotStride + c]; }
                        }
                    }
                    if (nb->n_lambda > 0 && stepWork->computeDhdl)
                    {
                        for (int k = 0; k < c_numForeignSlots; k++)
                        {
                            const float* slot = base + ad->windowForeignOffset + k * nb->foreignSlotStride;
                            for (int c = 0; c < 4; c++)
                            {
                                for (int idx = 0; idx < n1; idx++) { sums[4 + c * n1 + idx] += slot[c * n1 + idx]; }
                            }
                        }
                    }
                    if (enerd != nullptr) { nbnxm_gpu_get_window_energies(nb, w, enerd, haveSoftCore); }
                }
            }
            if (stepWork->computeVirial && shiftForces != nullptr)
            {
                for (int i = 0; i < 3 * c_numShiftVectors; i++)
                {
                    float sum = nb->nbst.fShift[i];
                    for (int k = 1; k <= c_numFshiftSlots; k++) { sum += nb->nbst.fShift[k * c_fshiftSlotStride + i]; }
                    shiftForces[i] += sum;
                }
            }
            /* gpu_reduce_staged_foreign_term, gpu_common.h:178-191 */
            if (nb->n_lambda > 0 && stepWork->computeDhdl && enerd != nullptr && enerd->foreign_energies != nullptr)
            {
                NBNXM_ASSERT(enerd->n_lambda == nb->n_lambda, "foreign-lambda accumulator has a different n_lambda");
                const int n1 = nb->n_lambda + 1;
                for (int idx = 0; idx <= nb->n_lambda; idx++)
                {
                    /* staged arrays (atom-pair foreign kernel) + the accumulator slots of the fused mode's kernel */
                    double t[4] = { nb->nbst.eLJForeign[idx], nb->nbst.eElecForeign[idx], nb->nbst.dvdlLJForeign[idx], nb->nbst.dvdlElecForeign[idx] };
                    for (int k = 0; k < c_numForeignSlots; k++)
                    {
                        const float* slot = nb->nbst.foreignSlots + k * nb->foreignSlotStride;
                        for (int c = 0; c < 4; c++) { t[c] += slot[c * n1 + idx]; }
                    }
                    enerd->foreign_energies[idx] += t[0] + t[1];
                    enerd->foreign_dhdl_vdw[idx] += t[2];
                    enerd->foreign_dhdl_coul[idx] += t[3];
                }
            }
        }
    }
    nb->timers[iloc].didPrune = nb->timers[iloc].didRollingPrune = false;
    nb->plist[iloc]->haveFreshList                               = false;
    return 1;
}

int nbnxm_gpu_try_finish_task(NbnxmGpu* nb, const nbnxm_step_workload_t* stepWork, int atomLocality,
                              int haveSoftCore, nbnxm_enerdata_t* enerd, float* shiftForces)
{
    return finishTask(nb, stepWork, atomLocality, haveSoftCore, enerd, shiftForces, false);
}

void nbnxm_gpu_wait_finish_task(NbnxmGpu* nb, const nbnxm_step_workload_t* stepWork,
                                int atomLocality, int haveSoftCore, nbnxm_enerdata_t* enerd,
                                float* shiftForces)
{
    finishTask(nb, stepWork, atomLocality, haveSoftCore, enerd, shiftForces, true);
}

void nbnxm_gpu_get_timings(NbnxmGpu* nb, nbnxm_gpu_timings_t* out)
{
    *out = nb->timings;
}

void nbnxm_gpu_reset_timings(NbnxmGpu* nb)
{
    for (auto& t : nb->timers)
    {
        t.nb_k.reset();
        t.fep_k.reset();
        t.prune_k.reset();
    }
    nb->timings = nbnxm_gpu_timings_t{};
}

void nbnxm_gpu_set_timing(NbnxmGpu* nb, int enable)
{
    nb->bDoTime = enable != 0;
}

int nbnxm_gpu_min_ci_balanced(NbnxmGpu* nb)
{
    /* The reference asks the list builder for 44 x #multiprocessors i-entries (cuda/nbnxm_cuda_data_mgmt.cu:82-109) because its kernel
     * balances by i-entry.  This kernel balances by wave-slot ranges that cut through i-entries (updateWorkPartition), so splitting only
     * adds i-entry starts: measured on MI355X, 96k atoms: unsplit list (2,637 entries) 61.6 us, 10,704 entries — what 44 x 256 would ask
     * for — 65.5 us, 19,551 entries 83.6 us.  0 = "no balancing" for the caller (pairlist.cpp:2572, 4102). */
    (void)nb;
    return 0;
}

int nbnxm_gpu_is_kernel_ewald_analytical(const NbnxmGpu* nb)
{
    return nb->nbparam->elecType == NBNXM_ELEC_EWALD_ANA || nb->nbparam->elecType == NBNXM_ELEC_EWALD_ANA_TWIN;
}

void* nbnxm_gpu_get_xq(NbnxmGpu* nb)
{
    return nb->atdat->xq;
}

void* nbnxm_gpu_get_f(NbnxmGpu* nb)
{
    nb->fDoubleBuffer = false; /* the caller may keep the pointer: no more swaps */
    return nb->atdat->f;
}

void* nbnxm_gpu_get_fshift(NbnxmGpu* nb)
{
    nb->outputsDoubleBuffer = false; /* the caller may keep the pointer: no more swaps of the output blocks */
    return nb->atdat->fShift;
}

void nbnxm_gpu_set_window_lambdas(NbnxmGpu* nb, int numWindows, int clustersPerWindow, const float* lambda_q, const float* lambda_v)
{
    NBParamGpu* nbp = nb->nbparam;
    freeDeviceBuffer(const_cast<float2**>(&nbp->windowLambda));
    freeDeviceBuffer(&nb->atdat->windowSlots);
    nbp->clustersPerWindow = 0;
    nb->numWindows         = 0;
    if (numWindows <= 0) { return; }
    NBNXM_ASSERT(clustersPerWindow > 0 && clustersPerWindow % c_numClPerSupercl == 0, "a window is a whole number of super-clusters");
    NBNXM_ASSERT(static_cast<long long>(numWindows) * clustersPerWindow * c_clSize >= nb->atdat->numAtoms,
                 "the windows do not cover the atoms (set the atom data first)");
    std::vector<float2> h(numWindows);
    for (int w = 0; w < numWindows; w++) { h[w] = make_float2(lambda_q[w], lambda_v[w]); }
    float2* d = nullptr;
    allocateDeviceBuffer(&d, numWindows);
    NBNXM_HIP_CHECK(hipStreamSynchronize(nb->deviceStreams[0].stream));
    NBNXM_HIP_CHECK(hipMemcpy(d, h.data(), sizeof(float2) * numWindows, hipMemcpyHostToDevice));
    nbp->windowLambda      = d;
    nbp->clustersPerWindow = clustersPerWindow;
    /* per-window accumulators: [energy slots | foreign-lambda slots] */
    NBAtomDataGpu* ad       = nb->atdat;
    ad->windowForeignOffset = c_numEnergySlots * c_energySlotStride;
    ad->windowSlotStride    = ad->windowForeignOffset + c_numForeignSlots * nb->foreignSlotStride;
    nb->numWindows          = numWindows;
    const size_t total      = static_cast<size_t>(numWindows) * ad->windowSlotStride;
    allocateDeviceBuffer(&ad->windowSlots, total);
    NBNXM_HIP_CHECK(hipMemset(ad->windowSlots, 0, sizeof(float) * total));
    nb->h_windowSlots.resize(total);
    nb->windowSums.assign(static_cast<size_t>(numWindows) * (4 + 4 * (nb->n_lambda + 1)), 0.0);
}

int nbnxm_gpu_get_window_energies(NbnxmGpu* nb, int window, nbnxm_enerdata_t* enerd, int haveSoftCore)
{
    if (window < 0 || window >= nb->numWindows || enerd == nullptr) { return -1; }
    const int     n1 = nb->n_lambda + 1;
    const double* w  = nb->windowSums.data() + static_cast<size_t>(window) * (4 + 4 * n1);
    enerd->e_lj += w[0];
    enerd->e_el += w[1];
    double* dvdl = haveSoftCore ? enerd->dvdl_nonlin : enerd->dvdl_lin;
    dvdl[0] += w[3];
    dvdl[1] += w[2];
    if (nb->n_lambda > 0 && enerd->foreign_energies != nullptr)
    {
        for (int idx = 0; idx < n1; idx++)
        {
            enerd->foreign_energies[idx] += w[4 + idx] + w[4 + n1 + idx];
            enerd->foreign_dhdl_vdw[idx] += w[4 + 2 * n1 + idx];
            enerd->foreign_dhdl_coul[idx] += w[4 + 3 * n1 + idx];
        }
    }
    return 0;
}

void* nbnxm_gpu_get_q4(NbnxmGpu* nb)
{
    return nb->atdat->q4;
}

void* nbnxm_gpu_get_stream(NbnxmGpu* nb, int iloc)
{
    return nb->deviceStreams[iloc].stream;
}

int nbnxm_gpu_have_short_range_work(const NbnxmGpu* nb, int iloc)
{
    return nb->haveWork[iloc] ? 1 : 0;
}

void* nbnxm_gpu_debug_get_cjpacked(NbnxmGpu* nb, int iloc)
{
    flushPendingPrune(nb, iloc); /* the caller is about to look at the masks */
    return nb->plist[iloc]->cjPacked;
}

#if defined(NBNXM_WAVE_TIMELINE) || defined(NBNXM_BLOCK_STATS)
/* diagnostics build only: copies the per-wave timeline of the last cluster-pair kernel launch */
void nbnxm_gpu_debug_timeline(NbnxmGpu* nb, unsigned long long* out, int numWaves)
{
    NBNXM_HIP_CHECK(hipStreamSynchronize(nb->deviceStreams[0].stream));
    NBNXM_ASSERT(nb->plist[0]->debugTimeline != nullptr, "no timeline buffer");
    NBNXM_HIP_CHECK(hipMemcpy(out, nb->plist[0]->debugTimeline, sizeof(unsigned long long) * 4 * numWaves, hipMemcpyDeviceToHost));
}
#endif

/* Experiment: clear + kernels of one force step captured into a hipGraph and replayed numSteps times */
void nbnxm_gpu_debug_graph_steps(NbnxmGpu* nb, const nbnxm_step_workload_t* stepWork, int numSteps)
{
    hipStream_t s = nb->deviceStreams[0].stream;
    /* a captured step replays with the buffer addresses of the capture: no swapping of force buffers here (and no allocation
     * inside the capture) */
    nb->fDoubleBuffer = false;
    /* steady state only: no fresh list, no dirty partition */
    nbnxm_gpu_clear_outputs(nb, stepWork->computeVirial);
    nbnxm_gpu_launch_kernel(nb, stepWork, NBNXM_LOCAL);
    NBNXM_HIP_CHECK(hipStreamSynchronize(s));
    const bool timing = nb->bDoTime;
    nb->bDoTime       = false;
    hipGraph_t     graph;
    hipGraphExec_t exec;
    NBNXM_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    nbnxm_gpu_clear_outputs(nb, stepWork->computeVirial);
    nbnxm_gpu_launch_kernel(nb, stepWork, NBNXM_LOCAL);
    NBNXM_HIP_CHECK(hipStreamEndCapture(s, &graph));
    NBNXM_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < numSteps; i++) { NBNXM_HIP_CHECK(hipGraphLaunch(exec, s)); }
    NBNXM_HIP_CHECK(hipStreamSynchronize(s));
    NBNXM_HIP_CHECK(hipGraphExecDestroy(exec));
    NBNXM_HIP_CHECK(hipGraphDestroy(graph));
    nb->bDoTime = timing;
}

void* nbnxm_gpu_debug_get_work_ranges(NbnxmGpu* nb, int iloc, int p, int* numRanges)
{
    NBNXM_ASSERT(p >= 0 && p < c_numWorkPartitions, "partition index is 0, 1 or 2");
    if (nb->plist[iloc]->workRangesDirty) { updateWorkPartition(nb, iloc); }
    *numRanges = nb->plist[iloc]->numWorkRanges[p];
    return nb->plist[iloc]->workRangeStart[p];
}

void nbnxm_gpu_debug_set_work_shares(NbnxmGpu* nb, int iloc, int p, const float* shares, int numRanges)
{
    NBNXM_ASSERT(p >= 0 && p < c_numWorkPartitions, "partition index is 0, 1 or 2");
    gpu_plist* d = nb->plist[iloc];
    if (d->workRangesDirty) { updateWorkPartition(nb, iloc); }
    NBNXM_ASSERT(numRanges == d->numWorkRanges[p] && numRanges == nb->numSimds * workPartitionWaves(p), "one share per wave slot of the device");
    setWorkShares(d, p, shares, numRanges, nb->deviceStreams[iloc].stream);
}

void nbnxm_gpu_debug_download(NbnxmGpu* nb, const void* devicePtr, void* hostPtr, size_t numBytes)
{
    NBNXM_HIP_CHECK(hipStreamSynchronize(nb->deviceStreams[0].stream));
    NBNXM_HIP_CHECK(hipMemcpy(hostPtr, devicePtr, numBytes, hipMemcpyDeviceToHost));
}

} // extern "C"
