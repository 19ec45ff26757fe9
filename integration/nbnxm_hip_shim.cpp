/*
 * nbnxm_hip_shim.cpp — the ONE translation unit a maintainer of the reference adds under src/gromacs/nbnxm/ to put
 * libnbnxm_hip.so (this repo, include/nbnxm_hip.h) behind the reference's own Nbnxm GPU interface.  It defines the functions the
 * reference declares in
 *     src/gromacs/nbnxm/gpu_data_mgmt.h:73-161      gpu_init ... gpu_free, cuda_copy_fepparams, gpu_init_feppairlist
 *     src/gromacs/nbnxm/nbnxm_gpu.h:93-311          gpu_copy_xq_to_gpu, gpu_launch_kernel, ..., gpu_wait_finish_task,
 *                                                   nbnxn_gpu_x_to_nbat_x, setupGpuShortRangeWork
 * with exactly those signatures (the compiler checks them against the reference's headers: tests/test_integration_shim.py runs
 * `g++ -fsyntax-only` on this file with -I into /root/reference in the build container) and forwards to the C ABI.
 * It replaces cuda/nbnxm_cuda.cu, cuda/nbnxm_cuda_data_mgmt.cu and the bodies of nbnxm_gpu_data_mgmt.cpp / gpu_common.h in a
 * build with GMX_GPU_HIP_NBNXM_FEP=ON; nothing in sim_util.cpp, kerneldispatch.cpp, pairlist.cpp or nbnxm_setup.cpp changes.
 *
 * NbnxmGpu stays an incomplete type on the reference's side (it only ever holds the pointer); the library's handle type of the
 * same name is what that pointer points to.
 *
 * Two ways to compile it:
 *  - in the reference's GPU build: the functions are defined in namespace Nbnxm, where gpu_data_mgmt.h / nbnxm_gpu.h declare them;
 *  - -DNBNXM_SHIM_CHECK_SIGNATURES (tests/test_integration_shim.py, `g++ -fsyntax-only`, CPU-only configuration of the reference's
 *    headers, where those declarations are inline stubs and a second definition would clash): the same functions are defined in
 *    namespace NbnxmShim and each is static_assert-ed to have the type of its Nbnxm:: declaration — a wrong parameter, const or
 *    return type, or a member name that does not exist on nbnxn_atomdata_t / NbnxnPairlistGpu / t_nblist / GridSet, fails there.
 *    The native stream of a DeviceStream exists only in a GPU flavour of device_stream.h (cudaStream_t today, hipStream_t in a HIP
 *    flavour), so that one accessor is stubbed in check mode.
 */
#include "gmxpre.h"

#include <vector>

#include <cstdlib>

#include "gromacs/gpu_utils/device_stream_manager.h"
#include "gromacs/gpu_utils/gpu_utils.h"
#include "gromacs/gpu_utils/gpueventsynchronizer.h"
#include "gromacs/listed_forces/listed_forces_gpu.h"
#include "gromacs/mdtypes/enerdata.h"
#include "gromacs/mdtypes/interaction_const.h"
#include "gromacs/mdtypes/locality.h"
#include "gromacs/mdtypes/nblist.h"
#include "gromacs/mdtypes/simulation_workload.h"
#include "gromacs/nbnxm/atomdata.h"
#include "gromacs/nbnxm/gpu_data_mgmt.h"
#include "gromacs/nbnxm/grid.h"
#include "gromacs/nbnxm/gridset.h"
#include "gromacs/nbnxm/nbnxm.h"
#include "gromacs/nbnxm/nbnxm_gpu.h"
#include "gromacs/nbnxm/pairlist.h"
#include "gromacs/nbnxm/pairlistparams.h"
#include "gromacs/nbnxm/pairlistsets.h"
#include "gromacs/pbcutil/ishift.h"
#include "gromacs/utility/arrayref.h"

/* The C ABI names the handle `struct NbnxmGpu`, as the reference does: one opaque type on both sides; the list element types
 * are the reference's own here. */
#define NBNXM_HIP_USE_REFERENCE_LIST_TYPES
#include "nbnxm_hip.h"

/* the list elements cross the boundary as they are: same size and member order as the ABI header lays out (nbnxm/pairlist.h:198-280) */
static_assert(sizeof(nbnxn_sci_t) == 16, "nbnxn_sci_t layout");
static_assert(sizeof(nbnxn_cj_packed_t) == 32, "nbnxn_cj_packed_t layout (cluster-pair split 2)");
static_assert(sizeof(nbnxn_excl_t) == 128, "nbnxn_excl_t layout (cluster-pair split 2)");

#ifdef NBNXM_SHIM_CHECK_SIGNATURES
#    include <type_traits>
#    define NBNXM_SHIM_NS NbnxmShim
using namespace Nbnxm; /* ElecType, VdwType, GridSet, Grid, ... as inside the namespace */
static void* nativeStream(const DeviceStream& /*stream*/)
{
    return nullptr;
}
#else
#    define NBNXM_SHIM_NS Nbnxm
static void* nativeStream(const DeviceStream& stream)
{
    return stream.stream();
}
#endif

namespace NBNXM_SHIM_NS
{

static nbnxm_step_workload_t toWorkload(const gmx::StepWorkload& sw)
{
    nbnxm_step_workload_t w;
    w.computeForces    = sw.computeForces;
    w.computeEnergy    = sw.computeEnergy;
    w.computeVirial    = sw.computeVirial;
    w.computeDhdl      = sw.computeDhdl;
    w.useGpuFBufferOps = sw.useGpuFBufferOps;
    return w;
}

/* Kernel flavours.  The selection rules are the reference's (nbnxm_gpu_data_mgmt.cpp:151-199,341-415, static there): analytical
 * Ewald unless GMX_GPU_NB_TAB_EWALD asks for the table (on MI355X the correction comes from an LDS table either way), twin-range
 * flavours when the cut-offs differ; LJ by modifier and combination rule. */
static ElecType pickElecType(const interaction_const_t& ic)
{
    if (ic.eeltype == CoulombInteractionType::Cut) { return ElecType::Cut; }
    if (usingRF(ic.eeltype)) { return ElecType::RF; }
    GMX_RELEASE_ASSERT(usingPme(ic.eeltype) || ic.eeltype == CoulombInteractionType::Ewald,
                       "electrostatics type without a GPU kernel");
    const bool tabulated = std::getenv("GMX_GPU_NB_TAB_EWALD") != nullptr && std::getenv("GMX_GPU_NB_ANA_EWALD") == nullptr;
    const bool twin      = ic.rcoulomb != ic.rvdw || std::getenv("GMX_GPU_NB_EWALD_TWINCUT") != nullptr;
    return tabulated ? (twin ? ElecType::EwaldTabTwin : ElecType::EwaldTab) : (twin ? ElecType::EwaldAnaTwin : ElecType::EwaldAna);
}

static VdwType pickVdwType(const interaction_const_t& ic, LJCombinationRule rule)
{
    if (ic.vdwtype == VanDerWaalsType::Pme)
    {
        return ic.ljpme_comb_rule == LongRangeVdW::Geom ? VdwType::EwaldGeom : VdwType::EwaldLB;
    }
    GMX_RELEASE_ASSERT(ic.vdwtype == VanDerWaalsType::Cut, "VdW type without a GPU kernel");
    switch (ic.vdw_modifier)
    {
        case InteractionModifiers::ForceSwitch: return VdwType::FSwitch;
        case InteractionModifiers::PotSwitch: return VdwType::PSwitch;
        default:
            return rule == LJCombinationRule::Geometric ? VdwType::CutCombGeom
                                                        : (rule == LJCombinationRule::LorentzBerthelot ? VdwType::CutCombLB : VdwType::Cut);
    }
}

/* ElecType / VdwType have the values of nbnxm_elec_type / nbnxm_vdw_type (nbnxm.h:181-214) */
static_assert(static_cast<int>(ElecType::EwaldAnaTwin) == NBNXM_ELEC_EWALD_ANA_TWIN && static_cast<int>(ElecType::RF) == NBNXM_ELEC_RF, "ElecType values");
static_assert(static_cast<int>(VdwType::EwaldLB) == NBNXM_VDW_EWALD_LB && static_cast<int>(VdwType::PSwitch) == NBNXM_VDW_PSWITCH, "VdwType values");

static nbnxm_interaction_params_t toParams(const interaction_const_t& ic, const PairlistParams& listParams, ElecType elec, VdwType vdw)
{
    nbnxm_interaction_params_t p = {};
    p.elecType          = static_cast<int>(elec);
    p.vdwType           = static_cast<int>(vdw);
    p.epsfac            = ic.epsfac;
    p.c_rf              = ic.reactionFieldShift;
    p.k_rf              = ic.reactionFieldCoefficient;
    p.ewaldcoeff_q      = ic.ewaldcoeff_q;
    p.sh_ewald          = ic.sh_ewald;
    p.ewaldcoeff_lj     = ic.ewaldcoeff_lj;
    p.sh_lj_ewald       = ic.sh_lj_ewald;
    p.rcoulomb          = ic.rcoulomb;
    p.rvdw              = ic.rvdw;
    p.rvdw_switch       = ic.rvdw_switch;
    p.rlistOuter        = listParams.rlistOuter;
    p.rlistInner        = listParams.rlistInner;
    p.useDynamicPruning = listParams.useDynamicPruning;
    p.dispersion_shift  = { ic.dispersion_shift.c2, ic.dispersion_shift.c3, ic.dispersion_shift.cpot };
    p.repulsion_shift   = { ic.repulsion_shift.c2, ic.repulsion_shift.c3, ic.repulsion_shift.cpot };
    p.vdw_switch        = { ic.vdw_switch.c3, ic.vdw_switch.c4, ic.vdw_switch.c5 };
    if (ic.coulombEwaldTables)
    {
        p.coulomb_tab       = const_cast<float*>(ic.coulombEwaldTables->tableF.data());
        p.coulomb_tab_size  = static_cast<int>(ic.coulombEwaldTables->tableF.size());
        p.coulomb_tab_scale = ic.coulombEwaldTables->scale;
    }
    return p;
}

/* the rank's stream manager (one NbnxmGpu per rank): nbnxn_gpu_x_to_nbat_x has to enqueue a wait into the locality's DeviceStream */
static const gmx::DeviceStreamManager* s_deviceStreamManager = nullptr;

NbnxmGpu* gpu_init(const gmx::DeviceStreamManager& deviceStreamManager,
                   const interaction_const_t*      ic,
                   const PairlistParams&           listParams,
                   const nbnxn_atomdata_t*         nbat,
                   bool                            bLocalAndNonlocal,
                   bool                            bFEP,
                   int                             n_lambda)
{
    s_deviceStreamManager                  = &deviceStreamManager;
    const nbnxn_atomdata_t::Params& params = nbat->params();
    const nbnxm_interaction_params_t p = toParams(*ic, listParams, pickElecType(*ic), pickVdwType(*ic, params.ljCombinationRule));
    return nbnxm_gpu_init(&p, params.numTypes, params.nbfp.data(), params.nbfp_comb.data(), bLocalAndNonlocal, bFEP, n_lambda,
                          nativeStream(deviceStreamManager.stream(gmx::DeviceStreamType::NonBondedLocal)),
                          bLocalAndNonlocal ? nativeStream(deviceStreamManager.stream(gmx::DeviceStreamType::NonBondedNonLocal)) : nullptr);
}

void cuda_copy_fepparams(NbnxmGpu*   nb,
                         const bool  bFEP,
                         const float alpha_coul,
                         const float alpha_vdw,
                         const int   lam_power,
                         const float sc_sigma6_def,
                         const float sc_sigma6_min,
                         const float lambda_q,
                         const float lambda_v,
                         const int   n_lambda,
                         gmx::EnumerationArray<FreeEnergyPerturbationCouplingType, std::vector<double>> all_lambda)
{
    nbnxm_gpu_copy_fepparams(nb, bFEP, alpha_coul, alpha_vdw, lam_power, sc_sigma6_def, sc_sigma6_min, lambda_q, lambda_v, n_lambda,
                             all_lambda[FreeEnergyPerturbationCouplingType::Coul].data(),
                             all_lambda[FreeEnergyPerturbationCouplingType::Vdw].data());
}

void gpu_init_pairlist(NbnxmGpu* nb, const struct NbnxnPairlistGpu* h_nblist, gmx::InteractionLocality iloc)
{
    nbnxm_gpu_init_pairlist(nb, static_cast<int>(iloc), h_nblist->na_ci, static_cast<int>(h_nblist->sci.size()),
                            h_nblist->sci.data(), static_cast<int>(h_nblist->cjPacked.size()), h_nblist->cjPacked.list_.data(),
                            static_cast<int>(h_nblist->excl.size()), h_nblist->excl.data());
}

void gpu_init_feppairlist(NbnxmGpu* nb, struct t_nblist* h_nblist, gmx::InteractionLocality iloc, const GridSet& gridSet)
{
    nbnxm_gpu_init_feppairlist(nb, static_cast<int>(iloc), h_nblist->nri, h_nblist->iinr.data(), h_nblist->shift.data(),
                               h_nblist->jindex.data(), h_nblist->nrj, h_nblist->jjnr.data(),
                               h_nblist->excl_fep.empty() ? nullptr : h_nblist->excl_fep.data(),
                               static_cast<int>(gridSet.atomIndices().size()), gridSet.atomIndices().data());
}

void gpu_init_atomdata(NbnxmGpu* nb, const nbnxn_atomdata_t* nbat)
{
    const nbnxn_atomdata_t::Params& q = nbat->params();
    nbnxm_gpu_init_atomdata(nb, nbat->numAtoms(), nbat->natoms_local, q.type.data(), q.lj_comb.data(), q.qA.data(), q.qB.data(),
                            q.typeA.data(), q.typeB.data(), q.lj_combA.data(), q.lj_combB.data());
}

void gpu_pme_loadbal_update_param(const struct nonbonded_verlet_t* nbv, const interaction_const_t& ic)
{
    if (nbv == nullptr || !nbv->useGpu()) { return; }
    NbnxmGpu* nb = nbv->gpu_nbv;
    /* the cut-offs, the Ewald coefficient, its flavour and its table change (nbnxm_gpu_data_mgmt.cpp:627-645); the library keeps
     * its VdW flavour, so that member is not read */
    const nbnxm_interaction_params_t p = toParams(ic, nbv->pairlistSets().params(), pickElecType(ic), VdwType::Cut);
    nbnxm_gpu_pme_loadbal_update_param(nb, &p);
}

void gpu_upload_shiftvec(NbnxmGpu* nb, const nbnxn_atomdata_t* nbatom)
{
    nbnxm_gpu_upload_shiftvec(nb, reinterpret_cast<const float*>(nbatom->shift_vec.data()));
}

void gpu_clear_outputs(NbnxmGpu* nb, bool computeVirial)
{
    nbnxm_gpu_clear_outputs(nb, computeVirial);
}

void gpu_free(NbnxmGpu* nb)
{
    nbnxm_gpu_free(nb);
}

int gpu_min_ci_balanced(NbnxmGpu* nb)
{
    return nbnxm_gpu_min_ci_balanced(nb);
}

bool gpu_is_kernel_ewald_analytical(const NbnxmGpu* nb)
{
    return nbnxm_gpu_is_kernel_ewald_analytical(nb) != 0;
}

DeviceBuffer<gmx::RVec> gpu_get_f(NbnxmGpu* nb)
{
    return static_cast<DeviceBuffer<gmx::RVec>>(nbnxm_gpu_get_f(nb));
}

void gpu_copy_xq_to_gpu(NbnxmGpu* nb, const struct nbnxn_atomdata_t* nbdata, gmx::AtomLocality aloc)
{
    /* nbat->XFormat must be nbatXYZQ, as for every GPU flavour of the reference */
    nbnxm_gpu_copy_xq_to_gpu(nb, nbdata->x().data(), static_cast<int>(aloc));
}

void gpu_launch_kernel(NbnxmGpu* nb, const gmx::StepWorkload& stepWork, gmx::InteractionLocality iloc)
{
    const nbnxm_step_workload_t w = toWorkload(stepWork);
    nbnxm_gpu_launch_kernel(nb, &w, static_cast<int>(iloc));
}

void gpu_launch_kernel_pruneonly(NbnxmGpu* nb, gmx::InteractionLocality iloc, int numParts)
{
    nbnxm_gpu_launch_kernel_pruneonly(nb, static_cast<int>(iloc), numParts);
}

void gpu_launch_cpyback(NbnxmGpu* nb, nbnxn_atomdata_t* nbatom, const gmx::StepWorkload& stepWork, gmx::AtomLocality aloc)
{
    const nbnxm_step_workload_t w = toWorkload(stepWork);
    nbnxm_gpu_launch_cpyback(nb, nbatom->out[0].f.data(), &w, static_cast<int>(aloc));
}

/* gpu_common.h:293-385.  The library does the staged reduction of gpu_common.h:139-191 itself (sums of its accumulator copies)
 * and hands back plain numbers, which are added where the reference adds them. */
bool gpu_try_finish_task(NbnxmGpu*                nb,
                         const gmx::StepWorkload& stepWork,
                         gmx::AtomLocality        aloc,
                         real*                    e_lj,
                         real*                    e_el,
                         double*                  dvdl_lj,
                         double*                  dvdl_el,
                         gmx::ArrayRef<gmx::RVec> shiftForces,
                         ForeignLambdaTerms*      foreign_term,
                         GpuTaskCompletion        completionKind,
                         gmx_wallcycle*           /*wcycle*/)
{
    const nbnxm_step_workload_t w = toWorkload(stepWork);
    const int                   n = (foreign_term != nullptr && stepWork.computeDhdl) ? foreign_term->numLambdas() : 0;
    std::vector<double>         fe(n + 1, 0.0), fc(n + 1, 0.0), fv(n + 1, 0.0);
    std::vector<float>          fshift(3 * gmx::c_numShiftVectors, 0.0F);
    nbnxm_enerdata_t            e = {};
    e.n_lambda                    = n;
    e.foreign_energies            = fe.data();
    e.foreign_dhdl_coul           = fc.data();
    e.foreign_dhdl_vdw            = fv.data();
    /* haveSoftCore = 1: everything arrives in dvdl_nonlin[]; the CALLER has picked the lin or nonlin destination (below) */
    if (completionKind == GpuTaskCompletion::Check)
    {
        if (!nbnxm_gpu_try_finish_task(nb, &w, static_cast<int>(aloc), 1, &e, fshift.data())) { return false; }
    }
    else
    {
        nbnxm_gpu_wait_finish_task(nb, &w, static_cast<int>(aloc), 1, &e, fshift.data());
    }
    if (stepWork.computeEnergy)
    {
        *e_lj += static_cast<real>(e.e_lj);
        *e_el += static_cast<real>(e.e_el);
    }
    *dvdl_el += e.dvdl_nonlin[0];
    *dvdl_lj += e.dvdl_nonlin[1];
    for (int i = 0; i <= n && n > 0; i++) /* gpu_reduce_staged_foreign_term, gpu_common.h:171-191; accumulate(): mdtypes/enerdata.h:123-130 */
    {
        foreign_term->accumulate(i, FreeEnergyPerturbationCouplingType::Vdw, fe[i], fv[i]);
        foreign_term->accumulate(i, FreeEnergyPerturbationCouplingType::Coul, 0.0, fc[i]);
    }
    if (stepWork.computeVirial)
    {
        for (int s = 0; s < gmx::c_numShiftVectors; s++)
        {
            for (int d = 0; d < DIM; d++) { shiftForces[s][d] += fshift[3 * s + d]; }
        }
    }
    return true;
}

/* gpu_common.h:405-435 */
float gpu_wait_finish_task(NbnxmGpu*                nb,
                           const gmx::StepWorkload& stepWork,
                           gmx::AtomLocality        aloc,
                           const bool               haveSoftCore,
                           gmx_enerdata_t*          enerd,
                           gmx::ArrayRef<gmx::RVec> shiftForces,
                           gmx_wallcycle*           wcycle)
{
    real*   e_lj    = enerd->grpp.energyGroupPairTerms[NonBondedEnergyTerms::LJSR].data();
    real*   e_el    = enerd->grpp.energyGroupPairTerms[NonBondedEnergyTerms::CoulombSR].data();
    auto&   dvdl    = haveSoftCore ? enerd->dvdl_nonlin : enerd->dvdl_lin;
    double* dvdl_lj = &dvdl[FreeEnergyPerturbationCouplingType::Vdw];
    double* dvdl_el = &dvdl[FreeEnergyPerturbationCouplingType::Coul];
    gpu_try_finish_task(nb, stepWork, aloc, e_lj, e_el, dvdl_lj, dvdl_el, shiftForces, &enerd->foreignLambdaTerms,
                        GpuTaskCompletion::Wait, wcycle);
    return 0.0F;
}

void nbnxn_gpu_init_x_to_nbat_x(const Nbnxm::GridSet& gridSet, NbnxmGpu* gpu_nbv)
{
    nbnxm_gpu_init_x_to_nbat_x(gpu_nbv, static_cast<int>(gridSet.atomIndices().size()), gridSet.atomIndices().data());
}

void nbnxn_gpu_x_to_nbat_x(const Nbnxm::Grid&      grid,
                           NbnxmGpu*               gpu_nbv,
                           DeviceBuffer<gmx::RVec> d_x,
                           GpuEventSynchronizer*   xReadyOnDevice,
                           gmx::AtomLocality       locality,
                           int                     /*gridId*/,
                           int                     /*numColumnsMax*/,
                           bool                    mustInsertNonLocalDependency)
{
    /* the coordinates must be on the device before the kernel reads them (nbnxm_gpu_buffer_ops.cpp: enqueueWaitEvent on the
     * locality's stream); GpuEventSynchronizer does not hand out its native event, so the wait is queued here, on the same native
     * stream the library launches on */
    if (xReadyOnDevice != nullptr)
    {
        xReadyOnDevice->enqueueWaitEvent(s_deviceStreamManager->stream(
                locality == gmx::AtomLocality::Local ? gmx::DeviceStreamType::NonBondedLocal : gmx::DeviceStreamType::NonBondedNonLocal));
    }
    const int begin = grid.cellOffset() * grid.numAtomsPerCell();
    nbnxm_gpu_x_to_nbat_x(gpu_nbv, d_x, nullptr, static_cast<int>(locality), begin, begin + grid.numCells() * grid.numAtomsPerCell(),
                          mustInsertNonLocalDependency);
}

void nbnxnInsertNonlocalGpuDependency(NbnxmGpu* nb, gmx::InteractionLocality interactionLocality)
{
    nbnxm_gpu_insert_nonlocal_dependency(nb, static_cast<int>(interactionLocality));
}

void setupGpuShortRangeWork(NbnxmGpu* nb, const gmx::ListedForcesGpu* listedForcesGpu, gmx::InteractionLocality iLocality)
{
    nbnxm_gpu_setup_short_range_work(nb, listedForcesGpu != nullptr && listedForcesGpu->haveInteractions(), static_cast<int>(iLocality));
}

bool haveGpuShortRangeWork(const NbnxmGpu* nb, gmx::InteractionLocality interactionLocality)
{
    return nbnxm_gpu_have_short_range_work(nb, static_cast<int>(interactionLocality)) != 0;
}

} // namespace NBNXM_SHIM_NS

#ifdef NBNXM_SHIM_CHECK_SIGNATURES
/* decltype(&f) is ill-formed for an overloaded name: every line also proves that the reference declares ONE function of that name */
#    define SAME_SIGNATURE(f) static_assert(std::is_same_v<decltype(&Nbnxm::f), decltype(&NbnxmShim::f)>, #f " differs from the reference's declaration")
SAME_SIGNATURE(gpu_init);
SAME_SIGNATURE(cuda_copy_fepparams);
SAME_SIGNATURE(gpu_init_pairlist);
SAME_SIGNATURE(gpu_init_feppairlist);
SAME_SIGNATURE(gpu_init_atomdata);
SAME_SIGNATURE(gpu_pme_loadbal_update_param);
SAME_SIGNATURE(gpu_upload_shiftvec);
SAME_SIGNATURE(gpu_clear_outputs);
SAME_SIGNATURE(gpu_free);
SAME_SIGNATURE(gpu_min_ci_balanced);
SAME_SIGNATURE(gpu_is_kernel_ewald_analytical);
SAME_SIGNATURE(gpu_get_f);
SAME_SIGNATURE(gpu_copy_xq_to_gpu);
SAME_SIGNATURE(gpu_launch_kernel);
SAME_SIGNATURE(gpu_launch_kernel_pruneonly);
SAME_SIGNATURE(gpu_launch_cpyback);
SAME_SIGNATURE(gpu_try_finish_task);
SAME_SIGNATURE(gpu_wait_finish_task);
SAME_SIGNATURE(nbnxn_gpu_init_x_to_nbat_x);
SAME_SIGNATURE(nbnxn_gpu_x_to_nbat_x);
SAME_SIGNATURE(nbnxnInsertNonlocalGpuDependency);
SAME_SIGNATURE(setupGpuShortRangeWork);
SAME_SIGNATURE(haveGpuShortRangeWork);
#endif
