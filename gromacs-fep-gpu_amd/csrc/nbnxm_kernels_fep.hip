/* Atom-pair FEP, foreign-lambda and prune kernel instantiations + the selection functions. */
#include "nbnxm_fep_kernel_impl.h"
#include "nbnxm_kernel_impl.h"
#include "nbnxm_kernels.h"

static int vdwKindOf(int vdwType)
{
    switch (vdwType)
    {
        case NBNXM_VDW_CUT: return VDK_CUT;
        case NBNXM_VDW_CUT_COMB_GEOM: return VDK_COMB_GEOM;
        case NBNXM_VDW_CUT_COMB_LB: return VDK_COMB_LB;
        case NBNXM_VDW_FSWITCH: return VDK_FSWITCH;
        case NBNXM_VDW_PSWITCH: return VDK_PSWITCH;
        case NBNXM_VDW_EWALD_GEOM: return VDK_EWALD_GEOM;
        case NBNXM_VDW_EWALD_LB: return VDK_EWALD_LB;
        default: return -1;
    }
}

NbKernelPtr selectNbKernel(int elecType, int vdwType, bool energy, bool fused)
{
    const int vdwKind = vdwKindOf(vdwType);
    if (vdwKind < 0) { return nullptr; }
    switch (elecType)
    {
        case NBNXM_ELEC_CUT: return nbKernelElecCut(vdwKind, energy, fused);
        case NBNXM_ELEC_RF: return nbKernelElecRF(vdwKind, energy, fused);
        case NBNXM_ELEC_EWALD_ANA: return nbKernelElecEwaldAna(vdwKind, energy, fused);
        case NBNXM_ELEC_EWALD_TAB: return nbKernelElecEwaldTab(vdwKind, energy, fused);
        case NBNXM_ELEC_EWALD_ANA_TWIN: return nbKernelElecEwaldAnaTwin(vdwKind, energy, fused);
        case NBNXM_ELEC_EWALD_TAB_TWIN: return nbKernelElecEwaldTabTwin(vdwKind, energy, fused);
        default: return nullptr;
    }
}

FepClusterKernelPtr selectFepClusterKernel(int elecType, int vdwType, bool energy, bool foreign)
{
    const int vdwKind = vdwKindOf(vdwType);
    if (vdwKind < 0) { return nullptr; }
    switch (elecType)
    {
        case NBNXM_ELEC_CUT: return nbKernelElecCutFepCluster(vdwKind, energy, foreign);
        case NBNXM_ELEC_RF: return nbKernelElecRFFepCluster(vdwKind, energy, foreign);
        case NBNXM_ELEC_EWALD_ANA: return nbKernelElecEwaldAnaFepCluster(vdwKind, energy, foreign);
        case NBNXM_ELEC_EWALD_TAB: return nbKernelElecEwaldTabFepCluster(vdwKind, energy, foreign);
        case NBNXM_ELEC_EWALD_ANA_TWIN: return nbKernelElecEwaldAnaTwinFepCluster(vdwKind, energy, foreign);
        case NBNXM_ELEC_EWALD_TAB_TWIN: return nbKernelElecEwaldTabTwinFepCluster(vdwKind, energy, foreign);
        default: return nullptr;
    }
}

/* The perturbed-pair math has three electrostatics forms (cut-off == RF with k_rf 0, nb_free_energy.cpp:377-386)
 * and LJ with or without potential switch; force switch evaluates plain shifted LJ for perturbed pairs, as in the
 * reference; LJ-PME is a run-time branch of fepPair (NBParamGpu::vdwType). */
template<int ELEC>
static FepKernelPtr pickFep(bool pswitch, bool energy)
{
    if (pswitch) { return energy ? nbnxmFepKernel<ELEC, true, true> : nbnxmFepKernel<ELEC, true, false>; }
    return energy ? nbnxmFepKernel<ELEC, false, true> : nbnxmFepKernel<ELEC, false, false>;
}

FepKernelPtr selectFepKernel(int elecType, int vdwType, bool energy)
{
    const bool pswitch = (vdwType == NBNXM_VDW_PSWITCH);
    switch (elecType)
    {
        case NBNXM_ELEC_CUT:
        case NBNXM_ELEC_RF: return pickFep<ELK_RF>(pswitch, energy);
        case NBNXM_ELEC_EWALD_ANA:
        case NBNXM_ELEC_EWALD_ANA_TWIN: return pickFep<ELK_EWALD_ANA>(pswitch, energy);
        case NBNXM_ELEC_EWALD_TAB:
        case NBNXM_ELEC_EWALD_TAB_TWIN: return pickFep<ELK_EWALD_TAB>(pswitch, energy);
        default: return nullptr;
    }
}

FepKernelPtr selectFepForeignKernel(int elecType, int vdwType)
{
    const bool pswitch = (vdwType == NBNXM_VDW_PSWITCH);
    switch (elecType)
    {
        case NBNXM_ELEC_CUT:
        case NBNXM_ELEC_RF: return pswitch ? nbnxmFepForeignKernel<ELK_RF, true> : nbnxmFepForeignKernel<ELK_RF, false>;
        case NBNXM_ELEC_EWALD_ANA:
        case NBNXM_ELEC_EWALD_ANA_TWIN:
        case NBNXM_ELEC_EWALD_TAB:
        case NBNXM_ELEC_EWALD_TAB_TWIN:
            /* energies only: analytical and tabulated flavours evaluate the same erf */
            return pswitch ? nbnxmFepForeignKernel<ELK_EWALD_ANA, true> : nbnxmFepForeignKernel<ELK_EWALD_ANA, false>;
        default: return nullptr;
    }
}

int nbKernelWavesPerEu(int vdwType, bool energy, bool fused)
{
    (void)fused;
    if (energy) { return c_nbWavesPerEu<VDK_CUT, true>; }
    switch (vdwKindOf(vdwType))
    {
        case VDK_CUT: return c_nbWavesPerEu<VDK_CUT, false>;
        case VDK_FSWITCH: return c_nbWavesPerEu<VDK_FSWITCH, false>;
        case VDK_PSWITCH: return c_nbWavesPerEu<VDK_PSWITCH, false>;
        case VDK_COMB_GEOM:
        case VDK_COMB_LB: return c_nbWavesPerEu<VDK_COMB_GEOM, false>;
        default: return c_nbWavesPerEu<VDK_EWALD_LB, false>;
    }
}

PruneKernelPtr selectPruneKernel(bool haveFreshList)
{
    return haveFreshList ? nbnxmPruneKernel<true> : nbnxmPruneKernel<false>;
}
