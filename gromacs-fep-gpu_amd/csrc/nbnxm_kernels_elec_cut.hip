/* Cluster-pair kernel instantiations, electrostatics flavour: elec_cut (see nbnxm_kernel_impl.h). */
#include "nbnxm_fep_cluster_kernel_impl.h"
#include "nbnxm_kernel_impl.h"
#include "nbnxm_kernels.h"

template<int VDW>
static NbKernelPtr pick(bool energy, bool fused)
{
        if (energy) { return fused ? nbnxmKernel<ELK_CUT, false, VDW, true, true> : nbnxmKernel<ELK_CUT, false, VDW, true, false>; }
    return fused ? nbnxmKernel<ELK_CUT, false, VDW, false, true> : nbnxmKernel<ELK_CUT, false, VDW, false, false>;
}

NbKernelPtr nbKernelElecCut(int vdwKind, bool energy, bool fused)
{
    switch (vdwKind)
    {
        case VDK_CUT: return pick<VDK_CUT>(energy, fused);
        case VDK_COMB_GEOM: return pick<VDK_COMB_GEOM>(energy, fused);
        case VDK_COMB_LB: return pick<VDK_COMB_LB>(energy, fused);
        case VDK_FSWITCH: return pick<VDK_FSWITCH>(energy, fused);
        case VDK_PSWITCH: return pick<VDK_PSWITCH>(energy, fused);
        case VDK_EWALD_GEOM: return pick<VDK_EWALD_GEOM>(energy, fused);
        case VDK_EWALD_LB: return pick<VDK_EWALD_LB>(energy, fused);
        default: return nullptr;
    }
}

template<int VDW>
static FepClusterKernelPtr pickFepCluster(bool energy, bool foreign)
{
    if (foreign) { return nbnxmFepClusterKernel<ELK_CUT, false, VDW, true, true>; }
    return energy ? nbnxmFepClusterKernel<ELK_CUT, false, VDW, true, false> : nbnxmFepClusterKernel<ELK_CUT, false, VDW, false, false>;
}

FepClusterKernelPtr nbKernelElecCutFepCluster(int vdwKind, bool energy, bool foreign)
{
    switch (vdwKind)
    {
        case VDK_CUT: return pickFepCluster<VDK_CUT>(energy, foreign);
        case VDK_COMB_GEOM: return pickFepCluster<VDK_COMB_GEOM>(energy, foreign);
        case VDK_COMB_LB: return pickFepCluster<VDK_COMB_LB>(energy, foreign);
        case VDK_FSWITCH: return pickFepCluster<VDK_FSWITCH>(energy, foreign);
        case VDK_PSWITCH: return pickFepCluster<VDK_PSWITCH>(energy, foreign);
        case VDK_EWALD_GEOM: return pickFepCluster<VDK_EWALD_GEOM>(energy, foreign);
        case VDK_EWALD_LB: return pickFepCluster<VDK_EWALD_LB>(energy, foreign);
        default: return nullptr;
    }
}
