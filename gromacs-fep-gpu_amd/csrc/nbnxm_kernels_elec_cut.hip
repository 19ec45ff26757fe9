/* Cluster-pair kernel instantiations, electrostatics flavour: elec_cut (see nbnxm_kernel_impl.h). */
#include "nbnxm_kernel_impl.h"
#include "nbnxm_kernels.h"

template<int VDW>
static NbKernelPtr pick(bool energy, bool fused, bool foreign)
{
    if (foreign) { return nbnxmKernel<ELK_CUT, false, VDW, true, true, true>; }
    if (energy) { return fused ? nbnxmKernel<ELK_CUT, false, VDW, true, true> : nbnxmKernel<ELK_CUT, false, VDW, true, false>; }
    return fused ? nbnxmKernel<ELK_CUT, false, VDW, false, true> : nbnxmKernel<ELK_CUT, false, VDW, false, false>;
}

NbKernelPtr nbKernelElecCut(int vdwKind, bool energy, bool fused, bool foreign)
{
    switch (vdwKind)
    {
        case VDK_CUT: return pick<VDK_CUT>(energy, fused, foreign);
        case VDK_COMB_GEOM: return pick<VDK_COMB_GEOM>(energy, fused, foreign);
        case VDK_COMB_LB: return pick<VDK_COMB_LB>(energy, fused, foreign);
        case VDK_FSWITCH: return pick<VDK_FSWITCH>(energy, fused, foreign);
        case VDK_PSWITCH: return pick<VDK_PSWITCH>(energy, fused, foreign);
        default: return nullptr;
    }
}
