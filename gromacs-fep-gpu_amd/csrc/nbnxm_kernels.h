/*
 * Kernel selection tables (the role of select_nbnxn_kernel / select_nbnxn_fep_kernel /
 * select_nbnxn_foreign_fep_kernel, nbnxm/cuda/nbnxm_cuda.cu:416-591).  Flavours are C++ template
 * instantiations, one translation unit per electrostatics kind so that they build in parallel.
 */
#ifndef NBNXM_KERNELS_H
#define NBNXM_KERNELS_H

#include "nbnxm_hip_types.h"

using NbKernelPtr = void (*)(const NbWorkDesc*, int, int, NBAtomDataGpu, NBParamGpu, gpu_plist, int, const nbnxn_sci_t*, const nbnxn_cj_packed_t*,
                             const nbnxn_excl_t*, const float4*, const int*, const float2*, const unsigned*,
                             const unsigned*, int, int, int, int, float4*, int, float4*, int, int, gpu_feplist);
using FepClusterKernelPtr = void (*)(NBAtomDataGpu, NBParamGpu, gpu_plist, int, const nbnxn_sci_t*, const nbnxn_cj_packed_t*,
                                     const nbnxn_excl_t*, const float4*, const int*, const float2*, const unsigned*, int);
using FepKernelPtr   = void (*)(NBAtomDataGpu, NBParamGpu, gpu_feplist, int);
using PruneKernelPtr = void (*)(NBAtomDataGpu, NBParamGpu, gpu_plist, int, int);

/* vdwKind: VDK_* of nbnxm_device_helpers.h; returns nullptr for an unsupported flavour */
NbKernelPtr nbKernelElecCut(int vdwKind, bool energy, bool fused);
FepClusterKernelPtr nbKernelElecCutFepCluster(int vdwKind, bool energy, bool foreign);
NbKernelPtr nbKernelElecRF(int vdwKind, bool energy, bool fused);
FepClusterKernelPtr nbKernelElecRFFepCluster(int vdwKind, bool energy, bool foreign);
NbKernelPtr nbKernelElecEwaldAna(int vdwKind, bool energy, bool fused);
FepClusterKernelPtr nbKernelElecEwaldAnaFepCluster(int vdwKind, bool energy, bool foreign);
NbKernelPtr nbKernelElecEwaldTab(int vdwKind, bool energy, bool fused);
FepClusterKernelPtr nbKernelElecEwaldTabFepCluster(int vdwKind, bool energy, bool foreign);
NbKernelPtr nbKernelElecEwaldAnaTwin(int vdwKind, bool energy, bool fused);
FepClusterKernelPtr nbKernelElecEwaldAnaTwinFepCluster(int vdwKind, bool energy, bool foreign);
NbKernelPtr nbKernelElecEwaldTabTwin(int vdwKind, bool energy, bool fused);
FepClusterKernelPtr nbKernelElecEwaldTabTwinFepCluster(int vdwKind, bool energy, bool foreign);

NbKernelPtr    selectNbKernel(int elecType, int vdwType, bool energy, bool fused);
/* fused mode: the kernel of the perturbed cluster pairs; foreign: its dH/dl-step flavour (implies energy) */
FepClusterKernelPtr selectFepClusterKernel(int elecType, int vdwType, bool energy, bool foreign);
FepKernelPtr   selectFepKernel(int elecType, int vdwType, bool energy);
FepKernelPtr   selectFepForeignKernel(int elecType, int vdwType);
PruneKernelPtr selectPruneKernel(bool haveFreshList);
/* waves per SIMD the flavour is compiled for (c_nbWavesPerEu): decides which work partition it runs on */
int nbKernelWavesPerEu(int vdwType, bool energy, bool fused);

#endif
