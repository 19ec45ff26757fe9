/*
 * Coordinate / force buffer operations around the non-bonded kernels, and the pack / unpack kernels of the GPU halo
 * exchange.  All of them are gather / scatter copies: HBM-bound, one thread per element, 64-lane coalesced on the
 * contiguous side.
 *
 *   nbnxm/cuda/nbnxm_gpu_buffer_ops_internal.cu:66-147   x (atom order) -> xq (grid order)
 *   nbnxm/nbnxm_gpu_buffer_ops.cpp:62-98                 its host wrapper (event wait, non-local dependency)
 *   nbnxm/nbnxm_gpu_data_mgmt.cpp:1400-1500              nbnxn_gpu_init_x_to_nbat_x
 *   nbnxm/nbnxm_gpu_data_mgmt.cpp:1093-1111,1305-1327    setupGpuShortRangeWork, nbnxnInsertNonlocalGpuDependency
 *   mdlib/gpuforcereduction_impl_internal.cu:57-127      f (grid order) -> f (atom order), + rvec force, accumulate
 *   domdec/gpuhaloexchange_impl_gpu.cu:62-116            packSendBufKernel / unpackRecvBufKernel
 */
#include "nbnxm_gpu_internal.h"

namespace
{

constexpr int c_bufOpsThreadsPerBlock = 256;

/* One thread per grid slot.  The reference walks (column, atom-in-column) with a 2D grid and leaves the filler
 * slots alone (its filler branch is unreachable, :96-108); the flat form does the same: fillers have index -1. */
__global__ void nbnxmXToXqKernel(float4* __restrict__ xq, const float3* __restrict__ x, const int* __restrict__ atomIndex,
                                 const int slotBegin, const int slotEnd)
{
    const int slot = slotBegin + static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
    if (slot >= slotEnd) { return; }
    const int a = atomIndex[slot];
    if (a < 0) { return; }
    /* x, y, z only: the charge in .w was set by gpu_init_atomdata */
    *reinterpret_cast<float3*>(&xq[slot]) = x[a];
}

template<bool addRvecForce, bool accumulate>
__global__ void nbnxmForceReductionKernel(const float3* __restrict__ nbnxmForce, const float3* __restrict__ rvecForceToAdd,
                                          float3* fTotal, const int* __restrict__ cell, const int numAtoms)
{
    const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= numAtoms) { return; }
    float3 t = nbnxmForce[cell[i]];
    if (accumulate)
    {
        const float3 b = fTotal[i];
        t              = make_float3(b.x + t.x, b.y + t.y, b.z + t.z); /* base + nbnxm, then + rvec: the reference's order */
    }
    if (addRvecForce)
    {
        const float3 r = rvecForceToAdd[i];
        t              = make_float3(t.x + r.x, t.y + r.y, t.z + r.z);
    }
    fTotal[i] = t;
}

template<bool usePbc>
__global__ void haloPackKernel(float3* __restrict__ packed, const float3* __restrict__ data, const int* __restrict__ map,
                               const int mapSize, const float3 shift)
{
    const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= mapSize) { return; }
    float3 v = data[map[i]];
    if (usePbc) { v = make_float3(v.x + shift.x, v.y + shift.y, v.z + shift.z); }
    packed[i] = v;
}

template<bool accumulate>
__global__ void haloUnpackKernel(float3* __restrict__ data, const float3* __restrict__ packed, const int* __restrict__ map,
                                 const int mapSize)
{
    const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= mapSize) { return; }
    const float3 v = packed[i];
    float3*      d = &data[map[i]];
    if (accumulate)
    {
        const float3 o = *d;
        *d             = make_float3(o.x + v.x, o.y + v.y, o.z + v.z);
    }
    else { *d = v; }
}

inline dim3 gridFor(int n)
{
    return dim3((n + c_bufOpsThreadsPerBlock - 1) / c_bufOpsThreadsPerBlock);
}

} // namespace

extern "C"
{

void nbnxm_gpu_init_x_to_nbat_x(NbnxmGpu* nb, int numAtomIndices, const int* atomIndices)
{
    NBNXM_ASSERT(numAtomIndices == nb->atdat->numAtoms, "atomIndices must cover every grid slot (numAtoms of gpu_init_atomdata)");
    if (numAtomIndices > nb->atomIndices_nalloc)
    {
        freeDeviceBuffer(&nb->atomIndices);
        nb->atomIndices_nalloc = static_cast<int>(numAtomIndices * 1.2) + 1024;
        allocateDeviceBuffer(&nb->atomIndices, nb->atomIndices_nalloc);
    }
    nb->h_atomIndices.resize(numAtomIndices);
    std::memcpy(nb->h_atomIndices.data, atomIndices, sizeof(int) * numAtomIndices);
    copyToDeviceBuffer(&nb->atomIndices, nb->h_atomIndices.data, 0, numAtomIndices, nb->deviceStreams[0].stream, true);
    nb->atomIndicesSize = numAtomIndices;
}

void nbnxm_gpu_insert_nonlocal_dependency(NbnxmGpu* nb, int iloc)
{
    if (!nb->bUseTwoStreams) { return; }
    hipStream_t s = nb->deviceStreams[iloc].stream;
    if (iloc == NBNXM_LOCAL) { NBNXM_HIP_CHECK(hipEventRecord(nb->misc_ops_and_local_H2D_done, s)); }
    else { NBNXM_HIP_CHECK(hipStreamWaitEvent(s, nb->misc_ops_and_local_H2D_done, 0)); }
}

void nbnxm_gpu_x_to_nbat_x(NbnxmGpu* nb, const void* d_x, void* xReadyOnDevice, int atomLocality, int slotBegin, int slotEnd,
                           int mustInsertNonLocalDependency)
{
    const int iloc = atomLocality;
    NBNXM_ASSERT(iloc == NBNXM_LOCAL || (iloc == NBNXM_NONLOCAL && nb->bUseTwoStreams), "bad locality");
    NBNXM_ASSERT(slotBegin >= 0 && slotBegin <= slotEnd && slotEnd <= nb->atomIndicesSize,
                 "grid slots outside the uploaded atomIndices (call nbnxm_gpu_init_x_to_nbat_x after each search)");
    hipStream_t s = nb->deviceStreams[iloc].stream;
    if (xReadyOnDevice != nullptr) { NBNXM_HIP_CHECK(hipStreamWaitEvent(s, static_cast<hipEvent_t>(xReadyOnDevice), 0)); }
    if (slotEnd > slotBegin)
    {
        NBNXM_ASSERT(d_x != nullptr, "coordinate buffer missing");
        hipLaunchKernelGGL(nbnxmXToXqKernel, gridFor(slotEnd - slotBegin), dim3(c_bufOpsThreadsPerBlock), 0, s, nb->atdat->xq,
                           static_cast<const float3*>(d_x), nb->atomIndices, slotBegin, slotEnd);
        NBNXM_HIP_CHECK(hipGetLastError());
    }
    if (mustInsertNonLocalDependency) { nbnxm_gpu_insert_nonlocal_dependency(nb, iloc); }
}

void nbnxm_gpu_setup_short_range_work(NbnxmGpu* nb, int haveListedForcesGpuInteractions, int iloc)
{
    nb->haveWork[iloc] = (nb->plist[iloc] != nullptr && nb->plist[iloc]->nsci != 0) || haveListedForcesGpuInteractions != 0;
}

void nbnxm_gpu_force_reduction_reinit(NbnxmGpu* nb, int numAtoms, const int* cell, int atomStart, int accumulate)
{
    NBNXM_ASSERT(numAtoms >= 0 && atomStart >= 0, "bad atom range");
    for (int i = 0; i < numAtoms; i++)
    {
        NBNXM_ASSERT(cell[i] >= 0 && cell[i] < nb->atdat->numAtoms, "cell index outside the nbnxm force buffer");
    }
    if (numAtoms > nb->cell_nalloc)
    {
        freeDeviceBuffer(&nb->cell);
        nb->cell_nalloc = static_cast<int>(numAtoms * 1.2) + 1024;
        allocateDeviceBuffer(&nb->cell, nb->cell_nalloc);
    }
    nb->h_cell.resize(numAtoms);
    if (numAtoms) { std::memcpy(nb->h_cell.data, cell, sizeof(int) * numAtoms); }
    copyToDeviceBuffer(&nb->cell, nb->h_cell.data, 0, numAtoms, nb->deviceStreams[0].stream, true);
    nb->reductionNumAtoms   = numAtoms;
    nb->reductionAtomStart  = atomStart;
    nb->reductionAccumulate = accumulate != 0;
}

void nbnxm_gpu_force_reduction_execute(NbnxmGpu* nb, void* d_baseForce, const void* d_rvecForceToAdd, void* stream)
{
    const int n = nb->reductionNumAtoms;
    if (n == 0) { return; }
    NBNXM_ASSERT(d_baseForce != nullptr, "base force buffer missing");
    hipStream_t   s    = stream ? static_cast<hipStream_t>(stream) : nb->deviceStreams[0].stream;
    float3*       base = static_cast<float3*>(d_baseForce) + nb->reductionAtomStart;
    const float3* rvec = d_rvecForceToAdd ? static_cast<const float3*>(d_rvecForceToAdd) + nb->reductionAtomStart : nullptr;
    const bool    add  = rvec != nullptr;
    auto          k    = add ? (nb->reductionAccumulate ? nbnxmForceReductionKernel<true, true> : nbnxmForceReductionKernel<true, false>)
                             : (nb->reductionAccumulate ? nbnxmForceReductionKernel<false, true> : nbnxmForceReductionKernel<false, false>);
    hipLaunchKernelGGL(k, gridFor(n), dim3(c_bufOpsThreadsPerBlock), 0, s, reinterpret_cast<const float3*>(nb->atdat->f), rvec, base,
                       nb->cell, n);
    NBNXM_HIP_CHECK(hipGetLastError());
}

void nbnxm_gpu_force_reduction_execute_range(NbnxmGpu* nb, void* d_baseForce, int atomBegin, int atomEnd, int accumulate, void* stream)
{
    NBNXM_ASSERT(nb->reductionAtomStart == 0 && atomBegin >= 0 && atomBegin <= atomEnd && atomEnd <= nb->reductionNumAtoms,
                 "range outside the cell map of nbnxm_gpu_force_reduction_reinit (atomStart 0)");
    const int n = atomEnd - atomBegin;
    if (n == 0) { return; }
    hipStream_t s    = stream ? static_cast<hipStream_t>(stream) : nb->deviceStreams[0].stream;
    float3*     base = static_cast<float3*>(d_baseForce) + atomBegin;
    auto        k    = accumulate ? nbnxmForceReductionKernel<false, true> : nbnxmForceReductionKernel<false, false>;
    hipLaunchKernelGGL(k, gridFor(n), dim3(c_bufOpsThreadsPerBlock), 0, s, reinterpret_cast<const float3*>(nb->atdat->f), nullptr, base,
                       nb->cell + atomBegin, n);
    NBNXM_HIP_CHECK(hipGetLastError());
}

void nbnxm_gpu_halo_pack_x(void* stream, const void* d_x, const int* d_map, int mapSize, const float* coordinateShift, void* d_sendBuf)
{
    if (mapSize <= 0) { return; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (coordinateShift != nullptr)
    {
        const float3 sh = make_float3(coordinateShift[0], coordinateShift[1], coordinateShift[2]);
        hipLaunchKernelGGL(haloPackKernel<true>, gridFor(mapSize), dim3(c_bufOpsThreadsPerBlock), 0, s, static_cast<float3*>(d_sendBuf),
                           static_cast<const float3*>(d_x), d_map, mapSize, sh);
    }
    else
    {
        hipLaunchKernelGGL(haloPackKernel<false>, gridFor(mapSize), dim3(c_bufOpsThreadsPerBlock), 0, s, static_cast<float3*>(d_sendBuf),
                           static_cast<const float3*>(d_x), d_map, mapSize, make_float3(0.0F, 0.0F, 0.0F));
    }
    NBNXM_HIP_CHECK(hipGetLastError());
}

void nbnxm_gpu_halo_unpack_f(void* stream, void* d_f, const int* d_map, int mapSize, const void* d_recvBuf, int accumulate)
{
    if (mapSize <= 0) { return; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (accumulate)
    {
        hipLaunchKernelGGL(haloUnpackKernel<true>, gridFor(mapSize), dim3(c_bufOpsThreadsPerBlock), 0, s, static_cast<float3*>(d_f),
                           static_cast<const float3*>(d_recvBuf), d_map, mapSize);
    }
    else
    {
        hipLaunchKernelGGL(haloUnpackKernel<false>, gridFor(mapSize), dim3(c_bufOpsThreadsPerBlock), 0, s, static_cast<float3*>(d_f),
                           static_cast<const float3*>(d_recvBuf), d_map, mapSize);
    }
    NBNXM_HIP_CHECK(hipGetLastError());
}

} // extern "C"
