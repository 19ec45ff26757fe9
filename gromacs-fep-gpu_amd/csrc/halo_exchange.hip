/*
 * GPU halo exchange of a domain-decomposed run over RCCL point-to-point (include/halo_hip.h).
 *
 *   domdec/gpuhaloexchange_impl_gpu.cpp:122-246   reinitHalo (index maps, buffers, offsets)
 *   domdec/gpuhaloexchange_impl_gpu.cpp:263-368   communicateHaloCoordinates / communicateHaloForces
 *   domdec/gpuhaloexchange_impl_gpu.cpp:370-511   the transfers (MPI on device pointers | peer copies + event handshake)
 *   domdec/gpuhaloexchange_impl_gpu.cu:62-116     packSendBufKernel / unpackRecvBufKernel
 *
 * Everything is queued on ONE stream; RCCL's send / receive are stream-ordered, so there is no host handshake per step
 * (the reference's peer-copy path exchanges event pointers with MPI_Sendrecv every step, :438-470).
 */
#include "halo_hip.h"

#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <rccl/rccl.h> /* types and prototypes only: the library is opened with dlopen */

#include "device_utils.h"
#include "nbnxm_gpu_internal.h"
#include "nbnxm_hip.h"

using namespace nbnxm_hip;

namespace
{

std::string g_haloError;

/* the few RCCL entry points used, resolved once */
struct Rccl
{
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*)                                                        = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int)                                 = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t)                                                           = nullptr;
    ncclResult_t (*GroupStart)()                                                                      = nullptr;
    ncclResult_t (*GroupEnd)()                                                                        = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)           = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)                 = nullptr;
    const char* (*GetErrorString)(ncclResult_t)                                                       = nullptr;
};

Rccl* rccl()
{
    static Rccl r;
    if (r.lib != nullptr) { return &r; }
    const char* names[] = { std::getenv("NBNXM_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names)
    {
        if (n == nullptr || n[0] == 0) { continue; }
        /* RTLD_NOLOAD first: the copy that is already in the process (PyTorch's), so that there is one RCCL, not two */
        void* lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (lib == nullptr) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
        if (lib != nullptr)
        {
            r.lib = lib;
            break;
        }
    }
    if (r.lib == nullptr)
    {
        g_haloError = std::string("cannot open librccl (") + dlerror() + ")";
        return nullptr;
    }
#define HALO_SYM(member, name)                                                      \
    r.member = reinterpret_cast<decltype(r.member)>(dlsym(r.lib, name));            \
    if (r.member == nullptr)                                                        \
    {                                                                               \
        g_haloError = std::string("librccl lacks ") + name;                         \
        r.lib       = nullptr;                                                      \
        return nullptr;                                                             \
    }
    HALO_SYM(GetUniqueId, "ncclGetUniqueId")
    HALO_SYM(CommInitRank, "ncclCommInitRank")
    HALO_SYM(CommDestroy, "ncclCommDestroy")
    HALO_SYM(GroupStart, "ncclGroupStart")
    HALO_SYM(GroupEnd, "ncclGroupEnd")
    HALO_SYM(Send, "ncclSend")
    HALO_SYM(Recv, "ncclRecv")
    HALO_SYM(GetErrorString, "ncclGetErrorString")
#undef HALO_SYM
    return &r;
}

#define HALO_RCCL_CHECK(expr)                                                                            \
    do                                                                                                   \
    {                                                                                                    \
        const ncclResult_t res_ = (expr);                                                                \
        if (res_ != ncclSuccess) { fatal(__FILE__, __LINE__, #expr, rccl()->GetErrorString(res_)); }     \
    } while (0)

constexpr int c_haloThreadsPerBlock = 256;
/* the local launch of a domain step in two parts (halo_gpu_domain_force_step) */
constexpr float c_defaultLocalPartFraction = 0.65F;

/* packed[i] = x[map[i]] + shift[shiftIndex[i]]: all destinations in one launch (the reference launches one
 * packSendBufKernel<usePbc> per pulse with one shift, gpuhaloexchange_impl_gpu.cu:62-88) */
__global__ void haloPackShiftedKernel(float3* __restrict__ packed, const float3* __restrict__ x, const int* __restrict__ map,
                                      const int* __restrict__ shiftIndex, const float3* __restrict__ shiftVectors, const int n)
{
    const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) { return; }
    const float3 v = x[map[i]];
    const float3 s = shiftVectors[shiftIndex[i]];
    packed[i]      = make_float3(v.x + s.x, v.y + s.y, v.z + s.z);
}

/* f[map[i]] (+)= packed[i]; an atom can be sent to several destinations, so it can occur several times in the map:
 * atomic adds (unpackRecvBufKernel, :90-116, runs once per pulse and needs none) */
template<bool accumulate>
__global__ void haloUnpackForcesKernel(float* __restrict__ f, const float3* __restrict__ packed, const int* __restrict__ map, const int n)
{
    const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) { return; }
    const float3 v = packed[i];
    float*       d = f + 3 * static_cast<size_t>(map[i]);
    if (accumulate)
    {
        atomicAdd(d + 0, v.x);
        atomicAdd(d + 1, v.y);
        atomicAdd(d + 2, v.z);
    }
    else
    {
        d[0] = v.x;
        d[1] = v.y;
        d[2] = v.z;
    }
}

/* the two force-buffer passes of a domain step, each fused into one launch (every launch on a stream costs its duration plus a
 * gap of a few microseconds):
 *   A (non-local stream, behind the non-local kernel): home rows of f = 0, halo rows of f = nbnxm forces of the halo slots
 *   B (local stream, behind the local kernel and the force halo): home rows += nbnxm forces of the home slots, and
 *     f[sendMap[j]] += what the other ranks computed on this rank's atoms — all with atomic adds, the rows are shared */
__global__ void domainHaloRowsKernel(float3* __restrict__ f, const float3* __restrict__ nbnxmForce, const int* __restrict__ cell,
                                     const int numHome, const int numAtoms)
{
    const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= numAtoms) { return; }
    f[i] = (i < numHome) ? make_float3(0.0F, 0.0F, 0.0F) : nbnxmForce[cell[i]];
}

__global__ void domainHomeRowsKernel(float* __restrict__ f, const float3* __restrict__ nbnxmForce, const int* __restrict__ cell,
                                     const int numHome, const float3* __restrict__ received, const int* __restrict__ sendMap,
                                     const int numReceived)
{
    const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
    float3    v;
    float*    d;
    if (i < numHome)
    {
        v = nbnxmForce[cell[i]];
        d = f + 3 * static_cast<size_t>(i);
    }
    else if (i < numHome + numReceived)
    {
        v = received[i - numHome];
        d = f + 3 * static_cast<size_t>(sendMap[i - numHome]);
    }
    else { return; }
    atomicAdd(d + 0, v.x);
    atomicAdd(d + 1, v.y);
    atomicAdd(d + 2, v.z);
}

} // namespace

struct HaloGpu
{
    ncclComm_t  comm   = nullptr;
    int         rank   = 0;
    int         nranks = 1;
    hipStream_t stream = nullptr;
    float3*     d_x    = nullptr;
    float3*     d_f    = nullptr;
    int         numHome = 0;
    /* send side */
    std::vector<int> sendPeer, sendOffset;
    int              numSendAtoms = 0;
    int*             d_sendMap        = nullptr;
    int*             d_sendShiftIndex = nullptr;
    float3*          d_shiftVectors   = nullptr;
    float3*          d_sendBuf        = nullptr; /* packed coordinates out; received forces in */
    int              sendAlloc = 0, shiftAlloc = 0;
    /* recv side */
    std::vector<int> recvPeer, recvAtomOffset, recvCount;
    hipEvent_t       xReady = nullptr, fReady = nullptr;
    /* HALO_GPU_HOST_TIMING: host microseconds spent queueing the parts of halo_gpu_domain_force_step */
    double                                hostTimingUs[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    long                                  hostTimingSteps = 0;
    /* the local launch of halo_gpu_domain_force_step in one or two parts */
    int   localParts        = 1;
    float localPartFraction = c_defaultLocalPartFraction;
    std::chrono::steady_clock::time_point hostTimingLast;
};

extern "C"
{

const char* halo_gpu_last_error(void)
{
    return g_haloError.c_str();
}

int halo_gpu_get_unique_id(void* uniqueId)
{
    Rccl* r = rccl();
    if (r == nullptr) { return 1; }
    static_assert(sizeof(ncclUniqueId) == HALO_GPU_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    if (r->GetUniqueId(&id) != ncclSuccess)
    {
        g_haloError = "ncclGetUniqueId failed";
        return 2;
    }
    std::memcpy(uniqueId, &id, sizeof(id));
    return 0;
}

HaloGpu* halo_gpu_create(const void* uniqueId, int rank, int nranks, void* stream)
{
    Rccl* r = rccl();
    if (r == nullptr) { return nullptr; }
    auto* h   = new HaloGpu;
    h->rank   = rank;
    h->nranks = nranks;
    h->stream = static_cast<hipStream_t>(stream);
    ncclUniqueId id;
    std::memcpy(&id, uniqueId, sizeof(id));
    const ncclResult_t res = r->CommInitRank(&h->comm, nranks, id, rank);
    if (res != ncclSuccess)
    {
        g_haloError = std::string("ncclCommInitRank: ") + r->GetErrorString(res);
        delete h;
        return nullptr;
    }
    NBNXM_HIP_CHECK(hipEventCreateWithFlags(&h->xReady, hipEventDisableTiming));
    NBNXM_HIP_CHECK(hipEventCreateWithFlags(&h->fReady, hipEventDisableTiming));
    /* the local launch of a domain step in two parts: on by default where the exchanges leave the device (see halo_gpu_domain_force_step) */
    h->localParts = (nranks > 1) ? 2 : 1;
    if (const char* env = std::getenv("HALO_GPU_LOCAL_PARTS")) { h->localParts = (std::atoi(env) == 2) ? 2 : 1; }
    if (const char* env = std::getenv("HALO_GPU_LOCAL_PART_FRACTION"))
    {
        const float v = static_cast<float>(std::atof(env));
        if (v > 0.05F && v < 0.95F) { h->localPartFraction = v; }
    }
    return h;
}

void halo_gpu_free(HaloGpu* h)
{
    if (h == nullptr) { return; }
    (void)hipStreamSynchronize(h->stream);
    if (h->hostTimingSteps > 0)
    {
        static const char* names[9] = { "halo x (pack, send/recv group)", "clear outputs", "x -> xq local", "local kernel", "x -> xq non-local",
                                        "non-local kernel", "halo rows kernel", "halo f (send/recv group)", "event, wait, home rows kernel" };
        std::fprintf(stderr, "halo_gpu: host microseconds per step spent queueing (%ld steps):\n", h->hostTimingSteps);
        for (int i = 0; i < 9; i++) { std::fprintf(stderr, "  %-34s %7.2f\n", names[i], h->hostTimingUs[i] / h->hostTimingSteps); }
    }
    if (h->comm != nullptr) { (void)rccl()->CommDestroy(h->comm); }
    (void)hipFree(h->d_sendMap);
    (void)hipFree(h->d_sendShiftIndex);
    (void)hipFree(h->d_shiftVectors);
    (void)hipFree(h->d_sendBuf);
    if (h->xReady) { (void)hipEventDestroy(h->xReady); }
    if (h->fReady) { (void)hipEventDestroy(h->fReady); }
    delete h;
}

void halo_gpu_reinit(HaloGpu* h, void* d_x, void* d_f, int numHome, int numSend, const int* sendPeer, const int* sendOffset,
                     const int* sendMap, const int* sendShiftIndex, int numShiftVectors, const float* shiftVectors, int numRecv,
                     const int* recvPeer, const int* recvAtomOffset, const int* recvCount)
{
    h->d_x     = static_cast<float3*>(d_x);
    h->d_f     = static_cast<float3*>(d_f);
    h->numHome = numHome;
    h->sendPeer.assign(sendPeer, sendPeer + numSend);
    h->sendOffset.assign(sendOffset, sendOffset + numSend + 1);
    h->numSendAtoms = h->sendOffset[numSend];
    for (int k = 0; k < numSend; k++)
    {
        NBNXM_ASSERT(sendPeer[k] >= 0 && sendPeer[k] < h->nranks && sendOffset[k] <= sendOffset[k + 1], "bad send link");
    }
    for (int i = 0; i < h->numSendAtoms; i++)
    {
        NBNXM_ASSERT(sendMap[i] >= 0 && sendMap[i] < numHome, "only home atoms are sent");
        NBNXM_ASSERT(sendShiftIndex[i] >= 0 && sendShiftIndex[i] < numShiftVectors, "shift index out of range");
    }
    h->recvPeer.assign(recvPeer, recvPeer + numRecv);
    h->recvAtomOffset.assign(recvAtomOffset, recvAtomOffset + numRecv);
    h->recvCount.assign(recvCount, recvCount + numRecv);
    for (int k = 0; k < numRecv; k++)
    {
        NBNXM_ASSERT(recvPeer[k] >= 0 && recvPeer[k] < h->nranks && recvAtomOffset[k] >= numHome && recvCount[k] >= 0, "bad receive link");
    }
    if (h->numSendAtoms > h->sendAlloc)
    {
        (void)hipFree(h->d_sendMap);
        (void)hipFree(h->d_sendShiftIndex);
        (void)hipFree(h->d_sendBuf);
        h->sendAlloc = static_cast<int>(h->numSendAtoms * 1.2) + 1024;
        NBNXM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_sendMap), sizeof(int) * h->sendAlloc));
        NBNXM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_sendShiftIndex), sizeof(int) * h->sendAlloc));
        NBNXM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_sendBuf), sizeof(float3) * h->sendAlloc));
    }
    if (numShiftVectors > h->shiftAlloc)
    {
        (void)hipFree(h->d_shiftVectors);
        h->shiftAlloc = numShiftVectors + 32;
        NBNXM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_shiftVectors), sizeof(float3) * h->shiftAlloc));
    }
    /* search steps are rare: plain synchronous copies */
    NBNXM_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (h->numSendAtoms > 0)
    {
        NBNXM_HIP_CHECK(hipMemcpy(h->d_sendMap, sendMap, sizeof(int) * h->numSendAtoms, hipMemcpyHostToDevice));
        NBNXM_HIP_CHECK(hipMemcpy(h->d_sendShiftIndex, sendShiftIndex, sizeof(int) * h->numSendAtoms, hipMemcpyHostToDevice));
    }
    if (numShiftVectors > 0)
    {
        NBNXM_HIP_CHECK(hipMemcpy(h->d_shiftVectors, shiftVectors, sizeof(float) * 3 * numShiftVectors, hipMemcpyHostToDevice));
    }
}

void halo_gpu_communicate_coordinates(HaloGpu* h, void* dependencyEvent)
{
    Rccl*       r = rccl();
    hipStream_t s = h->stream;
    if (dependencyEvent != nullptr) { NBNXM_HIP_CHECK(hipStreamWaitEvent(s, static_cast<hipEvent_t>(dependencyEvent), 0)); }
    if (h->numSendAtoms > 0)
    {
        hipLaunchKernelGGL(haloPackShiftedKernel, dim3((h->numSendAtoms + c_haloThreadsPerBlock - 1) / c_haloThreadsPerBlock),
                           dim3(c_haloThreadsPerBlock), 0, s, h->d_sendBuf, h->d_x, h->d_sendMap, h->d_sendShiftIndex, h->d_shiftVectors,
                           h->numSendAtoms);
        NBNXM_HIP_CHECK(hipGetLastError());
    }
    /* one group: every send and every receive of this rank; coordinates arrive in place (rows of d_x) */
    HALO_RCCL_CHECK(r->GroupStart());
    for (size_t k = 0; k < h->recvPeer.size(); k++)
    {
        if (h->recvCount[k] > 0)
        {
            HALO_RCCL_CHECK(r->Recv(h->d_x + h->recvAtomOffset[k], static_cast<size_t>(3) * h->recvCount[k], ncclFloat, h->recvPeer[k], h->comm, s));
        }
    }
    for (size_t k = 0; k < h->sendPeer.size(); k++)
    {
        const int n = h->sendOffset[k + 1] - h->sendOffset[k];
        if (n > 0) { HALO_RCCL_CHECK(r->Send(h->d_sendBuf + h->sendOffset[k], static_cast<size_t>(3) * n, ncclFloat, h->sendPeer[k], h->comm, s)); }
    }
    HALO_RCCL_CHECK(r->GroupEnd());
    NBNXM_HIP_CHECK(hipEventRecord(h->xReady, s));
}

/* the transfers of the force halo alone: halo rows out, what the others computed on this rank's atoms into the send buffer */
static void exchangeForces(HaloGpu* h)
{
    Rccl*       r = rccl();
    hipStream_t s = h->stream;
    /* the reverse of the coordinate exchange: the halo rows of d_f go to their owners as they are (contiguous, no pack),
     * what the others computed on this rank's atoms arrives in the send buffer, in the order of the send map */
    HALO_RCCL_CHECK(r->GroupStart());
    for (size_t k = 0; k < h->sendPeer.size(); k++)
    {
        const int n = h->sendOffset[k + 1] - h->sendOffset[k];
        if (n > 0) { HALO_RCCL_CHECK(r->Recv(h->d_sendBuf + h->sendOffset[k], static_cast<size_t>(3) * n, ncclFloat, h->sendPeer[k], h->comm, s)); }
    }
    for (size_t k = 0; k < h->recvPeer.size(); k++)
    {
        if (h->recvCount[k] > 0)
        {
            HALO_RCCL_CHECK(r->Send(h->d_f + h->recvAtomOffset[k], static_cast<size_t>(3) * h->recvCount[k], ncclFloat, h->recvPeer[k], h->comm, s));
        }
    }
    HALO_RCCL_CHECK(r->GroupEnd());
}

void halo_gpu_communicate_forces(HaloGpu* h, int accumulate, void* dependencyEvent)
{
    hipStream_t s = h->stream;
    if (dependencyEvent != nullptr) { NBNXM_HIP_CHECK(hipStreamWaitEvent(s, static_cast<hipEvent_t>(dependencyEvent), 0)); }
    exchangeForces(h);
    if (h->numSendAtoms > 0)
    {
        const dim3 grid((h->numSendAtoms + c_haloThreadsPerBlock - 1) / c_haloThreadsPerBlock);
        if (accumulate)
        {
            hipLaunchKernelGGL(haloUnpackForcesKernel<true>, grid, dim3(c_haloThreadsPerBlock), 0, s, reinterpret_cast<float*>(h->d_f),
                               h->d_sendBuf, h->d_sendMap, h->numSendAtoms);
        }
        else
        {
            hipLaunchKernelGGL(haloUnpackForcesKernel<false>, grid, dim3(c_haloThreadsPerBlock), 0, s, reinterpret_cast<float*>(h->d_f),
                               h->d_sendBuf, h->d_sendMap, h->numSendAtoms);
        }
        NBNXM_HIP_CHECK(hipGetLastError());
    }
    NBNXM_HIP_CHECK(hipEventRecord(h->fReady, s));
}

void* halo_gpu_coordinates_ready_event(HaloGpu* h)
{
    return h->xReady;
}

void* halo_gpu_forces_ready_event(HaloGpu* h)
{
    return h->fReady;
}

long long halo_gpu_bytes_per_step(const HaloGpu* h)
{
    long long n = h->numSendAtoms;
    for (int c : h->recvCount) { n += c; }
    return 12LL * n;
}

/* One domain's force step with the two-locality schedule of mdlib/sim_util.cpp:1783-1924 (do_force with GPU halo exchange and GPU
 * buffer ops), host side in C++ as in the reference:
 *   non-local stream: halo x -> x to xq (halo slots) -> non-local kernel -> halo rows of f to atom order -> halo f (added to home rows)
 *   local stream:     clear -> x to xq (home slots) -> local kernel [beside all of the above] -> wait for the non-local stream
 *                     -> home rows of f += nbnxm forces
 * The object's stream must be the non-local stream of nb. */
void halo_gpu_domain_force_step(HaloGpu* h, NbnxmGpu* nb, const nbnxm_step_workload_t* stepWork, int numHomeSlots, int numSlots, int numAtoms,
                                void* coordinatesReadyEvent)
{
    hipStream_t sLocal    = static_cast<hipStream_t>(nbnxm_gpu_get_stream(nb, NBNXM_LOCAL));
    hipStream_t sNonLocal = static_cast<hipStream_t>(nbnxm_gpu_get_stream(nb, NBNXM_NONLOCAL));
    NBNXM_ASSERT(sNonLocal == h->stream, "the halo object must have been created on the non-local stream of the non-bonded object");
    /* the halo coordinates leave first: the pack and the RCCL kernel find the device idle at the start of a step; queued behind the
     * local kernel — which fills every wave slot — the RCCL kernel would wait for slots, and the peers with it (on one GPU the order
     * makes no difference: 0.141 ms either way) */
    /* diagnostics (HALO_GPU_HOST_TIMING=1): host time spent queueing each part, printed by halo_gpu_free */
    static const bool s_hostTiming = (std::getenv("HALO_GPU_HOST_TIMING") != nullptr);
    auto              tick         = [&](int part) {
        if (s_hostTiming)
        {
            const auto now = std::chrono::steady_clock::now();
            h->hostTimingUs[part] += std::chrono::duration<double, std::micro>(now - h->hostTimingLast).count();
            h->hostTimingLast = now;
        }
    };
    if (s_hostTiming)
    {
        h->hostTimingLast = std::chrono::steady_clock::now();
        h->hostTimingSteps++;
    }
    /* The local kernel takes every wave slot until its balanced ranges retire together, so the non-local kernel runs behind it and
     * the force halo is exposed.  In two parts — most of the local list beside the coordinate halo, the rest behind the non-local
     * kernel (high-priority stream) beside the force halo — the step is max(L1, halo x) + non-local + max(L2, halo f)
     * (read at halo_gpu_create: HALO_GPU_LOCAL_PARTS=1 / 2 switches it off / on — the default is on with more than one rank, where the
     * exchanges cross xGMI —, HALO_GPU_LOCAL_PART_FRACTION sets L1's share of the local work).  Lists too short for two sets of
     * one range per wave slot run as one launch. */
    const bool twoParts = (h->localParts == 2 && (h->numSendAtoms > 0 || !h->recvPeer.empty()));
    nbnxm_gpu_set_local_launch_parts(nb, twoParts ? 2 : 1, h->localPartFraction);
    halo_gpu_communicate_coordinates(h, coordinatesReadyEvent);
    tick(0);
    nbnxm_gpu_clear_outputs(nb, stepWork->computeVirial);
    tick(1);
    nbnxm_gpu_x_to_nbat_x(nb, h->d_x, coordinatesReadyEvent, NBNXM_LOCAL, 0, numHomeSlots, 1);
    tick(2);
    if (twoParts) { nbnxm_gpu_launch_kernel_part(nb, stepWork, NBNXM_LOCAL, 1); }
    else { nbnxm_gpu_launch_kernel(nb, stepWork, NBNXM_LOCAL); }
    tick(3);
    nbnxm_gpu_x_to_nbat_x(nb, h->d_x, nullptr, NBNXM_NONLOCAL, numHomeSlots, numSlots, 1);
    tick(4);
    nbnxm_gpu_launch_kernel(nb, stepWork, NBNXM_NONLOCAL);
    tick(5);
    NBNXM_ASSERT(nb->reductionAtomStart == 0 && nb->reductionNumAtoms >= numAtoms, "the cell map must cover home and halo atoms");
    /* pass A: home rows zeroed, halo rows from the non-local kernel (one launch on the non-local stream) */
    hipLaunchKernelGGL(domainHaloRowsKernel, dim3((numAtoms + c_haloThreadsPerBlock - 1) / c_haloThreadsPerBlock), dim3(c_haloThreadsPerBlock), 0,
                       sNonLocal, h->d_f, reinterpret_cast<const float3*>(nb->atdat->f), nb->cell, h->numHome, numAtoms);
    NBNXM_HIP_CHECK(hipGetLastError());
    tick(6);
    exchangeForces(h);
    tick(7);
    NBNXM_HIP_CHECK(hipEventRecord(h->fReady, sNonLocal));
    if (twoParts) { nbnxm_gpu_launch_kernel_part(nb, stepWork, NBNXM_LOCAL, 2); }
    /* pass B: behind the local kernel (stream order) and the arrival of the force halo (event) */
    NBNXM_HIP_CHECK(hipStreamWaitEvent(sLocal, h->fReady, 0));
    const int n = h->numHome + h->numSendAtoms;
    hipLaunchKernelGGL(domainHomeRowsKernel, dim3((n + c_haloThreadsPerBlock - 1) / c_haloThreadsPerBlock), dim3(c_haloThreadsPerBlock), 0, sLocal,
                       reinterpret_cast<float*>(h->d_f), reinterpret_cast<const float3*>(nb->atdat->f), nb->cell, h->numHome, h->d_sendBuf,
                       h->d_sendMap, h->numSendAtoms);
    NBNXM_HIP_CHECK(hipGetLastError());
    tick(8);
}

void halo_gpu_pack_shifted(void* stream, const void* d_x, const int* d_map, const int* d_shiftIndex, int n, const float* d_shiftVectors,
                           void* d_packed)
{
    if (n <= 0) { return; }
    hipLaunchKernelGGL(haloPackShiftedKernel, dim3((n + c_haloThreadsPerBlock - 1) / c_haloThreadsPerBlock), dim3(c_haloThreadsPerBlock), 0,
                       static_cast<hipStream_t>(stream), static_cast<float3*>(d_packed), static_cast<const float3*>(d_x), d_map, d_shiftIndex,
                       reinterpret_cast<const float3*>(d_shiftVectors), n);
    NBNXM_HIP_CHECK(hipGetLastError());
}

} // extern "C"
