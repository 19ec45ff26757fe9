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
 *
 * Second transport, selected by the unique id (halo_gpu_get_unique_id_ex(.., HALO_GPU_TRANSPORT_PEER_COPY)): in-process peer copies,
 * the reference's thread-MPI path (:438-511): the ranks are host threads of ONE process (each with its own HaloGpu, streams and,
 * on a node, its own device), the receiver copies device to device out of the sender's buffer (hipMemcpyAsync / hipMemcpyPeerAsync)
 * behind an event the sender recorded, and the event handles travel through a mailbox in host memory (a mutex and a condition
 * variable stand in for the reference's MPI_Sendrecv of event pointers).  It runs the same pack / unpack kernels, link offsets and
 * two-locality schedule as the RCCL transport, with any number of ranks on one GPU.
 */
#include "halo_hip.h"

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
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

/* merged localities: ONE cluster kernel has written every force of the domain, so one pass moves home and halo rows to atom order */
__global__ void domainAllRowsKernel(float3* __restrict__ f, const float3* __restrict__ nbnxmForce, const int* __restrict__ cell, const int numAtoms)
{
    const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < numAtoms) { f[i] = nbnxmForce[cell[i]]; }
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

/* ---- one-sided transport: stores into the peer's rows, sequence flags, no transfer kernel, no event ------------------------
 *
 * The exchanges of a step are sub-megabyte, latency-bound messages between GPUs that are all linked to each other (xGMI): the
 * MI355X-native form is a store into the peer's memory, not a collective.  The sender's pack kernel writes the shifted coordinates
 * STRAIGHT into the receiver's halo rows and then publishes the exchange's sequence number in a flag in the receiver's memory; the
 * receiver's x -> xq kernel waits for the flags of its links before it reads the rows.  Forces go back the same way: the kernel that
 * moves the forces to atom order stores the halo rows into their owners' receive buffers and sets their flags, the owners' unpack-add
 * kernel waits for them.  Two more flags per link run the other way ("rows consumed"), so that the next step's stores cannot
 * overtake this step's readers.  All waits are polls of the waiter's OWN memory; all stores to a peer are release-ordered at system
 * scope behind the data (the last workgroup of the storing kernel, found with a counter, writes the flags).
 *
 * Flags of a rank (PushFlags, one 64-byte line each): [kind][link]
 *   xReady[k]     my receive link k: the peer has stored the rows of exchange seq                  (polled by my x -> xq)
 *   fConsumed[k]  my receive link k: the owner has added the forces I stored for exchange seq       (polled by my forces kernel)
 *   fReady[k]     my send link k: the peer has stored the forces on my atoms for exchange seq      (polled by my unpack-add)
 *   xConsumed[k]  my send link k: the peer's x -> xq has read the rows I stored for exchange seq    (polled by my pack kernel)
 */
constexpr int      c_pushMaxLinks   = 64;
constexpr int      c_pushFlagStride = 16; /* unsigned per flag: one 64-byte line each */
constexpr unsigned c_pushMaxSpins   = 4U << 20; /* x ~0.5 us of s_sleep: a wait of seconds is a lost peer — report it, do not hang the device */
enum PushFlagKind
{
    c_flagXReady = 0,
    c_flagFConsumed,
    c_flagFReady,
    c_flagXConsumed,
    c_numPushFlagKinds
};

/* what the kernels of one rank need to know about its links (device copy, rebuilt by halo_gpu_reinit) */
struct PushLinks
{
    int       numSend, numRecv;
    int       sendOffset[c_pushMaxLinks + 1]; /* entries [sendOffset[k], sendOffset[k + 1]) of the send map go to send link k */
    int       recvAtomOffset[c_pushMaxLinks]; /* rows [recvAtomOffset[k], + recvCount[k]) of d_x / d_f belong to receive link k */
    int       recvCount[c_pushMaxLinks];
    float3*   xDst[c_pushMaxLinks];           /* send link k: the peer's row of my entry 0 of that link, minus sendOffset[k] (so that dst = xDst[k] + i) */
    unsigned* xReadyDst[c_pushMaxLinks];      /* ... and the peer's xReady flag of this link */
    unsigned* fConsumedDst[c_pushMaxLinks];   /* send link k: the peer's fConsumed flag (I am the owner, the peer stored the forces) */
    float3*   fDst[c_pushMaxLinks];           /* receive link k: the owner's receive-buffer entry of my row recvAtomOffset[k], minus that offset */
    unsigned* fReadyDst[c_pushMaxLinks];      /* ... and the owner's fReady flag of this link */
    unsigned* xConsumedDst[c_pushMaxLinks];   /* receive link k: the sender's xConsumed flag */
};

/* Everything another device (or another XCD of this one) has stored, or is to read, is moved with SYSTEM-scope accesses — the
 * compiler gives them the cache-policy bits that bypass the caches between the wave and the memory (write-through stores, loads that
 * cannot hit a stale line).  No fence instruction anywhere: a system-scope release or acquire FENCE on this hardware writes back or
 * invalidates the whole L2 — measured with fences in these kernels: a 0.093 ms step became 0.241 ms.  Ordering comes from completion
 * instead: a wave waits until its own stores have been acknowledged (s_waitcnt) before its workgroup is counted as done, the last
 * workgroup then stores the flags; a waiter reads the rows only behind the barrier that follows the poll. */
__device__ __forceinline__ unsigned loadFlag(const unsigned* flag)
{
    return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void storeFlag(unsigned* flag, unsigned value)
{
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
/* (one 12-byte access per row, as asm: the scoped atomic builtins are 4 bytes wide — three transactions per row; measured at 168k halo
 * atoms: 0.449 ms per step against 0.379 ms over RCCL) */
typedef float nb_float3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void storeAcrossDevices(float3* dst, const float3 v)
{
    const nb_float3v d = { v.x, v.y, v.z };
    asm volatile("global_store_dwordx3 %0, %1, off sc0 sc1" ::"v"(dst), "v"(d) : "memory");
}
__device__ __forceinline__ float3 loadAcrossDevices(const float3* src)
{
    nb_float3v d;
    asm volatile("global_load_dwordx3 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(d) : "v"(src) : "memory");
    return make_float3(d.x, d.y, d.z);
}

/* the workgroup waits until the n flags have reached `want` and goes on behind the barrier.  A flag that does
 * not arrive within c_pushMaxSpins polls is reported in *error (mapped host memory; the host ends the run at its next call). */
__device__ __forceinline__ void waitForFlags(const unsigned* flags, const int n, const unsigned want, unsigned* error)
{
    /* one lane per flag: the polls of a workgroup's links run side by side (a system-scope load is a round trip to the memory) */
    if (static_cast<int>(threadIdx.x) < n)
    {
        const int k     = static_cast<int>(threadIdx.x);
        unsigned  spins = 0;
        while (static_cast<int>(loadFlag(flags + k * c_pushFlagStride) - want) < 0)
        {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > c_pushMaxSpins)
            {
                __hip_atomic_store(error, 1U + static_cast<unsigned>(k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
}

/* behind the stores of a kernel: the LAST workgroup to get here (a counter finds it) publishes `seq` in the n remote flags.  Every
 * wave waits for the acknowledgement of its own stores before its workgroup counts itself. */
__device__ __forceinline__ void publishWhenAllWorkgroupsAreDone(unsigned* const* flagDst, const int n, const unsigned seq, unsigned* doneCounter)
{
    __builtin_amdgcn_s_waitcnt(0); /* vmcnt(0): this wave's stores have been acknowledged */
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const unsigned done = __hip_atomic_fetch_add(doneCounter, 1U, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1U)
        {
            __hip_atomic_store(doneCounter, 0U, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int k = 0; k < n; k++) { storeFlag(flagDst[k], seq); }
        }
    }
}

__device__ __forceinline__ int linkOfIndex(const int* __restrict__ offsets, const int n, const int i)
{
    int k = 0;
    while (k + 1 < n && i >= offsets[k + 1]) { k++; }
    return k;
}

/* coordinates out: x[map[i]] + shift stored into the receiver's halo row; flags: wait xConsumed(seq - 1), publish xReady(seq) */
__global__ void haloPushCoordinatesKernel(const float3* __restrict__ x, const int* __restrict__ map, const int* __restrict__ shiftIndex,
                                          const float3* __restrict__ shiftVectors, const int n, const PushLinks* __restrict__ links,
                                          const unsigned* myFlags, const unsigned seq, unsigned* doneCounter, unsigned* error)
{
    if (seq > 1U) { waitForFlags(myFlags + c_flagXConsumed * c_pushMaxLinks * c_pushFlagStride, links->numSend, seq - 1U, error); }
    for (int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += static_cast<int>(gridDim.x * blockDim.x))
    {
        const int    k = linkOfIndex(links->sendOffset, links->numSend, i);
        const float3 v = x[map[i]];
        const float3 s = shiftVectors[shiftIndex[i]];
        storeAcrossDevices(links->xDst[k] + i, make_float3(v.x + s.x, v.y + s.y, v.z + s.z));
    }
    publishWhenAllWorkgroupsAreDone(links->xReadyDst, links->numSend, seq, doneCounter);
}

/* x -> xq of all grid slots of the domain behind the arrival of the halo rows; flags: wait xReady(seq), publish xConsumed(seq) */
__global__ void haloWaitXToXqKernel(float4* __restrict__ xq, float3* x, const float3* xRecv /* halo rows from numHome on, or nullptr: in x */,
                                    const int* __restrict__ atomIndex, const int numSlots, const int numHome,
                                    const PushLinks* __restrict__ links, const unsigned* myFlags, const unsigned seq, unsigned* doneCounter,
                                    unsigned* error)
{
    waitForFlags(myFlags + c_flagXReady * c_pushMaxLinks * c_pushFlagStride, links->numRecv, seq, error);
    for (int slot = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x); slot < numSlots; slot += static_cast<int>(gridDim.x * blockDim.x))
    {
        const int a = atomIndex[slot];
        /* (home rows are this rank's own; halo rows were stored by the peers) */
        if (a >= 0)
        {
            float3 v;
            if (a < numHome) { v = x[a]; }
            else if (xRecv == nullptr) { v = loadAcrossDevices(x + a); }
            else
            {
                v    = loadAcrossDevices(xRecv + (a - numHome));
                x[a] = v; /* the caller's array gets its halo rows too */
            }
            *reinterpret_cast<float3*>(&xq[slot]) = v;
        }
    }
    publishWhenAllWorkgroupsAreDone(links->xConsumedDst, links->numRecv, seq, doneCounter);
}

/* forces to atom order, the halo rows also stored into their owners' receive buffers; flags: wait fConsumed(seq - 1), publish fReady(seq) */
__global__ void haloPushForcesKernel(float3* __restrict__ f, const float3* __restrict__ nbnxmForce, const int* __restrict__ cell, const int numHome,
                                     const int numAtoms, const PushLinks* __restrict__ links, const unsigned* myFlags, const unsigned seq,
                                     unsigned* doneCounter, unsigned* error)
{
    if (seq > 1U) { waitForFlags(myFlags + c_flagFConsumed * c_pushMaxLinks * c_pushFlagStride, links->numRecv, seq - 1U, error); }
    for (int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x); i < numAtoms; i += static_cast<int>(gridDim.x * blockDim.x))
    {
        const float3 v = nbnxmForce[cell[i]];
        f[i]           = v;
        if (i >= numHome)
        {
            /* (a halo row that belongs to no link — none with a consistent plan — stays local) */
            for (int k = 0; k < links->numRecv; k++)
            {
                if (i >= links->recvAtomOffset[k] && i < links->recvAtomOffset[k] + links->recvCount[k])
                {
                    storeAcrossDevices(links->fDst[k] + i, v);
                    break;
                }
            }
        }
    }
    publishWhenAllWorkgroupsAreDone(links->fReadyDst, links->numRecv, seq, doneCounter);
}

/* f[sendMap[j]] += what the peers computed on this rank's atoms; flags: wait fReady(seq), publish fConsumed(seq) */
__global__ void haloWaitUnpackAddKernel(float* __restrict__ f, const float3* packed, const int* __restrict__ map, const int n,
                                        const PushLinks* __restrict__ links, const unsigned* myFlags, const unsigned seq, unsigned* doneCounter,
                                        unsigned* error)
{
    waitForFlags(myFlags + c_flagFReady * c_pushMaxLinks * c_pushFlagStride, links->numSend, seq, error);
    for (int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += static_cast<int>(gridDim.x * blockDim.x))
    {
        const float3 v = loadAcrossDevices(packed + i);
        float*       d = f + 3 * static_cast<size_t>(map[i]);
        atomicAdd(d + 0, v.x);
        atomicAdd(d + 1, v.y);
        atomicAdd(d + 2, v.z);
    }
    publishWhenAllWorkgroupsAreDone(links->fConsumedDst, links->numSend, seq, doneCounter);
}

/* ---- in-process peer-copy transport -------------------------------------------------------------------------------------- */

/* what a receiver needs to know about the sender of a step: where its buffers are and how its links are laid out.  Immutable;
 * a new one is made by every halo_gpu_reinit and travels with the posts of the steps that used it. */
struct PeerLayout
{
    int              device = 0;
    const float3*    d_f       = nullptr; /* forces in the rank's atom order: the halo rows are what the owners pull */
    const float3*    d_sendBuf = nullptr; /* packed coordinates: what the receivers pull */
    std::vector<int> sendPeer, sendOffset, recvPeer, recvAtomOffset, recvCount;
    /* one-sided transport: what the peers store into */
    float3*          d_x       = nullptr;
    float3*          d_recvBuf = nullptr;
    unsigned*        d_flags   = nullptr;
    long             generation = 0; /* the rank's count of halo_gpu_reinit calls */
};

enum PeerPostKind
{
    c_postXPacked = 0, /* the send buffer holds the packed coordinates of exchange seq */
    c_postXPulled,     /* this rank has copied what it needs of exchange seq out of the others' send buffers */
    c_postFReady,      /* the halo rows of d_f hold the forces of exchange seq */
    c_postFPulled,     /* this rank has copied what it needs of exchange seq out of the others' halo rows */
    c_numPostKinds
};
constexpr int c_peerRing = 4; /* posts (and their events) of the last c_peerRing exchanges stay valid; the guards keep ranks within one */

struct PeerPost
{
    long                              seq = -1;
    hipEvent_t                        event = nullptr;
    std::shared_ptr<const PeerLayout> layout;
};

/* the mailbox all ranks of one in-process "communicator" share */
struct PeerWorld
{
    std::mutex              mutex;
    std::condition_variable cv;
    int                     nranks   = 0;
    int                     attached = 0;
    std::vector<PeerPost>   posts; /* [(rank * c_numPostKinds + kind) * c_peerRing + seq % c_peerRing] */
    /* one-sided transport */
    bool                                            push = false;
    std::vector<std::shared_ptr<const PeerLayout>>  pushLayout;   /* [rank]: the layout of its last halo_gpu_reinit */
    std::vector<int>                                devices;      /* [rank] or -1 */
    bool                                            sharedDevice = false; /* two ranks of this world run on one device */
    std::vector<void*>                              retired;      /* flags and receive buffers of ranks that have gone: peers may still store into them */
    long                                            barrierCount = 0, barrierGeneration = 0;
};

std::mutex                                        g_worldsMutex;
std::map<std::string, std::shared_ptr<PeerWorld>> g_worlds;
constexpr char                                    c_peerIdMagic[8] = { 'H', 'A', 'L', 'O', 'P', 'E', 'E', 'R' };
constexpr char                                    c_pushIdMagic[8] = { 'H', 'A', 'L', 'O', 'P', 'U', 'S', 'H' };
constexpr char                                    c_ipcIdMagic[8]  = { 'H', 'A', 'L', 'O', 'I', 'P', 'C', 'P' };

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
    float3*          d_sendBuf        = nullptr; /* packed coordinates out */
    /* The forces the other ranks computed on this rank's atoms arrive in a buffer of their own (the reference's d_recvBuf_,
     * gpuhaloexchange_impl_gpu.h): the home-rows kernel of step n reads it on the LOCAL stream while step n + 1 already packs
     * coordinates on the non-local stream — with one shared buffer (round 2) that pack overwrote forces still being read. */
    float3*          d_recvBuf        = nullptr;
    int              sendAlloc = 0, shiftAlloc = 0;
    /* recv side */
    std::vector<int> recvPeer, recvAtomOffset, recvCount;
    hipEvent_t       xReady = nullptr, fReady = nullptr;
    /* recorded behind the last reader of d_recvBuf (the unpack / home-rows kernel); the next force exchange waits for it */
    hipEvent_t       recvBufConsumed         = nullptr;
    bool             recvBufConsumedRecorded = false;
    /* HALO_GPU_HOST_TIMING: host microseconds spent queueing the parts of halo_gpu_domain_force_step */
    double                                hostTimingUs[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    long                                  hostTimingSteps = 0;
    /* the local launch of halo_gpu_domain_force_step in one or two parts */
    int   localParts        = 1;
    float localPartFraction = c_defaultLocalPartFraction;
    std::chrono::steady_clock::time_point hostTimingLast;
    /* in-process peer-copy transport (world != nullptr) */
    std::shared_ptr<PeerWorld>        world;
    std::string                       worldKey;
    int                               device = 0;
    std::shared_ptr<const PeerLayout> layout;
    hipEvent_t                        peerEvents[c_numPostKinds][c_peerRing] = {};
    long                              xSeq = 0, fSeq = 0; /* exchanges done so far */
    std::vector<void*>                retiredBuffers;     /* buffers a peer may still be copying from: freed with the object */
    /* the peers of the exchanges BEFORE the last halo_gpu_reinit: a rank that is no longer a neighbour may still be copying out of
     * this rank's buffers, so the first exchange after a reinit also waits for their "pulled" posts */
    std::vector<int>                  formerSendPeer, formerRecvPeer;
    double                            peerTimeoutSeconds = 30.0;
    /* one-sided transport (world->push) */
    unsigned*  d_flags       = nullptr; /* c_numPushFlagKinds x c_pushMaxLinks flags of c_pushFlagStride unsigned: peers store, this rank polls */
    unsigned*  d_doneCounter = nullptr; /* one per kernel of the step */
    PushLinks* d_links       = nullptr;
    unsigned*  h_pushError   = nullptr; /* mapped host memory: a poll that gave up */
    unsigned*  d_pushError   = nullptr;
    long       generation    = 0;
    unsigned   pushSeq       = 0;       /* steps done */
    /* one-sided transport between PROCESSES (HALO_GPU_TRANSPORT_IPC_PUSH): the peers' buffers are opened through IPC handles that travel
     * over the caller's out-of-band channel (halo_gpu_push_export / _import).  Peers store the halo coordinates into d_xRecv — memory of
     * this library, exportable — instead of the caller's d_x; the x -> xq kernel reads them there and fills the halo rows of d_x. */
    bool                               ipcPush  = false;
    bool                               pushErrorReported = false;
    float3*                            d_xRecv  = nullptr;
    int                                xRecvAlloc = 0;
    struct IpcPeer
    {
        hipIpcMemHandle_t handle[3];     /* flags, receive buffer of the forces, receive buffer of the coordinates */
        void*             opened[3] = { nullptr, nullptr, nullptr };
        bool              valid     = false;
    };
    std::vector<IpcPeer>               ipcPeers; /* [rank] */
};

namespace
{

/* publish: exchange seq of this rank has reached `kind`; the event has been recorded on the rank's stream */
void peerPost(HaloGpu* h, PeerPostKind kind, long seq, hipStream_t s)
{
    hipEvent_t ev = h->peerEvents[kind][seq % c_peerRing];
    NBNXM_HIP_CHECK(hipEventRecord(ev, s));
    PeerWorld& w = *h->world;
    {
        std::lock_guard<std::mutex> lock(w.mutex);
        PeerPost&                   slot = w.posts[(static_cast<size_t>(h->rank) * c_numPostKinds + kind) * c_peerRing + seq % c_peerRing];
        slot.seq                         = seq;
        slot.event                       = ev;
        slot.layout                      = h->layout;
    }
    w.cv.notify_all();
}

/* host handshake (the reference's MPI_Sendrecv of event pointers, gpuhaloexchange_impl_gpu.cpp:438-470): blocks until rank `from`
 * has QUEUED exchange seq up to `kind`, then makes stream s wait for the event it recorded there.  A peer that never gets there
 * is a fatal error after peerTimeoutSeconds, not a hang. */
PeerPost peerWait(HaloGpu* h, int from, PeerPostKind kind, long seq, hipStream_t s)
{
    PeerWorld&                   w = *h->world;
    std::unique_lock<std::mutex> lock(w.mutex);
    const size_t                 idx = (static_cast<size_t>(from) * c_numPostKinds + kind) * c_peerRing + seq % c_peerRing;
    const bool ok = w.cv.wait_for(lock, std::chrono::duration<double>(h->peerTimeoutSeconds), [&] { return w.posts[idx].seq >= seq; });
    if (!ok)
    {
        char msg[256];
        static const char* names[c_numPostKinds] = { "coordinates packed", "coordinates pulled", "forces ready", "forces pulled" };
        std::snprintf(msg, sizeof(msg), "rank %d waited %.0f s for rank %d to reach '%s' of exchange %ld (in-process transport: every rank needs its own host thread)",
                      h->rank, h->peerTimeoutSeconds, from, names[kind], seq);
        fatal(__FILE__, __LINE__, "halo peer-copy handshake timed out", msg);
    }
    NBNXM_ASSERT(w.posts[idx].seq == seq, "a peer ran more than the ring of posts ahead (missing guard)");
    PeerPost post = w.posts[idx];
    lock.unlock();
    NBNXM_HIP_CHECK(hipStreamWaitEvent(s, post.event, 0));
    return post;
}

void peerCopy(void* dst, int dstDevice, const void* src, int srcDevice, size_t bytes, hipStream_t s)
{
    if (bytes == 0) { return; }
    if (dstDevice == srcDevice) { NBNXM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s)); }
    else { NBNXM_HIP_CHECK(hipMemcpyPeerAsync(dst, dstDevice, src, srcDevice, bytes, s)); }
}

/* index of the link to rank `me` in a peer's list of links (one message per peer pair and direction: halo_gpu_reinit checks it) */
int linkTo(const std::vector<int>& peers, int me)
{
    for (size_t k = 0; k < peers.size(); k++)
    {
        if (peers[k] == me) { return static_cast<int>(k); }
    }
    return -1;
}

/* coordinates, peer-copy transport: pack -> publish -> every receiver pulls its block out of the sender's buffer */
void peerExchangeCoordinates(HaloGpu* h)
{
    hipStream_t s = h->stream;
    const long  q = h->xSeq;
    /* the buffers this exchange overwrites must have been read by their last consumers: the send buffer by the ranks that pulled
     * exchange q - 1 out of it, the halo rows of d_f (rewritten later in this step, on this stream) by their owners */
    if (q > 0)
    {
        for (size_t k = 0; k < h->sendPeer.size(); k++) { peerWait(h, h->sendPeer[k], c_postXPulled, q - 1, s); }
        for (int peer : h->formerSendPeer) { peerWait(h, peer, c_postXPulled, q - 1, s); }
    }
    if (h->fSeq > 0)
    {
        for (size_t k = 0; k < h->recvPeer.size(); k++) { peerWait(h, h->recvPeer[k], c_postFPulled, h->fSeq - 1, s); }
        for (int peer : h->formerRecvPeer) { peerWait(h, peer, c_postFPulled, h->fSeq - 1, s); }
    }
    h->formerSendPeer.clear();
    h->formerRecvPeer.clear();
    if (h->numSendAtoms > 0)
    {
        hipLaunchKernelGGL(haloPackShiftedKernel, dim3((h->numSendAtoms + c_haloThreadsPerBlock - 1) / c_haloThreadsPerBlock),
                           dim3(c_haloThreadsPerBlock), 0, s, h->d_sendBuf, h->d_x, h->d_sendMap, h->d_sendShiftIndex, h->d_shiftVectors,
                           h->numSendAtoms);
        NBNXM_HIP_CHECK(hipGetLastError());
    }
    peerPost(h, c_postXPacked, q, s);
    for (size_t k = 0; k < h->recvPeer.size(); k++)
    {
        const PeerPost    post = peerWait(h, h->recvPeer[k], c_postXPacked, q, s);
        const PeerLayout& from = *post.layout;
        const int         link = linkTo(from.sendPeer, h->rank);
        NBNXM_ASSERT(link >= 0 && from.sendOffset[link + 1] - from.sendOffset[link] == h->recvCount[k],
                     "the two sides of a halo link disagree on its size");
        peerCopy(h->d_x + h->recvAtomOffset[k], h->device, from.d_sendBuf + from.sendOffset[link], from.device,
                 sizeof(float3) * static_cast<size_t>(h->recvCount[k]), s);
    }
    peerPost(h, c_postXPulled, q, s);
    h->xSeq = q + 1;
}

/* forces, peer-copy transport: publish the halo rows -> every owner pulls what was computed on its atoms into its receive buffer */
void peerExchangeForces(HaloGpu* h)
{
    hipStream_t s = h->stream;
    const long  q = h->fSeq;
    peerPost(h, c_postFReady, q, s);
    for (size_t k = 0; k < h->sendPeer.size(); k++)
    {
        const int n = h->sendOffset[k + 1] - h->sendOffset[k];
        const PeerPost    post = peerWait(h, h->sendPeer[k], c_postFReady, q, s);
        const PeerLayout& from = *post.layout;
        const int         link = linkTo(from.recvPeer, h->rank);
        NBNXM_ASSERT(link >= 0 && from.recvCount[link] == n, "the two sides of a halo link disagree on its size");
        peerCopy(h->d_recvBuf + h->sendOffset[k], h->device, from.d_f + from.recvAtomOffset[link], from.device, sizeof(float3) * static_cast<size_t>(n), s);
    }
    peerPost(h, c_postFPulled, q, s);
    h->fSeq = q + 1;
}

/* all ranks of the world (one-sided transport: at the end of a reinit, and between the phases of a step when ranks share a device) */
void worldBarrier(HaloGpu* h)
{
    PeerWorld&                   w = *h->world;
    std::unique_lock<std::mutex> lock(w.mutex);
    const long                   gen = w.barrierGeneration;
    if (++w.barrierCount == w.nranks)
    {
        w.barrierCount = 0;
        w.barrierGeneration++;
        w.cv.notify_all();
        return;
    }
    const bool ok = w.cv.wait_for(lock, std::chrono::duration<double>(h->peerTimeoutSeconds), [&] { return w.barrierGeneration != gen; });
    if (!ok) { fatal(__FILE__, __LINE__, "halo one-sided transport", "a rank did not reach the barrier (every rank needs its own host thread and the same sequence of calls)"); }
}

/* true: a kernel of an earlier step has given up waiting for a peer.  Ranks that are threads of one process (tests, rehearsals) stop there;
 * between processes (a node run) it is not fatal: the step is skipped, halo_gpu_push_status() says why, and the caller goes on with another
 * transport (bench.py's decomposition leg) — an abort here would take the rest of the job's output with it. */
bool checkPushError(HaloGpu* h)
{
    if (h->h_pushError == nullptr || *h->h_pushError == 0U) { return false; }
    char msg[160];
    std::snprintf(msg, sizeof(msg), "rank %d: a kernel gave up waiting for the flag of link %u (a peer never stored its side of an exchange)", h->rank,
                  *h->h_pushError - 1U);
    if (!h->ipcPush) { fatal(__FILE__, __LINE__, "halo one-sided transport", msg); }
    if (!h->pushErrorReported)
    {
        std::fprintf(stderr, "halo one-sided transport: %s; the steps of this object are skipped from here on\n", msg);
        h->pushErrorReported = true;
    }
    return true;
}

/* halo_gpu_reinit, one-sided transport: publish this rank's buffers, wait for the neighbours', derive every link's remote addresses */
void pushReinit(HaloGpu* h)
{
    PeerWorld& w = *h->world;
    NBNXM_ASSERT(static_cast<int>(h->sendPeer.size()) <= c_pushMaxLinks && static_cast<int>(h->recvPeer.size()) <= c_pushMaxLinks,
                 "more links than the one-sided transport provides flags for");
    h->generation++;
    auto layout            = std::make_shared<PeerLayout>();
    layout->device         = h->device;
    layout->d_f            = h->d_f;
    layout->d_sendBuf      = h->d_sendBuf;
    layout->sendPeer       = h->sendPeer;
    layout->sendOffset     = h->sendOffset;
    layout->recvPeer       = h->recvPeer;
    layout->recvAtomOffset = h->recvAtomOffset;
    layout->recvCount      = h->recvCount;
    layout->d_x            = h->d_x;
    layout->d_recvBuf      = h->d_recvBuf;
    layout->d_flags        = h->d_flags;
    layout->generation     = h->generation;
    h->layout              = layout;
    {
        std::lock_guard<std::mutex> lock(w.mutex);
        w.pushLayout[h->rank] = layout;
    }
    w.cv.notify_all();
    auto layoutOf = [&](int peer) {
        std::unique_lock<std::mutex> lock(w.mutex);
        const bool ok = w.cv.wait_for(lock, std::chrono::duration<double>(h->peerTimeoutSeconds),
                                      [&] { return w.pushLayout[peer] && w.pushLayout[peer]->generation >= h->generation; });
        if (!ok) { fatal(__FILE__, __LINE__, "halo one-sided transport", "a neighbour did not call halo_gpu_reinit (every rank re-registers at every search step)"); }
        return w.pushLayout[peer];
    };
    PushLinks L;
    std::memset(&L, 0, sizeof(L));
    L.numSend = static_cast<int>(h->sendPeer.size());
    L.numRecv = static_cast<int>(h->recvPeer.size());
    auto flagOf = [](unsigned* flags, int kind, int link) { return flags + (static_cast<size_t>(kind) * c_pushMaxLinks + link) * c_pushFlagStride; };
    for (int k = 0; k <= L.numSend; k++) { L.sendOffset[k] = h->sendOffset[k]; }
    for (int k = 0; k < L.numSend; k++)
    {
        const std::shared_ptr<const PeerLayout> peer = layoutOf(h->sendPeer[k]);
        const int                               link = linkTo(peer->recvPeer, h->rank);
        NBNXM_ASSERT(link >= 0 && peer->recvCount[link] == h->sendOffset[k + 1] - h->sendOffset[k], "the two sides of a halo link disagree on its size");
        if (peer->device != h->device)
        {
            int can = 0;
            NBNXM_HIP_CHECK(hipDeviceCanAccessPeer(&can, h->device, peer->device));
            NBNXM_ASSERT(can != 0, "the one-sided transport needs peer access between the devices of the ranks");
            const hipError_t e = hipDeviceEnablePeerAccess(peer->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { NBNXM_HIP_CHECK(e); }
            (void)hipGetLastError();
        }
        L.xDst[k]         = peer->d_x + peer->recvAtomOffset[link] - h->sendOffset[k];
        L.xReadyDst[k]    = flagOf(peer->d_flags, c_flagXReady, link);
        L.fConsumedDst[k] = flagOf(peer->d_flags, c_flagFConsumed, link);
    }
    for (int k = 0; k < L.numRecv; k++)
    {
        const std::shared_ptr<const PeerLayout> peer = layoutOf(h->recvPeer[k]);
        const int                               link = linkTo(peer->sendPeer, h->rank);
        NBNXM_ASSERT(link >= 0 && peer->sendOffset[link + 1] - peer->sendOffset[link] == h->recvCount[k], "the two sides of a halo link disagree on its size");
        if (peer->device != h->device)
        {
            const hipError_t e = hipDeviceEnablePeerAccess(peer->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { NBNXM_HIP_CHECK(e); }
            (void)hipGetLastError();
        }
        L.recvAtomOffset[k] = h->recvAtomOffset[k];
        L.recvCount[k]      = h->recvCount[k];
        L.fDst[k]           = peer->d_recvBuf + peer->sendOffset[link] - h->recvAtomOffset[k];
        L.fReadyDst[k]      = flagOf(peer->d_flags, c_flagFReady, link);
        L.xConsumedDst[k]   = flagOf(peer->d_flags, c_flagXConsumed, link);
    }
    NBNXM_HIP_CHECK(hipMemcpy(h->d_links, &L, sizeof(L), hipMemcpyHostToDevice));
    /* every rank has published and read: from here on the old buffers of any rank are out of use, and the device set is known */
    worldBarrier(h);
}

} // namespace

extern "C"
{

const char* halo_gpu_last_error(void)
{
    return g_haloError.c_str();
}

int halo_gpu_get_unique_id_ex(void* uniqueId, int transport)
{
    if (transport == HALO_GPU_TRANSPORT_RCCL) { return halo_gpu_get_unique_id(uniqueId); }
    if (transport == HALO_GPU_TRANSPORT_IPC_PUSH)
    {
        /* nothing to agree on: every rank builds the same id */
        std::memset(uniqueId, 0, HALO_GPU_UNIQUE_ID_BYTES);
        std::memcpy(uniqueId, c_ipcIdMagic, sizeof(c_ipcIdMagic));
        return 0;
    }
    if (transport != HALO_GPU_TRANSPORT_PEER_COPY && transport != HALO_GPU_TRANSPORT_PEER_PUSH)
    {
        g_haloError = "unknown halo transport";
        return 3;
    }
    /* an id that no RCCL id can be taken for: magic, then a process-wide counter */
    static std::mutex counterMutex;
    static long long  counter = 0;
    std::memset(uniqueId, 0, HALO_GPU_UNIQUE_ID_BYTES);
    std::memcpy(uniqueId, transport == HALO_GPU_TRANSPORT_PEER_PUSH ? c_pushIdMagic : c_peerIdMagic, sizeof(c_peerIdMagic));
    std::lock_guard<std::mutex> lock(counterMutex);
    const long long             value = ++counter;
    std::memcpy(static_cast<char*>(uniqueId) + sizeof(c_peerIdMagic), &value, sizeof(value));
    return 0;
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

/* Buffers that other PROCESSES open through hipIpc handles get an allocation of their own: the runtime carves requests below 2 MB
 * out of shared 2 MB blocks, a handle then stands for the whole block, and a process that opens two handles into one block gets
 * "invalid device pointer" for the second (seen with four ranks on one device, whose small buffers shared a block).  A request of a
 * multiple of 2 MB is never carved. */
static size_t ipcExportableBytes(size_t bytes)
{
    constexpr size_t c_block = size_t(2) << 20;
    return ((bytes + c_block - 1) / c_block) * c_block;
}

/* Memory that OTHER devices store into while this device's kernels poll or read it (flags, receive buffers of the one-sided transports):
 * fine-grained device memory, so that this device's L2 does not keep lines a peer has written behind its back — on one device
 * everything goes through the same L2 and any memory would do; across xGMI a remote store lands in HBM, not in the owner's L2.
 * (hipMalloc's coarse-grained memory is only coherent between devices at kernel boundaries.)  Falls back to hipMalloc where the
 * runtime has no fine-grained pool. */
static void mallocForPeerStores(void** p, size_t bytes)
{
    const size_t n = ipcExportableBytes(bytes);
    if (hipExtMallocWithFlags(p, n, hipDeviceMallocFinegrained) != hipSuccess)
    {
        (void)hipGetLastError();
        NBNXM_HIP_CHECK(hipMalloc(p, n));
    }
}

HaloGpu* halo_gpu_create(const void* uniqueId, int rank, int nranks, void* stream)
{
    if (rank < 0 || rank >= nranks)
    {
        g_haloError = "rank outside [0, nranks)";
        return nullptr;
    }
    const bool pushTransport     = (std::memcmp(uniqueId, c_pushIdMagic, sizeof(c_pushIdMagic)) == 0);
    const bool peerCopyTransport = pushTransport || (std::memcmp(uniqueId, c_peerIdMagic, sizeof(c_peerIdMagic)) == 0);
    auto*      h                 = new HaloGpu;
    h->rank   = rank;
    h->nranks = nranks;
    h->stream = static_cast<hipStream_t>(stream);
    NBNXM_HIP_CHECK(hipGetDevice(&h->device));
    auto allocatePushState = [&]() {
        const size_t numFlags = static_cast<size_t>(c_numPushFlagKinds) * c_pushMaxLinks * c_pushFlagStride;
        mallocForPeerStores(reinterpret_cast<void**>(&h->d_flags), sizeof(unsigned) * numFlags);
        NBNXM_HIP_CHECK(hipMemset(h->d_flags, 0, sizeof(unsigned) * numFlags));
        NBNXM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_doneCounter), sizeof(unsigned) * 4 * c_pushFlagStride));
        NBNXM_HIP_CHECK(hipMemset(h->d_doneCounter, 0, sizeof(unsigned) * 4 * c_pushFlagStride));
        NBNXM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_links), sizeof(PushLinks)));
        NBNXM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->h_pushError), sizeof(unsigned), hipHostMallocMapped));
        *h->h_pushError = 0U;
        NBNXM_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&h->d_pushError), h->h_pushError, 0));
    };
    if (std::memcmp(uniqueId, c_ipcIdMagic, sizeof(c_ipcIdMagic)) == 0)
    {
        /* one-sided transport between processes: nothing collective here; the buffers are exchanged at halo_gpu_reinit time
         * (halo_gpu_push_export / halo_gpu_push_import) */
        h->ipcPush = true;
        h->ipcPeers.resize(nranks);
        allocatePushState();
        if (const char* env = std::getenv("HALO_GPU_PEER_TIMEOUT")) { h->peerTimeoutSeconds = std::max(1.0, std::atof(env)); }
    }
    else if (peerCopyTransport)
    {
        /* in-process peer copies: attach to the mailbox of this id (the first rank makes it); nothing collective happens here */
        h->worldKey.assign(static_cast<const char*>(uniqueId), HALO_GPU_UNIQUE_ID_BYTES);
        std::lock_guard<std::mutex> lock(g_worldsMutex);
        std::shared_ptr<PeerWorld>& w = g_worlds[h->worldKey];
        if (!w)
        {
            w         = std::make_shared<PeerWorld>();
            w->nranks = nranks;
            w->posts.resize(static_cast<size_t>(nranks) * c_numPostKinds * c_peerRing);
            w->push = pushTransport;
            w->pushLayout.resize(nranks);
            w->devices.assign(nranks, -1);
        }
        if (w->nranks != nranks)
        {
            g_haloError = "the ranks of one in-process halo communicator disagree on their number";
            delete h;
            return nullptr;
        }
        w->attached++;
        for (int r = 0; r < nranks; r++)
        {
            if (r != rank && w->devices[r] == h->device) { w->sharedDevice = true; }
        }
        w->devices[rank] = h->device;
        h->world = w;
        if (pushTransport) { allocatePushState(); }
        for (int k = 0; k < c_numPostKinds; k++)
        {
            for (int i = 0; i < c_peerRing; i++) { NBNXM_HIP_CHECK(hipEventCreateWithFlags(&h->peerEvents[k][i], hipEventDisableTiming)); }
        }
        if (const char* env = std::getenv("HALO_GPU_PEER_TIMEOUT")) { h->peerTimeoutSeconds = std::max(1.0, std::atof(env)); }
    }
    else
    {
        Rccl* r = rccl();
        if (r == nullptr)
        {
            delete h;
            return nullptr;
        }
        ncclUniqueId id;
        std::memcpy(&id, uniqueId, sizeof(id));
        const ncclResult_t res = r->CommInitRank(&h->comm, nranks, id, rank);
        if (res != ncclSuccess)
        {
            g_haloError = std::string("ncclCommInitRank: ") + r->GetErrorString(res);
            delete h;
            return nullptr;
        }
    }
    NBNXM_HIP_CHECK(hipEventCreateWithFlags(&h->xReady, hipEventDisableTiming));
    NBNXM_HIP_CHECK(hipEventCreateWithFlags(&h->fReady, hipEventDisableTiming));
    NBNXM_HIP_CHECK(hipEventCreateWithFlags(&h->recvBufConsumed, hipEventDisableTiming));
    /* the local launch of a domain step in two parts: on by default where the exchanges leave the device (see halo_gpu_domain_force_step) */
    h->localParts = (nranks > 1) ? 2 : 1;
    if (const char* env = diagnosticsEnv("HALO_GPU_LOCAL_PARTS")) { h->localParts = (std::atoi(env) == 2) ? 2 : 1; }
    if (const char* env = diagnosticsEnv("HALO_GPU_LOCAL_PART_FRACTION"))
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
    if (h->world)
    {
        /* peers may still have copies out of this rank's buffers in flight: the whole device goes idle first (free is rare) */
        (void)hipDeviceSynchronize();
        std::lock_guard<std::mutex> lock(g_worldsMutex);
        if (h->world->push)
        {
            /* the peers' last kernels may still store flags (and forces) into this rank's memory: it lives until the last rank has gone */
            h->world->retired.push_back(h->d_flags);
            h->world->retired.push_back(h->d_recvBuf);
            h->d_flags   = nullptr;
            h->d_recvBuf = nullptr;
        }
        if (--h->world->attached == 0)
        {
            for (void* b : h->world->retired) { (void)hipFree(b); }
            g_worlds.erase(h->worldKey);
        }
        for (int k = 0; k < c_numPostKinds; k++)
        {
            for (int i = 0; i < c_peerRing; i++) { (void)hipEventDestroy(h->peerEvents[k][i]); }
        }
    }
    for (HaloGpu::IpcPeer& p : h->ipcPeers)
    {
        for (void* o : p.opened)
        {
            if (o != nullptr) { (void)hipIpcCloseMemHandle(o); }
        }
    }
    (void)hipFree(h->d_xRecv);
    (void)hipFree(h->d_flags);
    (void)hipFree(h->d_doneCounter);
    (void)hipFree(h->d_links);
    if (h->h_pushError != nullptr) { (void)hipHostFree(h->h_pushError); }
    for (void* b : h->retiredBuffers) { (void)hipFree(b); }
    (void)hipFree(h->d_sendMap);
    (void)hipFree(h->d_sendShiftIndex);
    (void)hipFree(h->d_shiftVectors);
    (void)hipFree(h->d_sendBuf);
    (void)hipFree(h->d_recvBuf);
    if (h->xReady) { (void)hipEventDestroy(h->xReady); }
    if (h->fReady) { (void)hipEventDestroy(h->fReady); }
    if (h->recvBufConsumed) { (void)hipEventDestroy(h->recvBufConsumed); }
    delete h;
}

void halo_gpu_reinit(HaloGpu* h, void* d_x, void* d_f, int numHome, int numSend, const int* sendPeer, const int* sendOffset,
                     const int* sendMap, const int* sendShiftIndex, int numShiftVectors, const float* shiftVectors, int numRecv,
                     const int* recvPeer, const int* recvAtomOffset, const int* recvCount)
{
    h->d_x     = static_cast<float3*>(d_x);
    h->d_f     = static_cast<float3*>(d_f);
    h->numHome = numHome;
    if (h->world)
    {
        /* (every rank posts "pulled" for every exchange, neighbour or not, so waiting for a former peer cannot block for good) */
        h->formerSendPeer.insert(h->formerSendPeer.end(), h->sendPeer.begin(), h->sendPeer.end());
        h->formerRecvPeer.insert(h->formerRecvPeer.end(), h->recvPeer.begin(), h->recvPeer.end());
    }
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
    /* one message per peer and direction (both transports match the links of a pair of ranks by peer) */
    for (int k = 0; k < numSend; k++)
    {
        for (int l = 0; l < k; l++) { NBNXM_ASSERT(sendPeer[l] != sendPeer[k], "two send links to the same rank: merge them into one message"); }
    }
    for (int k = 0; k < numRecv; k++)
    {
        for (int l = 0; l < k; l++) { NBNXM_ASSERT(recvPeer[l] != recvPeer[k], "two receive links from the same rank: merge them into one message"); }
    }
    /* the device must be done with the old maps and buffers before they are replaced (search steps are rare); the home-rows kernel
     * of the domain step reads the send map and the receive buffer on the local stream */
    NBNXM_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (h->recvBufConsumedRecorded) { NBNXM_HIP_CHECK(hipEventSynchronize(h->recvBufConsumed)); }
    if (h->numSendAtoms > h->sendAlloc)
    {
        (void)hipFree(h->d_sendMap);
        (void)hipFree(h->d_sendShiftIndex);
        if (((h->world && h->world->push) || h->ipcPush) && h->d_recvBuf != nullptr) { h->retiredBuffers.push_back(h->d_recvBuf); }
        else { (void)hipFree(h->d_recvBuf); }
        /* peer-copy transport: another rank may still be copying out of the old send buffer on ITS stream: it is kept until the
         * object goes (reallocation happens a few times per run at most) */
        if (h->world && h->d_sendBuf != nullptr) { h->retiredBuffers.push_back(h->d_sendBuf); }
        else { (void)hipFree(h->d_sendBuf); }
        h->sendAlloc = static_cast<int>(h->numSendAtoms * 1.2) + 1024;
        NBNXM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_sendMap), sizeof(int) * h->sendAlloc));
        NBNXM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_sendShiftIndex), sizeof(int) * h->sendAlloc));
        NBNXM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_sendBuf), sizeof(float3) * h->sendAlloc));
        if (h->ipcPush || (h->world && h->world->push)) { mallocForPeerStores(reinterpret_cast<void**>(&h->d_recvBuf), sizeof(float3) * h->sendAlloc); }
        else { NBNXM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_recvBuf), sizeof(float3) * h->sendAlloc)); }
    }
    if (h->ipcPush)
    {
        (void)checkPushError(h);
        int numHalo = 0;
        for (int k = 0; k < numRecv; k++) { numHalo = std::max(numHalo, recvAtomOffset[k] + recvCount[k] - numHome); }
        if (numHalo > h->xRecvAlloc || h->d_xRecv == nullptr)
        {
            if (h->d_xRecv != nullptr) { h->retiredBuffers.push_back(h->d_xRecv); } /* a peer may still have it open */
            h->xRecvAlloc = static_cast<int>(numHalo * 1.2) + 1024;
            mallocForPeerStores(reinterpret_cast<void**>(&h->d_xRecv), sizeof(float3) * h->xRecvAlloc);
        }
        if (h->d_recvBuf == nullptr) { mallocForPeerStores(reinterpret_cast<void**>(&h->d_recvBuf), sizeof(float3) * 1024); } /* (a rank that sends nothing still exports a buffer) */
        h->generation++;
        /* (the links' remote addresses follow with halo_gpu_push_import, once the ranks have exchanged what halo_gpu_push_export gives) */
    }
    else if (h->world && h->world->push)
    {
        (void)checkPushError(h);
        pushReinit(h);
    }
    else if (h->world)
    {
        auto layout            = std::make_shared<PeerLayout>();
        layout->device         = h->device;
        layout->d_f            = h->d_f;
        layout->d_sendBuf      = h->d_sendBuf;
        layout->sendPeer       = h->sendPeer;
        layout->sendOffset     = h->sendOffset;
        layout->recvPeer       = h->recvPeer;
        layout->recvAtomOffset = h->recvAtomOffset;
        layout->recvCount      = h->recvCount;
        h->layout              = layout;
    }
    if (numShiftVectors > h->shiftAlloc)
    {
        (void)hipFree(h->d_shiftVectors);
        h->shiftAlloc = numShiftVectors + 32;
        NBNXM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&h->d_shiftVectors), sizeof(float3) * h->shiftAlloc));
    }
    /* search steps are rare: plain synchronous copies */
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

/* what a rank tells the others about itself, one-sided transport between processes: fixed-size record, plain bytes */
struct PushExportRecord
{
    char              magic[8];
    int               rank, numSend, numRecv, numHome;
    long long         generation;
    hipIpcMemHandle_t handle[3]; /* flags, receive buffer of the forces, receive buffer of the coordinates */
    int               sendPeer[c_pushMaxLinks], sendOffset[c_pushMaxLinks + 1];
    int               recvPeer[c_pushMaxLinks], recvAtomOffset[c_pushMaxLinks], recvCount[c_pushMaxLinks];
};

int halo_gpu_push_export_bytes(void)
{
    return static_cast<int>(sizeof(PushExportRecord));
}

int halo_gpu_push_export(HaloGpu* h, void* record)
{
    if (!h->ipcPush)
    {
        g_haloError = "halo_gpu_push_export: the object was not created with the HALO_GPU_TRANSPORT_IPC_PUSH id";
        return 1;
    }
    NBNXM_ASSERT(static_cast<int>(h->sendPeer.size()) <= c_pushMaxLinks && static_cast<int>(h->recvPeer.size()) <= c_pushMaxLinks,
                 "more links than the one-sided transport provides flags for");
    PushExportRecord r;
    std::memset(&r, 0, sizeof(r));
    std::memcpy(r.magic, c_ipcIdMagic, sizeof(r.magic));
    r.rank       = h->rank;
    r.numSend    = static_cast<int>(h->sendPeer.size());
    r.numRecv    = static_cast<int>(h->recvPeer.size());
    r.numHome    = h->numHome;
    r.generation = h->generation;
    NBNXM_HIP_CHECK(hipIpcGetMemHandle(&r.handle[0], h->d_flags));
    NBNXM_HIP_CHECK(hipIpcGetMemHandle(&r.handle[1], h->d_recvBuf));
    NBNXM_HIP_CHECK(hipIpcGetMemHandle(&r.handle[2], h->d_xRecv));
    for (int k = 0; k < r.numSend; k++) { r.sendPeer[k] = h->sendPeer[k]; }
    for (int k = 0; k <= r.numSend; k++) { r.sendOffset[k] = h->sendOffset[k]; }
    for (int k = 0; k < r.numRecv; k++)
    {
        r.recvPeer[k]       = h->recvPeer[k];
        r.recvAtomOffset[k] = h->recvAtomOffset[k];
        r.recvCount[k]      = h->recvCount[k];
    }
    std::memcpy(record, &r, sizeof(r));
    return 0;
}

int halo_gpu_push_import(HaloGpu* h, const void* records, int numRanks)
{
    if (!h->ipcPush || numRanks != h->nranks)
    {
        g_haloError = "halo_gpu_push_import: not a HALO_GPU_TRANSPORT_IPC_PUSH object, or a record count that differs from the number of ranks";
        return 1;
    }
    const PushExportRecord* all = static_cast<const PushExportRecord*>(records);
    for (int r = 0; r < numRanks; r++)
    {
        if (std::memcmp(all[r].magic, c_ipcIdMagic, sizeof(c_ipcIdMagic)) != 0 || all[r].rank != r || all[r].generation != h->generation)
        {
            g_haloError = "halo_gpu_push_import: record " + std::to_string(r) + " is not that rank's export of this search step";
            return 2;
        }
    }
    std::string openFailure;
    /* pointers into a peer's three buffers; this rank's own are used directly (a handle cannot be opened by the process that made it) */
    auto opened = [&](int peer, int which) -> char* {
        if (peer == h->rank) { return reinterpret_cast<char*>(which == 0 ? static_cast<void*>(h->d_flags) : which == 1 ? static_cast<void*>(h->d_recvBuf) : static_cast<void*>(h->d_xRecv)); }
        HaloGpu::IpcPeer& p = h->ipcPeers[peer];
        if (!p.valid || std::memcmp(&p.handle[which], &all[peer].handle[which], sizeof(hipIpcMemHandle_t)) != 0 || p.opened[which] == nullptr)
        {
            if (p.opened[which] != nullptr) { (void)hipIpcCloseMemHandle(p.opened[which]); }
            p.handle[which] = all[peer].handle[which];
            /* (four ranks on one device: the open now and then fails with "invalid device pointer" while the exporting process is still busy
             * with its own opens, and succeeds a moment later: a few attempts before giving up) */
            hipError_t err = hipSuccess;
            for (int attempt = 0; attempt < 40; attempt++)
            {
                err = hipIpcOpenMemHandle(&p.opened[which], p.handle[which], hipIpcMemLazyEnablePeerAccess);
                if (err == hipSuccess) { break; }
                (void)hipGetLastError();
                p.opened[which] = nullptr;
                std::this_thread::sleep_for(std::chrono::milliseconds(25));
            }
            if (err != hipSuccess)
            {
                /* not fatal: the caller falls back to another transport (bench.py's decomposition leg does) */
                openFailure = std::string("halo_gpu_push_import: hipIpcOpenMemHandle of rank ") + std::to_string(peer) + "'s buffer " + std::to_string(which)
                              + ": " + hipGetErrorString(err);
                return nullptr;
            }
        }
        return static_cast<char*>(p.opened[which]);
    };
    /* open everything first: nothing of the links is touched when a peer's buffer cannot be opened */
    for (int k = 0; k < static_cast<int>(h->sendPeer.size()); k++)
    {
        for (int which : { 0, 2 }) { if (opened(h->sendPeer[k], which) == nullptr) { g_haloError = openFailure; return 3; } }
    }
    for (int k = 0; k < static_cast<int>(h->recvPeer.size()); k++)
    {
        for (int which : { 0, 1 }) { if (opened(h->recvPeer[k], which) == nullptr) { g_haloError = openFailure; return 3; } }
    }
    auto linkIn = [](const int* peers, int n, int me) {
        for (int k = 0; k < n; k++)
        {
            if (peers[k] == me) { return k; }
        }
        return -1;
    };
    auto flagOf = [](char* flags, int kind, int link) {
        return reinterpret_cast<unsigned*>(flags) + (static_cast<size_t>(kind) * c_pushMaxLinks + link) * c_pushFlagStride;
    };
    PushLinks L;
    std::memset(&L, 0, sizeof(L));
    L.numSend = static_cast<int>(h->sendPeer.size());
    L.numRecv = static_cast<int>(h->recvPeer.size());
    for (int k = 0; k <= L.numSend; k++) { L.sendOffset[k] = h->sendOffset[k]; }
    for (int k = 0; k < L.numSend; k++)
    {
        const int               peer = h->sendPeer[k];
        const PushExportRecord& p    = all[peer];
        const int               link = linkIn(p.recvPeer, p.numRecv, h->rank);
        NBNXM_ASSERT(link >= 0 && p.recvCount[link] == h->sendOffset[k + 1] - h->sendOffset[k], "the two sides of a halo link disagree on its size");
        /* my entries of this link land in the peer's coordinate buffer at (row - numHome of the peer) */
        L.xDst[k]         = reinterpret_cast<float3*>(opened(peer, 2)) + (p.recvAtomOffset[link] - p.numHome) - h->sendOffset[k];
        L.xReadyDst[k]    = flagOf(opened(peer, 0), c_flagXReady, link);
        L.fConsumedDst[k] = flagOf(opened(peer, 0), c_flagFConsumed, link);
    }
    for (int k = 0; k < L.numRecv; k++)
    {
        const int               peer = h->recvPeer[k];
        const PushExportRecord& p    = all[peer];
        const int               link = linkIn(p.sendPeer, p.numSend, h->rank);
        NBNXM_ASSERT(link >= 0 && p.sendOffset[link + 1] - p.sendOffset[link] == h->recvCount[k], "the two sides of a halo link disagree on its size");
        L.recvAtomOffset[k] = h->recvAtomOffset[k];
        L.recvCount[k]      = h->recvCount[k];
        L.fDst[k]           = reinterpret_cast<float3*>(opened(peer, 1)) + p.sendOffset[link] - h->recvAtomOffset[k];
        L.fReadyDst[k]      = flagOf(opened(peer, 0), c_flagFReady, link);
        L.xConsumedDst[k]   = flagOf(opened(peer, 0), c_flagXConsumed, link);
    }
    for (HaloGpu::IpcPeer& p : h->ipcPeers) { p.valid = true; }
    NBNXM_HIP_CHECK(hipMemcpy(h->d_links, &L, sizeof(L), hipMemcpyHostToDevice));
    return 0;
}

/* 0: fine; otherwise 1 + the link whose flag a kernel of the one-sided transport gave up waiting for (the results of that step are void) */
int halo_gpu_push_status(const HaloGpu* h)
{
    return (h->h_pushError != nullptr) ? static_cast<int>(*h->h_pushError) : 0;
}

void halo_gpu_communicate_coordinates(HaloGpu* h, void* dependencyEvent)
{
    hipStream_t s = h->stream;
    if (dependencyEvent != nullptr) { NBNXM_HIP_CHECK(hipStreamWaitEvent(s, static_cast<hipEvent_t>(dependencyEvent), 0)); }
    NBNXM_ASSERT(!(h->world && h->world->push) && !h->ipcPush, "the one-sided transport has no separate exchange calls: its stores and waits live in the kernels of "
                                               "halo_gpu_domain_force_step (merged localities)");
    if (h->world)
    {
        peerExchangeCoordinates(h);
        NBNXM_HIP_CHECK(hipEventRecord(h->xReady, s));
        return;
    }
    Rccl* r = rccl();
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
    hipStream_t s = h->stream;
    /* the receive buffer is overwritten: its reader of the previous step (possibly on another stream) must be done */
    if (h->recvBufConsumedRecorded) { NBNXM_HIP_CHECK(hipStreamWaitEvent(s, h->recvBufConsumed, 0)); }
    NBNXM_ASSERT(!(h->world && h->world->push) && !h->ipcPush, "the one-sided transport has no separate exchange calls (halo_gpu_domain_force_step)");
    if (h->world)
    {
        peerExchangeForces(h);
        return;
    }
    Rccl* r = rccl();
    /* the reverse of the coordinate exchange: the halo rows of d_f go to their owners as they are (contiguous, no pack),
     * what the others computed on this rank's atoms arrives in the receive buffer, in the order of the send map */
    HALO_RCCL_CHECK(r->GroupStart());
    for (size_t k = 0; k < h->sendPeer.size(); k++)
    {
        const int n = h->sendOffset[k + 1] - h->sendOffset[k];
        if (n > 0) { HALO_RCCL_CHECK(r->Recv(h->d_recvBuf + h->sendOffset[k], static_cast<size_t>(3) * n, ncclFloat, h->sendPeer[k], h->comm, s)); }
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
                               h->d_recvBuf, h->d_sendMap, h->numSendAtoms);
        }
        else
        {
            hipLaunchKernelGGL(haloUnpackForcesKernel<false>, grid, dim3(c_haloThreadsPerBlock), 0, s, reinterpret_cast<float*>(h->d_f),
                               h->d_recvBuf, h->d_sendMap, h->numSendAtoms);
        }
        NBNXM_HIP_CHECK(hipGetLastError());
    }
    h->recvBufConsumedRecorded = false; /* the reader ran on this very stream: stream order is enough */
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
    NBNXM_ASSERT(sNonLocal == h->stream || nbnxm_gpu_get_merged_localities(nb),
                 "the halo object must have been created on the non-local stream of the non-bonded object");
    /* the halo coordinates leave first: the pack and the RCCL kernel find the device idle at the start of a step; queued behind the
     * local kernel — which fills every wave slot — the RCCL kernel would wait for slots, and the peers with it (on one GPU the order
     * makes no difference: 0.141 ms either way) */
    /* diagnostics (HALO_GPU_HOST_TIMING=1): host time spent queueing each part, printed by halo_gpu_free */
    static const bool s_hostTiming = (diagnosticsEnv("HALO_GPU_HOST_TIMING") != nullptr);
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
    if ((h->world && h->world->push) || h->ipcPush)
    {
        /* One-sided transport (see the kernels above): five launches on ONE stream, no transfer kernel, no group call, no event —
         *     store x into the peers' rows | wait + x to xq | merged cluster kernel | forces to atom order + store the halo rows into their
         *     owners' buffers | wait + add what the peers computed on my atoms.
         * Ranks that SHARE a device (tests on a one-GPU box) are a special case: their streams share the device's few hardware queues, and
         * a kernel that waits for a flag in front of the kernel that sets it in the same queue would wait for ever; between the phases
         * all ranks therefore meet at a host barrier, which makes every waiter's producer be queued first.  Ranks on devices of their own
         * need none. */
        NBNXM_ASSERT(nbnxm_gpu_get_merged_localities(nb) && h->stream == sLocal,
                     "the one-sided transport runs the merged-localities schedule on the LOCAL stream of the non-bonded object");
        if (checkPushError(h)) { return; }
        const bool     shared = h->world && h->world->sharedDevice && h->nranks > 1; /* (ranks in processes of their own: their queues are their own) */
        const unsigned seq    = ++h->pushSeq;
        const unsigned* flags = h->d_flags;
        auto counter = [&](int kernel) { return h->d_doneCounter + kernel * c_pushFlagStride; };
        /* few, large workgroups with a grid-stride loop: every workgroup ends with one add to the kernel's counter, and adds to one address
         * take ~10 ns each (555 workgroups of 256 threads: 5 us of a 10 us kernel) */
        static constexpr int c_pushThreads = 1024, c_pushMaxBlocks = 64;
        auto blocks = [](int n) { return dim3(static_cast<unsigned>(std::min<int>(c_pushMaxBlocks, std::max<int>(1, (n + c_pushThreads - 1) / c_pushThreads)))); };
        nbnxm_gpu_set_local_launch_parts(nb, 1, h->localPartFraction);
        if (coordinatesReadyEvent != nullptr) { NBNXM_HIP_CHECK(hipStreamWaitEvent(sLocal, static_cast<hipEvent_t>(coordinatesReadyEvent), 0)); }
        hipLaunchKernelGGL(haloPushCoordinatesKernel, blocks(h->numSendAtoms), dim3(c_pushThreads), 0, sLocal, h->d_x, h->d_sendMap,
                           h->d_sendShiftIndex, h->d_shiftVectors, h->numSendAtoms, h->d_links, flags, seq, counter(0), h->d_pushError);
        NBNXM_HIP_CHECK(hipGetLastError());
        tick(0);
        if (shared) { worldBarrier(h); }
        nbnxm_gpu_clear_outputs(nb, stepWork->computeVirial);
        tick(1);
        NBNXM_ASSERT(numSlots <= nb->atomIndicesSize, "grid slots outside the uploaded atomIndices (call nbnxm_gpu_init_x_to_nbat_x after each search)");
        hipLaunchKernelGGL(haloWaitXToXqKernel, blocks(numSlots), dim3(c_pushThreads), 0, sLocal, nb->atdat->xq, h->d_x, h->ipcPush ? h->d_xRecv : nullptr,
                           nb->atomIndices, numSlots, h->numHome, h->d_links, flags, seq, counter(1), h->d_pushError);
        NBNXM_HIP_CHECK(hipGetLastError());
        tick(2);
        nbnxm_gpu_launch_kernel(nb, stepWork, NBNXM_LOCAL);
        tick(3);
        NBNXM_ASSERT(nb->reductionAtomStart == 0 && nb->reductionNumAtoms >= numAtoms, "the cell map must cover home and halo atoms");
        hipLaunchKernelGGL(haloPushForcesKernel, blocks(numAtoms), dim3(c_pushThreads), 0, sLocal, h->d_f,
                           reinterpret_cast<const float3*>(nb->atdat->f), nb->cell, h->numHome, numAtoms, h->d_links, flags, seq, counter(2), h->d_pushError);
        NBNXM_HIP_CHECK(hipGetLastError());
        tick(6);
        if (shared) { worldBarrier(h); }
        hipLaunchKernelGGL(haloWaitUnpackAddKernel, blocks(h->numSendAtoms), dim3(c_pushThreads), 0, sLocal, reinterpret_cast<float*>(h->d_f),
                           h->d_recvBuf, h->d_sendMap, h->numSendAtoms, h->d_links, flags, seq, counter(3), h->d_pushError);
        NBNXM_HIP_CHECK(hipGetLastError());
        tick(8);
        /* (two barriers are enough: every kernel that waits — x to xq for the peers' coordinate stores, unpack-add for their force stores,
         * the next step's two storing kernels for this step's "consumed" flags — is then queued behind the kernels it waits for) */
        return;
    }
    if (nbnxm_gpu_get_merged_localities(nb))
    {
        /* Merged localities (nbnxm_gpu_set_merged_localities): ONE cluster-kernel launch evaluates home x home and home x halo, and
         * the whole step is ONE stream — the object's, which must then be nb's LOCAL stream:
         *     pack -> halo x -> x to xq (all slots, one kernel) -> clear (a pointer swap) -> merged kernel -> all rows of f to atom order
         *     -> halo f -> f_home += received
         * Measured on one GPU (96k home + 46k halo atoms, rocprofv3 kernel trace, profiles/r03): the two-stream schedule queues ~25 HIP
         * calls per step, a third of them event records and waits, and is bound by the host (0.10-0.11 ms to queue, 0.127 ms per step)
         * although its kernels need ~0.10 ms; every cross-stream dependency also costs the device a few microseconds of idle time.
         * One stream needs no event at all and one x -> xq launch instead of two.  What it gives up is the overlap of the coordinate
         * halo (~20 us: pack, transfer, x -> xq) with local pair work; what it gains besides is one start and drain of the machine per
         * step instead of two or three (81 us for the two cluster kernels of this domain, 58 us as one list). */
        NBNXM_ASSERT(h->stream == sLocal, "merged localities: the halo object must have been created on the LOCAL stream of the non-bonded object");
        nbnxm_gpu_set_local_launch_parts(nb, 1, h->localPartFraction);
        halo_gpu_communicate_coordinates(h, coordinatesReadyEvent);
        tick(0);
        nbnxm_gpu_clear_outputs(nb, stepWork->computeVirial);
        tick(1);
        nbnxm_gpu_x_to_nbat_x(nb, h->d_x, nullptr, NBNXM_LOCAL, 0, numSlots, 0);
        tick(2);
        nbnxm_gpu_launch_kernel(nb, stepWork, NBNXM_LOCAL);
        tick(3);
        NBNXM_ASSERT(nb->reductionAtomStart == 0 && nb->reductionNumAtoms >= numAtoms, "the cell map must cover home and halo atoms");
        hipLaunchKernelGGL(domainAllRowsKernel, dim3((numAtoms + c_haloThreadsPerBlock - 1) / c_haloThreadsPerBlock), dim3(c_haloThreadsPerBlock), 0,
                           sLocal, h->d_f, reinterpret_cast<const float3*>(nb->atdat->f), nb->cell, numAtoms);
        NBNXM_HIP_CHECK(hipGetLastError());
        tick(6);
        h->recvBufConsumedRecorded = false; /* one stream: the reader of the receive buffer is ordered by the stream */
        exchangeForces(h);
        tick(7);
        if (h->numSendAtoms > 0)
        {
            hipLaunchKernelGGL(haloUnpackForcesKernel<true>, dim3((h->numSendAtoms + c_haloThreadsPerBlock - 1) / c_haloThreadsPerBlock),
                               dim3(c_haloThreadsPerBlock), 0, sLocal, reinterpret_cast<float*>(h->d_f), h->d_recvBuf, h->d_sendMap, h->numSendAtoms);
            NBNXM_HIP_CHECK(hipGetLastError());
        }
        tick(8);
        return;
    }
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
                       reinterpret_cast<float*>(h->d_f), reinterpret_cast<const float3*>(nb->atdat->f), nb->cell, h->numHome, h->d_recvBuf,
                       h->d_sendMap, h->numSendAtoms);
    NBNXM_HIP_CHECK(hipGetLastError());
    /* the receive buffer was read on the LOCAL stream: the next step's force exchange (non-local stream) must not overtake it */
    NBNXM_HIP_CHECK(hipEventRecord(h->recvBufConsumed, sLocal));
    h->recvBufConsumedRecorded = true;
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
