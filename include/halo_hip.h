/*
 * halo_hip.h — C ABI of the GPU halo exchange of a domain-decomposed run on MI355X: coordinates out to the ranks that need
 * them and forces back, over RCCL point-to-point (ncclSend / ncclRecv groups, xGMI inside a node), stream-ordered on the
 * caller's non-local stream so that it overlaps the local non-bonded kernel.
 *
 * Replaces gmx::GpuHaloExchange (domdec/gpuhaloexchange.h:75-160; implementation domdec/gpuhaloexchange_impl_gpu.cpp:122-511,
 * kernels domdec/gpuhaloexchange_impl_gpu.cu:62-183):
 *     reinitHalo(d_x, d_f)                               -> halo_gpu_reinit
 *     communicateHaloCoordinates(box, dependencyEvent)   -> halo_gpu_communicate_coordinates
 *     communicateHaloForces(accumulate, dependencyEvents)-> halo_gpu_communicate_forces
 *     getForcesReadyOnDeviceEvent()                      -> halo_gpu_forces_ready_event
 * MI355X-first differences: the reference has one object per (dimension, pulse) and forwards halo atoms dimension by dimension
 * (MPI_Sendrecv of device pointers, or peer copies plus an MPI handshake of events per pulse).  xGMI links every GPU of a node to
 * every other, so ONE object per rank talks to all its neighbours directly: one pack kernel for all destinations, one
 * ncclGroup of sends and receives (coordinates land in place in the receiver's coordinate array, no unpack), one unpack-and-add
 * kernel for the forces.  A "link" is one (peer, periodic image) pair; the coordinate shift of the image is applied while
 * packing (packSendBufKernel's usePbc path).
 *
 * librccl is opened at run time (dlopen: the copy already in the process — PyTorch brings one — or /opt/rocm/lib/librccl.so.1);
 * this library has no link-time dependency on it, and single-GPU users never load it.
 */
#ifndef HALO_HIP_H
#define HALO_HIP_H

#include "nbnxm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct HaloGpu HaloGpu;

#define HALO_GPU_UNIQUE_ID_BYTES 128

/* ncclGetUniqueId: call on ONE rank and hand the 128 bytes to all ranks (any out-of-band channel).  Returns 0 on success. */
int halo_gpu_get_unique_id(void* uniqueId);

/* The transport is chosen with the id.  HALO_GPU_TRANSPORT_RCCL: as above, one process per rank.  HALO_GPU_TRANSPORT_PEER_COPY: the
 * ranks are host threads of ONE process (the reference's thread-MPI build, domdec/gpuhaloexchange_impl_gpu.cpp:438-511): a receiver
 * copies device to device out of the sender's buffer (hipMemcpyAsync, hipMemcpyPeerAsync between the devices of a node) behind the
 * event the sender recorded; the event handles go through a mailbox in host memory, where the reference sends them with MPI_Sendrecv.
 * Every call that exchanges data blocks its host thread until the peers have QUEUED their side of that exchange, so each rank needs
 * its own host thread (a rank that only talks to itself needs none); a peer that never arrives is a fatal error after
 * HALO_GPU_PEER_TIMEOUT seconds (default 30), not a hang.  Any number of ranks can share one GPU with this transport — RCCL refuses
 * that —, which is how the multi-rank schedule of halo_gpu_domain_force_step is tested on a one-GPU box. */
#define HALO_GPU_TRANSPORT_RCCL 0
#define HALO_GPU_TRANSPORT_PEER_COPY 1
/* HALO_GPU_TRANSPORT_PEER_PUSH: one-sided.  Ranks as with PEER_COPY (host threads of one process, one device each on a node, peer access
 * between the devices); but nothing is copied and no event is exchanged: the pack kernel of the sender stores the shifted coordinates
 * straight into the receiver's halo rows and publishes the step's sequence number in a flag in the receiver's memory, the receiver's
 * x -> xq kernel waits for the flags of its links; forces return the same way (forces-to-atom-order kernel -> the owners' receive
 * buffers, their unpack-add kernel waits).  For sub-megabyte, latency-bound messages between GPUs that are all linked to each other
 * this is the native form: five kernels per step on one stream, no transfer kernel, no group call.  The exchanges live inside
 * halo_gpu_domain_force_step with merged localities; halo_gpu_communicate_coordinates / _forces are not available.
 * halo_gpu_reinit is a rendezvous of the ranks here (every rank calls it at every search step).  Ranks sharing a device (tests)
 * are kept deadlock-free by host barriers between the phases of a step. */
#define HALO_GPU_TRANSPORT_PEER_PUSH 2
/* HALO_GPU_TRANSPORT_IPC_PUSH: the same one-sided exchange between PROCESSES, one per GPU (the bench's and mdrun's shape).  The peers'
 * buffers are opened through hipIpc handles: after every halo_gpu_reinit each rank calls halo_gpu_push_export, the ranks exchange the
 * records over whatever channel they have (an all-gather: MPI, torch.distributed), and each calls halo_gpu_push_import with all of
 * them, in rank order.  The coordinates are stored into a buffer of this library (exportable, unlike the caller's array); the waiting
 * x -> xq kernel copies them into the halo rows of d_x as well.  halo_gpu_get_unique_id_ex needs no agreement for this transport. */
#define HALO_GPU_TRANSPORT_IPC_PUSH 3
int halo_gpu_get_unique_id_ex(void* uniqueId, int transport);

/* RCCL id: ncclCommInitRank (collective over the nranks); peer-copy id: attaches to the process's mailbox of that id (not collective).
 * stream: the hipStream_t the exchanges are queued on (the non-local stream of the non-bonded module).  Returns NULL on failure
 * (halo_gpu_last_error() has the text). */
HaloGpu* halo_gpu_create(const void* uniqueId, int rank, int nranks, void* stream);
void     halo_gpu_free(HaloGpu* h);
/* HALO_GPU_TRANSPORT_IPC_PUSH: the bytes of one export record; this rank's record (after halo_gpu_reinit); all ranks' records in rank order.
 * Return 0 on success (halo_gpu_last_error() has the text otherwise).  halo_gpu_push_status: 0, or 1 + the link a kernel of the one-sided
 * transports gave up waiting for (a peer that never stored its side; the step's results are void, the next call into the object is fatal). */
int      halo_gpu_push_export_bytes(void);
int      halo_gpu_push_export(HaloGpu* h, void* record);
int      halo_gpu_push_import(HaloGpu* h, const void* records, int numRanks);
int      halo_gpu_push_status(const HaloGpu* h);
const char* halo_gpu_last_error(void);

/* reinitHalo — after every domain repartitioning / search.
 *  d_x, d_f        float3 arrays in the rank's atom order: home atoms [0, numHome), then the halo blocks
 *  send side: numSend destinations; destination k gets the home atoms sendMap[sendOffset[k] .. sendOffset[k+1]) in that order;
 *             entry i is shifted by shiftVectors[3 * sendShiftIndex[i] ..] while packing (the periodic image the receiver sees)
 *  recv side: numRecv sources; source k's atoms occupy x/f rows [recvAtomOffset[k], recvAtomOffset[k] + recvCount[k])
 * Host arrays are copied; the maps go to the device. */
void halo_gpu_reinit(HaloGpu* h, void* d_x, void* d_f, int numHome, int numSend, const int* sendPeer, const int* sendOffset,
                     const int* sendMap, const int* sendShiftIndex, int numShiftVectors, const float* shiftVectors, int numRecv,
                     const int* recvPeer, const int* recvAtomOffset, const int* recvCount);

/* communicateHaloCoordinates: [wait dependencyEvent (hipEvent_t or NULL: coordinates updated)] pack -> group(send, recv) on the
 * object's stream.  The halo rows of d_x are valid for work queued on that stream afterwards. */
void halo_gpu_communicate_coordinates(HaloGpu* h, void* dependencyEvent);

/* communicateHaloForces: [wait dependencyEvent] group(send the halo rows of d_f to their owners, receive what the others
 * computed on this rank's atoms) -> f[sendMap[i]] += received (accumulate != 0) or = received. */
void halo_gpu_communicate_forces(HaloGpu* h, int accumulate, void* dependencyEvent);

/* recorded on the object's stream after the last communicate_* call (hipEvent_t) */
void* halo_gpu_coordinates_ready_event(HaloGpu* h);
void* halo_gpu_forces_ready_event(HaloGpu* h);

/* The whole force step of one domain in one call (host side in C++, as in the reference's do_force): halo x beside the local kernel,
 * x -> xq per locality, local and non-local non-bonded kernels on their two streams, forces to atom order per locality, halo f added
 * to the home rows (mdlib/sim_util.cpp:1783-1924).  nb: the domain's non-bonded object (include/nbnxm_hip.h) with both localities, its
 * cell map uploaded with nbnxm_gpu_force_reduction_reinit(nb, numAtoms, cell, 0, .) and its atom indices with
 * nbnxm_gpu_init_x_to_nbat_x; numHomeSlots / numSlots: grid slots of the home zone / of both zones; numAtoms: home + halo atoms.
 * h must have been created on nb's non-local stream — on its LOCAL stream when nb runs merged localities (nbnxm_gpu_set_merged_localities):
 * the step is then one stream and one cluster-kernel launch, pack -> halo x -> x to xq -> kernel -> rows of f -> halo f -> add.
 * coordinatesReadyEvent (hipEvent_t or NULL): the home rows of d_x have been
 * updated (UpdateConstrainGpu's xUpdatedOnDeviceEvent) — both streams wait for it before they read coordinates.  On return everything
 * is queued; the home rows of d_f are final on nb's local stream. */
void halo_gpu_domain_force_step(HaloGpu* h, NbnxmGpu* nb, const nbnxm_step_workload_t* stepWork, int numHomeSlots, int numSlots, int numAtoms,
                                void* coordinatesReadyEvent);

/* bytes this rank sends per step (coordinates out + forces back), for reporting */
long long halo_gpu_bytes_per_step(const HaloGpu* h);

/* The pack and unpack-add kernels alone, on caller-provided buffers (tests, and transports other than RCCL):
 *   packed[i] = x[map[i]] + shiftVectors[shiftIndex[i]]            f[map[i]] (+)= packed[i] */
void halo_gpu_pack_shifted(void* stream, const void* d_x, const int* d_map, const int* d_shiftIndex, int n, const float* d_shiftVectors,
                           void* d_packed);

#ifdef __cplusplus
}
#endif
#endif
