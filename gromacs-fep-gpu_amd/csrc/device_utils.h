/*
 * Thin HIP versions of the gpu_utils wrappers the Nbnxm path uses in the reference:
 *   DeviceBuffer<T> + allocate/free/reallocate/copyTo/copyFrom/clearDeviceBufferAsync
 *       (gpu_utils/devicebuffer.cuh:66-300, devicebuffer_datatype.h:55-56: DeviceBuffer<T> = T*)
 *   DeviceStream (gpu_utils/device_stream.h:91), pinned host allocation (gpu_utils/pmalloc.h)
 *   GpuRegionTimer (gpu_utils/gpuregiontimer.h:64) on hipEvents
 * Errors are fatal, like GMX_RELEASE_ASSERT / gmx_fatal in the reference.
 */
#ifndef NBNXM_DEVICE_UTILS_H
#define NBNXM_DEVICE_UTILS_H

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace nbnxm_hip
{

void setLastError(const char* msg);

[[noreturn]] inline void fatal(const char* file, int line, const char* what, const char* detail)
{
    char buf[1024];
    std::snprintf(buf, sizeof(buf), "nbnxm_hip fatal error at %s:%d: %s%s%s", file, line, what,
                  detail ? ": " : "", detail ? detail : "");
    setLastError(buf);
    std::fprintf(stderr, "%s\n", buf);
    std::fflush(stderr);
    /* a test runner that captures file descriptor 2 loses the line above when the process aborts (that is how an assertion of
     * round 2 came to leave no trace in the GPU-box log): NBNXM_HIP_FATAL_LOG names a file that gets it too */
    if (const char* logName = std::getenv("NBNXM_HIP_FATAL_LOG"))
    {
        if (FILE* log = std::fopen(logName, "a"))
        {
            std::fprintf(log, "%s\n", buf);
            std::fclose(log);
        }
    }
    std::abort();
}

/* Environment variables that change what the library does exist for measurements (A/B runs of one build) only: they are read when
 * NBNXM_HIP_DIAGNOSTICS=1 and ignored otherwise, so that a drop-in's behaviour does not hinge on the environment it happens to run in
 * (INTEGRATION.md has the table).  Not gated: NBNXM_HIP_FATAL_LOG, NBNXM_HIP_RCCL_LIB, HALO_GPU_PEER_TIMEOUT — where to write, what to
 * load, how long to wait; none of them changes a result or a launch. */
inline const char* diagnosticsEnv(const char* name)
{
    /* (read at every call — object creation, a handful per run —, so that a test can switch it on for one object) */
    const char* v = std::getenv("NBNXM_HIP_DIAGNOSTICS");
    return (v != nullptr && v[0] == '1') ? std::getenv(name) : nullptr;
}

#define NBNXM_HIP_CHECK(expr)                                                                        \
    do                                                                                               \
    {                                                                                                \
        hipError_t nbnxmHipStatus_ = (expr);                                                         \
        if (nbnxmHipStatus_ != hipSuccess)                                                           \
        {                                                                                            \
            ::nbnxm_hip::fatal(__FILE__, __LINE__, #expr, hipGetErrorString(nbnxmHipStatus_));       \
        }                                                                                            \
    } while (0)

#define NBNXM_ASSERT(cond, msg)                                                                      \
    do                                                                                               \
    {                                                                                                \
        if (!(cond)) { ::nbnxm_hip::fatal(__FILE__, __LINE__, "assertion failed: " #cond, msg); }    \
    } while (0)

template<typename T>
using DeviceBuffer = T*;

template<typename T>
void allocateDeviceBuffer(DeviceBuffer<T>* buffer, size_t numValues)
{
    NBNXM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(buffer), std::max<size_t>(1, numValues) * sizeof(T)));
}

template<typename T>
void freeDeviceBuffer(DeviceBuffer<T>* buffer)
{
    if (*buffer != nullptr)
    {
        NBNXM_HIP_CHECK(hipFree(*buffer));
        *buffer = nullptr;
    }
}

/* Grows with 20 % over-allocation (over_alloc_large in the reference), contents are NOT kept. */
template<typename T>
void reallocateDeviceBuffer(DeviceBuffer<T>* buffer, size_t numValues, int* currentNumValues, int* currentMaxNumValues)
{
    if (static_cast<long long>(numValues) > *currentMaxNumValues)
    {
        freeDeviceBuffer(buffer);
        *currentMaxNumValues = static_cast<int>(numValues * 1.2 + 1024);
        allocateDeviceBuffer(buffer, *currentMaxNumValues);
    }
    *currentNumValues = static_cast<int>(numValues);
}

template<typename T>
void copyToDeviceBuffer(DeviceBuffer<T>* buffer, const T* hostBuffer, size_t startingOffset, size_t numValues,
                        hipStream_t stream, bool async)
{
    if (numValues == 0) { return; }
    NBNXM_HIP_CHECK(hipMemcpyAsync(*buffer + startingOffset, hostBuffer, numValues * sizeof(T),
                                   hipMemcpyHostToDevice, stream));
    if (!async) { NBNXM_HIP_CHECK(hipStreamSynchronize(stream)); }
}

/* Is this host pointer page-locked memory the runtime knows (hipHostMalloc / hipHostRegister — what the reference's HostVector
 * with its pinning allocator hands over)?  Then a hipMemcpyAsync from it is a DMA of its own and needs no staging copy. */
inline bool isPinnedHostMemory(const void* p)
{
    if (p == nullptr) { return false; }
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess)
    {
        (void)hipGetLastError(); /* pageable memory is "invalid value" here: an answer, not an error */
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

template<typename T>
void copyFromDeviceBuffer(T* hostBuffer, DeviceBuffer<T>* buffer, size_t startingOffset, size_t numValues,
                          hipStream_t stream, bool async)
{
    if (numValues == 0) { return; }
    NBNXM_HIP_CHECK(hipMemcpyAsync(hostBuffer, *buffer + startingOffset, numValues * sizeof(T),
                                   hipMemcpyDeviceToHost, stream));
    if (!async) { NBNXM_HIP_CHECK(hipStreamSynchronize(stream)); }
}

template<typename T>
void clearDeviceBufferAsync(DeviceBuffer<T>* buffer, size_t startingOffset, size_t numValues, hipStream_t stream)
{
    if (numValues == 0) { return; }
    NBNXM_HIP_CHECK(hipMemsetAsync(*buffer + startingOffset, 0, numValues * sizeof(T), stream));
}

/* pinned, page-locked host memory that outlives the async copies reading it */
template<typename T>
struct PinnedBuffer
{
    T*     data = nullptr;
    size_t size = 0, capacity = 0;
    void   resize(size_t n)
    {
        if (n > capacity)
        {
            release();
            capacity = static_cast<size_t>(n * 1.2) + 256;
            NBNXM_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&data), capacity * sizeof(T), hipHostMallocDefault));
        }
        size = n;
    }
    void release()
    {
        if (data) { NBNXM_HIP_CHECK(hipHostFree(data)); }
        data     = nullptr;
        capacity = size = 0;
    }
    ~PinnedBuffer() { if (data) { (void)hipHostFree(data); } }
};

struct DeviceStream
{
    hipStream_t stream = nullptr;
    bool        owned  = false;
    /* highPriority: the non-local stream, as DeviceStreamManager creates it (gpu_utils/device_stream_manager.cpp:107-108: NonBondedNonLocal is
     * DeviceStreamPriority::High), so that the work the force halo waits for is dispatched ahead of whatever else is queued */
    void        init(void* external, bool highPriority = false)
    {
        if (external) { stream = static_cast<hipStream_t>(external); }
        else
        {
            if (highPriority)
            {
                int lo = 0, hi = 0;
                NBNXM_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
                NBNXM_HIP_CHECK(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, hi));
            }
            else { NBNXM_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); }
            owned = true;
        }
    }
    void synchronize() const { NBNXM_HIP_CHECK(hipStreamSynchronize(stream)); }
    bool completed() const
    {
        hipError_t s = hipStreamQuery(stream);
        if (s == hipSuccess) { return true; }
        if (s != hipErrorNotReady) { NBNXM_HIP_CHECK(s); }
        return false;
    }
    void destroy()
    {
        if (owned && stream) { NBNXM_HIP_CHECK(hipStreamDestroy(stream)); }
        stream = nullptr;
        owned  = false;
    }
};

/* One open/close pair per step; elapsed time is read after the stream has been synchronised. */
struct GpuRegionTimer
{
    hipEvent_t start = nullptr, stop = nullptr;
    bool       pending = false;
    double     totalMs = 0;
    int        count   = 0;
    void       init()
    {
        NBNXM_HIP_CHECK(hipEventCreate(&start));
        NBNXM_HIP_CHECK(hipEventCreate(&stop));
    }
    void openTimingRegion(hipStream_t s) { NBNXM_HIP_CHECK(hipEventRecord(start, s)); }
    void closeTimingRegion(hipStream_t s)
    {
        NBNXM_HIP_CHECK(hipEventRecord(stop, s));
        pending = true;
    }
    void accumulate()
    {
        if (!pending) { return; }
        float ms = 0;
        NBNXM_HIP_CHECK(hipEventSynchronize(stop));
        NBNXM_HIP_CHECK(hipEventElapsedTime(&ms, start, stop));
        totalMs += ms;
        count++;
        pending = false;
    }
    void reset()
    {
        totalMs = 0;
        count   = 0;
        pending = false;
    }
    void destroy()
    {
        if (start) { (void)hipEventDestroy(start); }
        if (stop) { (void)hipEventDestroy(stop); }
        start = stop = nullptr;
    }
};

} // namespace nbnxm_hip

#endif
