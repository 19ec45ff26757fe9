// Micro-benchmark: what does a no-return float atomic cost on gfx950 by ACCESS SHAPE and by cache-policy bits?
//   shapes (per wave-instruction):  j24  = 24 lanes, 96 contiguous bytes (the cluster kernel's j-force add)
//                                   j48  = 48 lanes, 192 contiguous bytes (two adjoining j-clusters in one add)
//                                   i12  = 64 lanes at a 12-byte stride (the i-force add of round 2: 768 bytes touched)
//                                   c256 = 64 lanes, 256 contiguous bytes
//                                   drop = all lanes beyond the buffer (the kernel's dummy atomics)
//   bits: none (agent scope, memory side), sc0?, sc1, nt
// Every add is +1.0 into a force-array-sized buffer at a pseudo-random row; the sum of the buffer afterwards says whether adds
// were lost (an atomic executed in one XCD's L2 would lose adds to lines that another XCD holds).  With --xcd-copies every
// XCD adds into its own copy of the buffer (HW_REG_XCC_ID).
// Build: hipcc -O3 --offload-arch=gfx950 atomic_shapes.hip -o atomic_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int c_rows = 12744;          // j-clusters of the 96k box (101,952 slots / 8)
constexpr int c_rowBytes = 96;         // 8 atoms x 3 floats
constexpr int c_bufBytes = c_rows * c_rowBytes;

template<int BITS>
__device__ inline void atomAdd(float v, unsigned off, float* base)
{
    if (BITS == 0) asm volatile("global_atomic_add_f32 %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
    if (BITS == 1) asm volatile("global_atomic_add_f32 %0, %1, %2 sc1" ::"v"(off), "v"(v), "s"(base) : "memory");
    if (BITS == 2) asm volatile("global_atomic_add_f32 %0, %1, %2 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
    if (BITS == 3) asm volatile("global_atomic_add_f32 %0, %1, %2 sc1 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
}

// SHAPE: 0 j24, 1 j48, 2 i12 (three instructions, one per component), 3 c256, 4 drop (buffer atomic out of range), 5 j24 as buffer atomic
template<int SHAPE, int BITS>
__global__ void k(float* buf, int iters, int xcdCopies, int valuPerAtomic)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    float*         base = buf;
    if (xcdCopies)
    {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        base = buf + static_cast<size_t>(xcc & 7u) * (c_bufBytes / 4);
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, c_bufBytes, 0x00020000);
    unsigned h = wave * 2654435761u + 12345u;
    float    acc = lane * 0.5f;
    for (int it = 0; it < iters; it++)
    {
        h            = h * 1664525u + 1013904223u;
        unsigned row = (h >> 8) % (c_rows - 8);
        if (SHAPE == 0)
        {
            const unsigned tj = lane >> 3, ti = lane & 7u;
            if (ti < 3u) atomAdd<BITS>(1.0f, row * c_rowBytes + (tj * 3u + ti) * 4u, base);
        }
        else if (SHAPE == 1)
        {
            row &= ~1u;
            if (lane < 48u) atomAdd<BITS>(1.0f, row * c_rowBytes + lane * 4u, base);
        }
        else if (SHAPE == 2)
        {
            row &= ~7u;
            atomAdd<BITS>(1.0f, row * c_rowBytes + lane * 12u, base);
            atomAdd<BITS>(1.0f, row * c_rowBytes + lane * 12u + 4u, base);
            atomAdd<BITS>(1.0f, row * c_rowBytes + lane * 12u + 8u, base);
        }
        else if (SHAPE == 3)
        {
            row &= ~7u;
            atomAdd<BITS>(1.0f, row * c_rowBytes + lane * 4u, base);
        }
        else if (SHAPE == 4) { __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(1.0f, rsrc, 0x7FFFFFF0, 0, 0); }
        else if (SHAPE == 5)
        {
            const unsigned tj = lane >> 3, ti = lane & 7u;
            const int off = (ti < 3u) ? static_cast<int>(row * c_rowBytes + (tj * 3u + ti) * 4u) : 0x7FFFFFF0;
            __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(1.0f, rsrc, off, 0, 0);
        }
        for (int v = 0; v < valuPerAtomic; v++) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc)); }
    }
    if (acc == 123.456f) { buf[0] = acc; }
}

typedef void (*kern_t)(float*, int, int, int);

int main(int argc, char** argv)
{
    int       wavesPerSimd = 5;
    const int iters        = 200;
    float* buf;
    CHECK(hipMalloc(&buf, static_cast<size_t>(c_bufBytes) * 8));
    std::vector<float> h(static_cast<size_t>(c_bufBytes) / 4 * 8);
    struct V { const char* name; kern_t fn; double addsPerIter; double requestsPerIter; };
    const V vs[] = {
        { "j24  96 B (2 lines)        agent", k<0, 0>, 24, 2 },
        { "j24  96 B                  sc1  ", k<0, 1>, 24, 2 },
        { "j24  96 B                  nt   ", k<0, 2>, 24, 2 },
        { "j24  96 B                  sc1nt", k<0, 3>, 24, 2 },
        { "j24  96 B  buffer atomic   agent", k<5, 0>, 24, 2 },
        { "j48  192 B (3 lines)       agent", k<1, 0>, 48, 3 },
        { "i12  3 x 64 @12 B (36 ln)  agent", k<2, 0>, 192, 36 },
        { "c256 256 B (4 lines)       agent", k<3, 0>, 64, 4 },
        { "c256 256 B                 sc1  ", k<3, 1>, 64, 4 },
        { "c256 256 B                 nt   ", k<3, 2>, 64, 4 },
        { "drop (all lanes out of range)   ", k<4, 0>, 0, 0 },
    };
    for (int valu = 0; valu <= 64; valu += 64)
    {
        for (int copies = 0; copies < 2; copies++)
        {
            printf("--- %d waves/SIMD, %d v_fma between atomics, %s ---\n", wavesPerSimd, valu, copies ? "one buffer copy per XCD" : "one buffer");
            for (const V& v : vs)
            {
                const int blocks = 256 * wavesPerSimd;
                CHECK(hipMemset(buf, 0, static_cast<size_t>(c_bufBytes) * 8));
                hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, buf, 4, copies, valu);
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemset(buf, 0, static_cast<size_t>(c_bufBytes) * 8));
                CHECK(hipDeviceSynchronize());
                hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(v.fn, dim3(blocks), dim3(256), 0, 0, buf, iters, copies, valu);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipMemcpy(h.data(), buf, static_cast<size_t>(c_bufBytes) * 8, hipMemcpyDeviceToHost));
                double sum = 0;
                for (float x : h) sum += x;
                const double waves    = blocks * 4.0;
                const double expected = waves * iters * v.addsPerIter;
                const double nsPerIterPerCu = ms * 1e6 / (iters * wavesPerSimd * 4.0);
                printf("%s  %.3f ms  %7.1f ns per iteration per CU  %6.2f ns per 64-B line  sum/expected %.6f\n", v.name, ms, nsPerIterPerCu,
                       v.requestsPerIter > 0 ? nsPerIterPerCu / v.requestsPerIter : 0.0, expected > 0 ? sum / expected : 0.0);
            }
        }
    }
    return 0;
}
