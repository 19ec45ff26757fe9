// Micro-benchmark: does gfx950 issue a wave64 VALU instruction faster when part of the wave is masked off by EXEC?
// (lanes 0-31 only, lanes 0-15 only, even lanes only, one lane) against all 64 lanes.  Build: hipcc -O3 --offload-arch=gfx950 valu_exec.hip -o valu_exec
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP16(X) X X X X X X X X X X X X X X X X
__global__ void k(float* out, int iters, int mode)
{
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float b = 1.0001f, c = 0.5f;
    const unsigned lane = threadIdx.x & 63u;
    bool on = true;
    if (mode == 1) on = lane < 32u;
    if (mode == 2) on = lane < 16u;
    if (mode == 3) on = (lane & 1u) == 0u;
    if (mode == 4) on = lane == 0u;
    if (mode == 5) on = lane >= 32u;
    if (mode == 6) on = (lane & 16u) == 0u;   // rows 0 and 2
    if (on)
    {
        for (int it = 0; it < iters; it++)
        {
            REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
                  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
int main()
{
    const int iters = 2000, wavesPerSimd = 5, blocks = 256 * wavesPerSimd;
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
    const char* names[] = { "all 64 lanes", "lanes 0-31", "lanes 0-15", "even lanes", "lane 0", "lanes 32-63", "rows 0 and 2" };
    for (int mode = 0; mode < 7; mode++)
    {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 10, mode);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, mode);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-14s %.3f ms  %.3f ns per wave64 v_fma per SIMD\n", names[mode], ms, ms * 1e6 / (64.0 * iters * wavesPerSimd));
    }
    return 0;
}
