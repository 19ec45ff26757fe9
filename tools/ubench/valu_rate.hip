// Micro-benchmark: issue rate of wave64 VALU instruction kinds on gfx950 (cycles per instruction per SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP16(X) X X X X X X X X X X X X X X X X
template<int KIND>
__global__ void k(float* out, int iters, unsigned long long* cyc)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.001f + i;
    float b = 1.0001f, c = 0.5f;
    int   m = threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++)
    {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[2]) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(b), "v"(c));) }
        if (KIND == 1) { REP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[1]) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[2]) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[3]) : "v"(b));) }
        if (KIND == 2) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[1]) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[2]) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[3]) : "v"(b));) }
        if (KIND == 3) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[0]) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[1]) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[2]) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[3]) : "v"(b));) }
        if (KIND == 4) { REP16(asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[0]) : "v"(m)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[1]) : "v"(m)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[2]) : "v"(m)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[3]) : "v"(m));) }
        if (KIND == 5) { REP16(asm volatile("v_rsq_f32 %0, %0" : "+v"(a[0])); asm volatile("v_rsq_f32 %0, %0" : "+v"(a[1])); asm volatile("v_rsq_f32 %0, %0" : "+v"(a[2])); asm volatile("v_rsq_f32 %0, %0" : "+v"(a[3]));) }
        if (KIND == 6) { REP16(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[0])); asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[1])); asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[2])); asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[3]));) }
        if (KIND == 7) { REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[0]), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[1]), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[2]), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[3]), "v"(b) : "vcc");) }
        if (KIND == 8) { REP16(asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000000" : "+v"(a[0]) : "v"(b)); asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000000" : "+v"(a[1]) : "v"(b)); asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000000" : "+v"(a[2]) : "v"(b)); asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f000000" : "+v"(a[3]) : "v"(b));) }
        if (KIND == 9) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[0]) : "v"(*(double*)&a[4])); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[2]) : "v"(*(double*)&a[6])); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[0]) : "v"(*(double*)&a[4])); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[2]) : "v"(*(double*)&a[6]));) }
        if (KIND == 10) { REP16(asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[1]) : "v"(b), "v"(c)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[2]) : "v"(b)); asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[3]) : "v"(m));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template<int KIND>
int run(const char* name, int wavesPerSimd)
{
    const int iters = 2000, blocks = 256 * 4 * wavesPerSimd / 4; // 256-thread blocks (4 waves): blocks = CUs*wavesPerSimd
    float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double instrPerWave = 64.0 * iters;          // 16 x 4 per iteration
    // s_memtime ticks at 100 MHz-ish constant clock? report both wall-based and tick-based
    const double wavesPerSimdTotal = wavesPerSimd;
    const double nsPerInstrPerSimd = ms * 1e6 / (instrPerWave * wavesPerSimdTotal);
    printf("%-14s waves/SIMD %d: %.3f ms, %.3f ns per wave64 instr per SIMD (= %.2f cycles @2.4GHz), memtime ticks/instr/wave %.2f\n",
           name, wavesPerSimd, ms, nsPerInstrPerSimd, nsPerInstrPerSimd * 2.4, avg / instrPerWave);
    hipFree(out); hipFree(cyc);
    return 0;
}

int main()
{
    for (int w : { 1, 2, 4, 8 })
    {
        run<0>("v_fma_f32", w); run<1>("v_mul_f32", w); run<2>("v_add_f32", w); run<3>("v_cndmask", w);
        run<4>("v_and_b32", w); run<5>("v_rsq_f32", w); run<6>("v_add_dpp", w); run<7>("v_cmp_f32", w);
        run<8>("v_fmaak_f32", w); run<9>("v_pk_fma_f32", w); run<10>("mixed", w);
    }
    return 0;
}
