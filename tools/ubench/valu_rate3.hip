// Micro-benchmark (round 3): issue rate of the instruction forms of the cluster kernel's pair block on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP16(X) X X X X X X X X X X X X X X X X
template<int KIND>
__global__ void k(float* out, int iters, unsigned long long mskIn, float scIn)
{
    float a[4];
    for (int i = 0; i < 4; i++) a[i] = threadIdx.x * 0.001f + i + 1.0f;
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
    unsigned long long d[3] = { threadIdx.x, threadIdx.x + 7ull, 12345ull };
    float b = 1.0001f, c = 0.5f;
    int   m = (threadIdx.x * 4) & 255;
    unsigned long long msk = __builtin_amdgcn_readfirstlane((unsigned)mskIn) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mskIn >> 32)) << 32);
    unsigned long long msk2 = 0; int sr = 0;
    float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scIn)));
    for (int it = 0; it < iters; it++)
    {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[2]) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(b), "v"(c));) }
        if (KIND == 1) { REP16(asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[0]) : "s"(sc), "v"(c)); asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[1]) : "s"(sc), "v"(c)); asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[2]) : "s"(sc), "v"(c)); asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[3]) : "s"(sc), "v"(c));) }
        if (KIND == 2) { REP16(asm volatile("v_fmaak_f32 %0, %0, %1, 0x4b000000" : "+v"(a[0]) : "v"(b)); asm volatile("v_fmaak_f32 %0, %0, %1, 0x4b000000" : "+v"(a[1]) : "v"(b)); asm volatile("v_fmaak_f32 %0, %0, %1, 0x4b000000" : "+v"(a[2]) : "v"(b)); asm volatile("v_fmaak_f32 %0, %0, %1, 0x4b000000" : "+v"(a[3]) : "v"(b));) }
        if (KIND == 3) { REP16(asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[0]) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[1]) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[2]) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[3]) : "v"(b), "v"(c));) }
        if (KIND == 4) { REP16(asm volatile("v_and_b32 %0, 0x3ff8, %0" : "+v"(a[0])); asm volatile("v_and_b32 %0, 0x3ff8, %0" : "+v"(a[1])); asm volatile("v_and_b32 %0, 0x3ff8, %0" : "+v"(a[2])); asm volatile("v_and_b32 %0, 0x3ff8, %0" : "+v"(a[3]));) }
        if (KIND == 5) { REP16(asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[0]) : "v"(b)); asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[1]) : "v"(b)); asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[2]) : "v"(b)); asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[3]) : "v"(b));) }
        if (KIND == 6) { REP16(asm volatile("v_bfe_i32 %0, %0, 2, 1" : "+v"(a[0])); asm volatile("v_bfe_i32 %0, %0, 2, 1" : "+v"(a[1])); asm volatile("v_bfe_i32 %0, %0, 2, 1" : "+v"(a[2])); asm volatile("v_bfe_i32 %0, %0, 2, 1" : "+v"(a[3]));) }
        if (KIND == 7) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(m)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[1]) : "v"(m)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[2]) : "v"(m)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[3]) : "v"(m));) }
        if (KIND == 8) { REP16(asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b)); asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[1]) : "v"(b)); asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[2]) : "v"(b)); asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[3]) : "v"(b));) }
        if (KIND == 9) { REP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[1]) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[2]) : "v"(b)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[3]) : "v"(b));) }
        if (KIND == 10) { REP16(asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[0])); asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[1])); asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[2])); asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[3]));) }
        if (KIND == 11) { REP16(asm volatile("v_fract_f32 %0, %0" : "+v"(a[0])); asm volatile("v_fract_f32 %0, %0" : "+v"(a[1])); asm volatile("v_fract_f32 %0, %0" : "+v"(a[2])); asm volatile("v_fract_f32 %0, %0" : "+v"(a[3]));) }
        if (KIND == 12) { REP16(asm volatile("v_rsq_f32 %0, %0" : "+v"(a[0])); asm volatile("v_rsq_f32 %0, %0" : "+v"(a[1])); asm volatile("v_rsq_f32 %0, %0" : "+v"(a[2])); asm volatile("v_rsq_f32 %0, %0" : "+v"(a[3]));) }
        if (KIND == 13) { REP16(asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(msk2) : "s"(sc), "v"(a[0])); asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(msk2) : "s"(sc), "v"(a[1])); asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(msk2) : "s"(sc), "v"(a[2])); asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(msk2) : "s"(sc), "v"(a[3]));) }
        if (KIND == 14) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[0]) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[1]) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[2]) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[3]) : "v"(b));) }
        if (KIND == 15) { REP16(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[0])); asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[1])); asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[2])); asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[3]));) }
        if (KIND == 16) { REP16(asm volatile("v_mov_b32 %0, %1" : "=v"(a[0]) : "v"(b)); asm volatile("v_mov_b32 %0, %1" : "=v"(a[1]) : "v"(b)); asm volatile("v_mov_b32 %0, %1" : "=v"(a[2]) : "v"(b)); asm volatile("v_mov_b32 %0, %1" : "=v"(a[3]) : "v"(b));) }
        if (KIND == 17) { REP16(asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[0])); asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[1])); asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[2])); asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[3]));) }
        if (KIND == 18) { REP16(asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[0])); asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[1])); asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[2])); asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a[3]));) }
        if (KIND == 19) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[0]) : "v"(m)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[1]) : "v"(m)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[2]) : "v"(m)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[3]) : "v"(m));) }
    }
    float s = (float)d[0] + (float)d[1] + (float)msk2 + sr;
    for (int i = 0; i < 4; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template<int KIND>
int run(const char* name, int wavesPerSimd)
{
    const int iters = 1000, blocks = 256 * wavesPerSimd;
    float* out;
    CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, 0x5555555555555555ull, 1.0001f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 0x5555555555555555ull, 1.0001f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / (64.0 * iters * wavesPerSimd);
    printf("%-16s waves/SIMD %d: %.3f ns per wave-instr per SIMD\n", name, wavesPerSimd, ns);
    (void)hipFree(out);
    return 0;
}
int main()
{
    for (int w : { 1, 5 }) {
        run<0>("v_fma_vvv", w);
        run<1>("v_fma_sgpr", w);
        run<2>("v_fmaak_literal", w);
        run<3>("v_fma_neg", w);
        run<4>("v_and_literal", w);
        run<5>("v_and_vv", w);
        run<6>("v_bfe_i32", w);
        run<7>("v_add_u32", w);
        run<8>("v_sub_f32", w);
        run<9>("v_mul_f32", w);
        run<10>("v_cvt_u32_f32", w);
        run<11>("v_fract_f32", w);
        run<12>("v_rsq_f32", w);
        run<13>("v_cmp_gt_sgpr_src", w);
        run<14>("v_cndmask_vcc", w);
        run<15>("v_add_dpp_quad", w);
        run<16>("v_mov_b32", w);
        run<17>("v_lshlrev_b32", w);
        run<18>("v_ashrrev_i32", w);
        run<19>("v_mul_lo_u32", w);
    }
    return 0;
}
