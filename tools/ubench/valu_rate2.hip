// Micro-benchmark (round 2): issue rate of more wave64 instruction kinds on gfx950.
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
    unsigned long long d[3] = { threadIdx.x, threadIdx.x + 7ull, 12345ull };
    float b = 1.0001f, c = 0.5f;
    int   m = (threadIdx.x * 4) & 255;
    unsigned long long msk = __builtin_amdgcn_readfirstlane((unsigned)mskIn) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mskIn >> 32)) << 32);
    unsigned long long msk2 = 0; int sr = 0;
    float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, scIn)));
    for (int it = 0; it < iters; it++)
    {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[2]) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(b), "v"(c));) }
        if (KIND == 1) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[0]) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[1]) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[2]) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[3]) : "v"(b));) }
        if (KIND == 2) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "s"(msk)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[1]) : "v"(b), "s"(msk)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[2]) : "v"(b), "s"(msk)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[3]) : "v"(b), "s"(msk));) }
        if (KIND == 3) { REP16(asm volatile("v_cndmask_b32_e64 %0, 1.0, 0, %1" : "=v"(a[0]) : "s"(msk)); asm volatile("v_cndmask_b32_e64 %0, 1.0, 0, %1" : "=v"(a[1]) : "s"(msk)); asm volatile("v_cndmask_b32_e64 %0, 1.0, 0, %1" : "=v"(a[2]) : "s"(msk)); asm volatile("v_cndmask_b32_e64 %0, 1.0, 0, %1" : "=v"(a[3]) : "s"(msk));) }
        if (KIND == 4) { REP16(asm volatile("v_mov_b32 %0, %1" : "=v"(a[0]) : "v"(b)); asm volatile("v_mov_b32 %0, %1" : "=v"(a[1]) : "v"(b)); asm volatile("v_mov_b32 %0, %1" : "=v"(a[2]) : "v"(b)); asm volatile("v_mov_b32 %0, %1" : "=v"(a[3]) : "v"(b));) }
        if (KIND == 5) { REP16(asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[1]) : "v"(b)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[2]) : "v"(b)); asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[3]) : "v"(b));) }
        if (KIND == 6) { REP16(asm volatile("v_rcp_f32 %0, %0" : "+v"(a[0])); asm volatile("v_rcp_f32 %0, %0" : "+v"(a[1])); asm volatile("v_rcp_f32 %0, %0" : "+v"(a[2])); asm volatile("v_rcp_f32 %0, %0" : "+v"(a[3]));) }
        if (KIND == 7) { REP16(asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[0])); asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[1])); asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[2])); asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(a[3]));) }
        if (KIND == 8) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[0]) : "v"(m)); asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[1]) : "v"(m)); asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[2]) : "v"(m)); asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[3]) : "v"(m));) }
        if (KIND == 9) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(d[0&1]) : "v"(d[2])); asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(d[1&1]) : "v"(d[2])); asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(d[2&1]) : "v"(d[2])); asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(d[3&1]) : "v"(d[2]));) }
        if (KIND == 10) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[0&1]) : "v"(m), "v"(m) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[1&1]) : "v"(m), "v"(m) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[2&1]) : "v"(m), "v"(m) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[3&1]) : "v"(m), "v"(m) : "vcc");) }
        if (KIND == 11) { REP16(asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(msk2) : "v"(a[0]), "v"(b)); asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(msk2) : "v"(a[1]), "v"(b)); asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(msk2) : "v"(a[2]), "v"(b)); asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(msk2) : "v"(a[3]), "v"(b));) }
        if (KIND == 12) { REP16(asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[0]) : "v"(a[(0+1)&3]), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[1]) : "v"(a[(1+1)&3]), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[2]) : "v"(a[(2+1)&3]), "v"(b) : "vcc"); asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[3]) : "v"(a[(3+1)&3]), "v"(b) : "vcc");) }
        if (KIND == 13) { REP16(asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sr) : "v"(a[0])); asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sr) : "v"(a[1])); asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sr) : "v"(a[2])); asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sr) : "v"(a[3]));) }
        if (KIND == 14) { REP16(asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[0]) : "v"(m)); asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[1]) : "v"(m)); asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[2]) : "v"(m)); asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[3]) : "v"(m));) }
        if (KIND == 15) { REP16(asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[0])); asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[1])); asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[2])); asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[3]));) }
        if (KIND == 16) { REP16(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[1]) : "v"(b), "v"(c)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[2]) : "v"(b), "v"(c)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[3]) : "v"(b), "v"(c));) }
        if (KIND == 17) { REP16(asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[0]) : "s"(sc)); asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[1]) : "s"(sc)); asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[2]) : "s"(sc)); asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[3]) : "s"(sc));) }
        if (KIND == 18) { REP16(asm volatile("v_and_b32 %0, 1, %0" : "+v"(a[0])); asm volatile("v_and_b32 %0, 1, %0" : "+v"(a[1])); asm volatile("v_and_b32 %0, 1, %0" : "+v"(a[2])); asm volatile("v_and_b32 %0, 1, %0" : "+v"(a[3]));) }
        if (KIND == 19) { REP16(asm volatile("v_cmp_eq_u32 vcc, 0, %0" :: "v"(a[0]) : "vcc"); asm volatile("v_cmp_eq_u32 vcc, 0, %0" :: "v"(a[1]) : "vcc"); asm volatile("v_cmp_eq_u32 vcc, 0, %0" :: "v"(a[2]) : "vcc"); asm volatile("v_cmp_eq_u32 vcc, 0, %0" :: "v"(a[3]) : "vcc");) }

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
    for (int w : { 1, 4, 8 }) {
        run<0>("v_fma_f32", w);
        run<1>("v_cndmask_vcc", w);
        run<2>("v_cndmask_sgpr", w);
        run<3>("v_cndmask_const", w);
        run<4>("v_mov_b32", w);
        run<5>("v_max_f32", w);
        run<6>("v_rcp_f32", w);
        run<7>("v_lshrrev_b32", w);
        run<8>("v_lshl_add_u32", w);
        run<9>("v_lshl_add_u64", w);
        run<10>("v_mad_u64_u32", w);
        run<11>("v_cmp_f32_sgpr", w);
        run<12>("v_cmp_cnd_pair", w);
        run<13>("v_readfirstlane", w);
        run<14>("ds_bpermute", w);
        run<15>("v_add_dpp_ror8", w);
        run<16>("v_fmac_f32", w);
        run<17>("v_mul_sgpr", w);
        run<18>("v_and_or", w);
        run<19>("v_cmp_eq_u32", w);

    }
    return 0;
}
