// Checks the semantics the cluster-pair kernel relies on for LDS-direct loads on gfx950:
//   global_load_lds_dwordx4 / _dword  voffset, saddr   with M0 = LDS byte address:
//   active lane l writes its 16 (4) bytes to M0 + 16 (4) * l; inactive lanes write nothing; vmcnt counts it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const float4* __restrict__ src, const int* __restrict__ words, const int* __restrict__ idx, float* __restrict__ out)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const int      lane    = threadIdx.x & 63;
    const int      wave    = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned ldsBase = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds)) + 2048u * wave;
    for (int i = lane; i < 512; i += 64) { reinterpret_cast<float*>(lds + 2048 * wave)[i] = -1.0f; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int a = idx[64 * wave + lane];
    if (lane < 32)
    {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsBase), "v"(a * 16), "s"(src) : "memory");
    }
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(ldsBase + 1024u), "v"(a * 4), "s"(words) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float* l = reinterpret_cast<const float*>(lds + 2048 * wave);
    for (int i = lane; i < 512; i += 64) { out[512 * wave + i] = l[i]; }
}
int main()
{
    const int n = 1000, nw = 4;
    std::vector<float4> src(n);
    std::vector<int> words(n), idx(64 * nw);
    for (int i = 0; i < n; i++) { src[i] = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f); words[i] = 7 * i + 3; }
    for (int i = 0; i < 64 * nw; i++) { idx[i] = (i * 37 + 11) % n; }
    float4* dsrc; int *dw, *didx; float* dout;
    hipMalloc(&dsrc, n * 16); hipMalloc(&dw, n * 4); hipMalloc(&didx, idx.size() * 4); hipMalloc(&dout, 512 * nw * 4);
    hipMemcpy(dsrc, src.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(dw, words.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(didx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64 * nw), 2048 * nw, 0, dsrc, dw, didx, dout);
    std::vector<float> out(512 * nw);
    if (hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL: runtime error\n"); return 1; }
    int bad = 0;
    for (int w = 0; w < nw; w++)
    {
        const float* o = &out[512 * w];
        for (int l = 0; l < 32; l++)
        {
            const int a = idx[64 * w + l];
            if (o[4 * l] != src[a].x || o[4 * l + 1] != src[a].y || o[4 * l + 2] != src[a].z || o[4 * l + 3] != src[a].w) { if (bad++ < 5) printf("xq mismatch w%d l%d got %g want %g\n", w, l, o[4 * l], src[a].x); }
        }
        for (int i = 128; i < 256; i++) { if (o[i] != -1.0f) { if (bad++ < 5) printf("inactive lanes wrote LDS w%d i%d %g\n", w, i, o[i]); } }
        for (int l = 0; l < 64; l++)
        {
            const int a = idx[64 * w + l];
            int got; std::memcpy(&got, &o[256 + l], 4);
            if (got != words[a]) { if (bad++ < 5) printf("word mismatch w%d l%d got %d want %d\n", w, l, got, words[a]); }
        }
    }
    printf("LDS-direct load semantics: %s\n", bad ? "MISMATCH" : "OK");
    return bad != 0;
}
