// Micro-benchmark: host time of one kernel launch call by kernel-argument size and dynamic LDS size (queue kept short: sync every 64 launches)
// Build: hipcc -O3 --offload-arch=gfx950 launch_cost.hip -o launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template<int N> struct Blob { int v[N]; };
template<int N> __global__ void k(Blob<N> b, int* out) { if (b.v[0] == 12345 && threadIdx.x == 0) { out[0] = b.v[N - 1]; } }
template<int N> double run(int* out, int lds, int blocks, hipStream_t s)
{
    Blob<N> b{};
    for (int i = 0; i < 64; i++) { hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), lds, s, b, out); }
    (void)hipStreamSynchronize(s);
    double us = 0; int n = 0;
    for (int rep = 0; rep < 50; rep++)
    {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 64; i++) { hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), lds, s, b, out); }
        us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        n += 64;
        (void)hipStreamSynchronize(s);
    }
    return us / n;
}
int main()
{
    int* out; CHECK(hipMalloc(&out, 64));
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int lds : { 0, 30000 })
        for (int blocks : { 64, 2048 })
        {
            printf("LDS %5d B, %4d blocks: 16 B args %.2f us, 256 B %.2f us, 1 KB %.2f us, 3.9 KB %.2f us per launch call\n", lds, blocks,
                   run<2>(out, lds, blocks, s), run<62>(out, lds, blocks, s), run<254>(out, lds, blocks, s), run<990>(out, lds, blocks, s));
        }
    return 0;
}
