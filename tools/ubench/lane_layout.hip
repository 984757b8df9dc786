// Micro-benchmark: what does the lane-per-chunk access pattern of the speculative stages cost, and what would a lane-transposed
// layout buy? One lane per chunk, every lane streams its chunk in 64-byte blocks (4 x float4) through a double-buffered register
// queue, does a token dependent recurrence, and stores the block.
//   natural:    sample i of the stream at x[i]              -> a wave's 64 lanes touch 64 addresses L*8 bytes apart
//   transposed: block j of chunk k at ((j*K) + k) * 64 bytes -> a wave's 64 lanes touch 4 KB of consecutive memory
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lane_layout.hip -o /tmp/lane_layout ; run: /tmp/lane_layout [K] [L]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Blk { float4 a, b, c, d; };
template <int MODE_IN, int MODE_OUT, int D>
__global__ __launch_bounds__(64) void k_lane(const float4 *x, float4 *y, int K, long long L /* samples per chunk, multiple of 8*D */, float g0)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= K)
        return;
    const long long nb = L / 8; // blocks per chunk
    auto addr_in = [&](long long j) -> const float4 * { return MODE_IN == 0 ? x + ((long long)k * nb + j) * 4 : x + (j * K + k) * 4; };
    auto addr_out = [&](long long j) -> float4 * { return MODE_OUT == 0 ? y + ((long long)k * nb + j) * 4 : y + (j * K + k) * 4; };
    float g = g0;
    Blk qa[D], qb[D];
    for (int d = 0; d < D; d++) { const float4 *p = addr_in(d); qa[d] = Blk{p[0], p[1], p[2], p[3]}; }
    for (long long j = 0; j < nb; j += 2 * D)
    {
        for (int d = 0; d < D; d++) { const float4 *p = addr_in(j + D + d); qb[d] = Blk{p[0], p[1], p[2], p[3]}; }
        for (int d = 0; d < D; d++)
        {
            Blk &q = qa[d];
            float *f = reinterpret_cast<float *>(&q);
            for (int i = 0; i < 16; i += 2) { f[i] *= g; f[i + 1] *= g; g = g + 1e-6f * (1.0f - (f[i] * f[i] + f[i + 1] * f[i + 1])); }
            float4 *o = addr_out(j + d); o[0] = q.a; o[1] = q.b; o[2] = q.c; o[3] = q.d;
        }
        if (j + 2 * D < nb)
            for (int d = 0; d < D; d++) { const float4 *p = addr_in(j + 2 * D + d); qa[d] = Blk{p[0], p[1], p[2], p[3]}; }
        for (int d = 0; d < D; d++)
        {
            Blk &q = qb[d];
            float *f = reinterpret_cast<float *>(&q);
            for (int i = 0; i < 16; i += 2) { f[i] *= g; f[i + 1] *= g; g = g + 1e-6f * (1.0f - (f[i] * f[i] + f[i + 1] * f[i + 1])); }
            float4 *o = addr_out(j + D + d); o[0] = q.a; o[1] = q.b; o[2] = q.c; o[3] = q.d;
        }
    }
}
template <int MI, int MO, int D>
static void run(const char *name, const float4 *x, float4 *y, int K, long long L)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int it = 0; it < 2; it++)
    {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_lane<MI, MO, D>), dim3((K + 63) / 64), dim3(64), 0, 0, x, y, K, L, 1.0f);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    }
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = 16.0 * K * L;
    printf("%-34s K %d L %lld depth %d: %7.3f ms  %6.2f TB/s (read + write)\n", name, K, L, D, ms, bytes / ms * 1e-9);
}
int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 65280;
    const long long L = argc > 2 ? atoll(argv[2]) : 32768;
    const size_t bytes = (size_t)K * L * 8;
    float4 *x, *y;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes));
    CK(hipMemset(x, 0, bytes)); CK(hipMemset(y, 0, bytes));
    run<0, 0, 4>("natural -> natural", x, y, K, L);
    run<0, 0, 2>("natural -> natural", x, y, K, L);
    run<0, 1, 4>("natural -> transposed", x, y, K, L);
    run<1, 1, 4>("transposed -> transposed", x, y, K, L);
    run<1, 1, 2>("transposed -> transposed", x, y, K, L);
    run<1, 1, 1>("transposed -> transposed", x, y, K, L);
    run<1, 0, 4>("transposed -> natural", x, y, K, L);
    return 0;
}
