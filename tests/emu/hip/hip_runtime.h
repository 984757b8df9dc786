// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY. A minimal stand-in for <hip/hip_runtime.h> that lets the
// demodulator's HIP sources (satdump_amd/csrc/demod_{kernels,engine}.hip, unchanged) be compiled for the HOST, so that the
// engine's host logic (chunk speculation, boundary certificates, hand-off, re-run rounds) can be exercised by the CPU test
// suite in a container that has no GPU. Kernels run one block at a time, the threads of a block as cooperative fibers that
// switch at __syncthreads(). Nothing here is ever loaded by the product: satdump_amd/capi.py only loads lib/libsdhip.so; the
// twin is built into tests/emu/_build by tests/emu/build.py and opened explicitly by tests/test_demod_emu_cpu.py.
// Not a performance model, not a fallback: timing, occupancy and wave-level behaviour do not exist here.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3
{
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_idx
{
    unsigned x, y, z;
};
extern emu_idx threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

// HIP's small vector types (plain structs are enough for the .x/.y/.z/.w member access the sources use)
struct short2 { short x, y; };
struct char2 { signed char x, y; };
struct uchar2 { unsigned char x, y; };
struct int2 { int x, y; };
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
struct double2 { double x, y; };
inline double2 make_double2(double x, double y) { return double2{x, y}; }

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
typedef void *hipStream_t;
enum hipMemcpyKind
{
    hipMemcpyHostToHost,
    hipMemcpyHostToDevice,
    hipMemcpyDeviceToHost,
    hipMemcpyDeviceToDevice,
    hipMemcpyDefault
};
enum
{
    hipHostMallocDefault = 0,
    hipDeviceAttributeMultiprocessorCount = 1
};

inline const char *hipGetErrorString(hipError_t) { return "host twin"; }
inline hipError_t hipMalloc(void **p, size_t n)
{ // device memory is uninitialised: poison it so that a kernel relying on zeros shows up
    *p = malloc(n ? n : 1);
    if (*p)
        memset(*p, getenv("EMU_POISON_ZERO") ? 0 : 0xA5, n);
    return *p ? 0 : 2;
}
inline hipError_t hipFree(void *p)
{
    free(p);
    return 0;
}
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned)
{
    *p = malloc(n ? n : 1);
    return *p ? 0 : 2;
}
inline hipError_t hipHostFree(void *p)
{
    free(p);
    return 0;
}
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind)
{
    memmove(d, s, n);
    return 0;
}
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr)
{
    memmove(d, s, n);
    return 0;
}
inline hipError_t hipMemset(void *d, int v, size_t n)
{
    memset(d, v, n);
    return 0;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr)
{
    memset(d, v, n);
    return 0;
}
inline hipError_t hipStreamCreate(hipStream_t *s)
{
    *s = nullptr;
    return 0;
}
constexpr unsigned hipStreamNonBlocking = 1;
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDevice(int *d)
{
    *d = 0;
    return 0;
}
inline hipError_t hipDeviceGetAttribute(int *v, int, int)
{
    *v = 8; // "CUs": keeps the persistent kernels' grids small
    return 0;
}
template <class K>
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *occ, K, int, size_t)
{
    *occ = 2;
    return 0;
}

// events (the library's profiling scope): no clock here
typedef void *hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t *e)
{
    *e = nullptr;
    return 0;
}
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t)
{
    *ms = 0.0f;
    return 0;
}
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipGetDeviceCount(int *n)
{
    *n = 1;
    return 0;
}

// ---- wave-level operations (the FEC kernels): the 64 fibers of a wave meet at every collective (emu_runtime.cpp). All live
// lanes of the wave must call the same collectives in the same order (true for convergent code, which is what the kernels have).
unsigned long long emu_wave_xchg(unsigned long long v, int src_lane); // value lane src_lane passed to the same call
unsigned long long emu_ballot(int pred);
inline int emu_lane() { return (int)((threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)) & 63u); }
inline unsigned long long __ballot(int pred) { return emu_ballot(pred); }
inline int __shfl_xor(int v, int mask) { return (int)(unsigned)emu_wave_xchg((unsigned)v, emu_lane() ^ mask); }
inline double __shfl_xor(double v, int mask)
{
    unsigned long long u;
    memcpy(&u, &v, 8);
    u = emu_wave_xchg(u, emu_lane() ^ mask);
    memcpy(&v, &u, 8);
    return v;
}
inline int __shfl_down(int v, int d) { return (int)(unsigned)emu_wave_xchg((unsigned)v, emu_lane() + d < 64 ? emu_lane() + d : emu_lane()); }
inline int __shfl(int v, int lane) { return (int)(unsigned)emu_wave_xchg((unsigned)v, lane & 63); }
inline int emu_readlane(int v, int lane) { return (int)(unsigned)emu_wave_xchg((unsigned)v, lane & 63); }
inline int emu_update_dpp(int, int src, int ctrl, int, int, bool)
{ // the four controls common.h uses: all of them are permutations inside a row of 16 lanes
    const int l = emu_lane();
    int from = l;
    if (ctrl == 0xB1)
        from = l ^ 1; // quad_perm [1,0,3,2]
    else if (ctrl == 0x4E)
        from = l ^ 2; // quad_perm [2,3,0,1]
    else if (ctrl == 0x141)
        from = (l & ~7) | (7 - (l & 7)); // row_half_mirror
    else if (ctrl == 0x140)
        from = (l & ~15) | (15 - (l & 15)); // row_mirror
    else
        abort();
    return (int)(unsigned)emu_wave_xchg((unsigned)src, from);
}
inline unsigned emu_perm(unsigned s0, unsigned s1, unsigned sel)
{ // v_perm_b32: selector 0-3 = bytes of s1, 4-7 = bytes of s0, 12 = 0x00, >= 13 = 0xff
    const unsigned long long c = ((unsigned long long)s0 << 32) | s1;
    unsigned r = 0;
    for (int i = 0; i < 4; i++)
    {
        const unsigned k = (sel >> (8 * i)) & 0xff;
        const unsigned b = k < 8 ? (unsigned)((c >> (8 * k)) & 0xff) : (k == 12 ? 0u : (k >= 13 ? 0xffu : 0u));
        r |= b << (8 * i);
    }
    return r;
}
#define __builtin_amdgcn_readlane emu_readlane
#define __builtin_amdgcn_update_dpp emu_update_dpp
#define __builtin_amdgcn_perm emu_perm
#define __builtin_amdgcn_wave_barrier() ((void)emu_ballot(1)) // the wave's lanes meet: what lockstep execution gives the hardware for free (one lane's LDS write before another lane's next read)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
inline void __threadfence() {}
inline void __threadfence_block() {}
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline unsigned __brev(unsigned v)
{
    unsigned r = 0;
    for (int i = 0; i < 32; i++)
        r |= ((v >> i) & 1u) << (31 - i);
    return r;
}

// ---- kernel launch: blocks one after the other, the threads of a block as fibers (emu_runtime.cpp)
void emu_launch(dim3 grid, dim3 block, const std::function<void()> &thread_body);
void __syncthreads();
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })

// ---- device-side helpers the demodulator sources use
inline unsigned __float_as_uint(float f)
{
    unsigned u;
    memcpy(&u, &f, 4);
    return u;
}
inline float __uint_as_float(unsigned u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline void __sincosf(float x, float *s, float *c)
{ // HIP's fast intrinsic (only used for speculative warm-up estimates, never for certified output)
    *s = sinf(x);
    *c = cosf(x);
}
template <class T>
inline T atomicAdd(T *p, T v)
{
    const T o = *p;
    *p = o + v;
    return o;
}
template <class T>
inline T atomicOr(T *p, T v)
{
    const T o = *p;
    *p = o | v;
    return o;
}
int __syncthreads_or(int pred); // emu_runtime.cpp
template <class T>
inline T atomicExch(T *p, T v)
{
    const T o = *p;
    *p = v;
    return o;
}
template <class T>
inline T atomicMax(T *p, T v)
{
    const T o = *p;
    if (v > o)
        *p = v;
    return o;
}
template <class T>
inline T atomicMin(T *p, T v)
{
    const T o = *p;
    if (v < o)
        *p = v;
    return o;
}
using std::max;
using std::min;
