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
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }

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
using std::max;
using std::min;
