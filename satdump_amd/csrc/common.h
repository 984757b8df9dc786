// common.h -- shared host/device helpers for libsdhip (gfx950 only, no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>

namespace sdhip
{
    void set_error(const std::string &msg);

    struct HipError : std::runtime_error
    {
        using std::runtime_error::runtime_error;
    };

#define SD_HIP(expr)                                                                                                  \
    do                                                                                                                \
    {                                                                                                                 \
        hipError_t _e = (expr);                                                                                       \
        if (_e != hipSuccess)                                                                                         \
            throw sdhip::HipError(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + \
                                  std::to_string(__LINE__));                                                          \
    } while (0)

    // Optional per-kernel timing (HIP events recorded on the launch stream, around every launch). Off by default;
    // bench.py turns it on through sdhip_prof_enable() to get the roofline figures of the dominant kernel.
    struct ProfScope
    {
        int idx;
        hipStream_t st;
        ProfScope(const char *name, hipStream_t stream);
        ~ProfScope();
    };

    // Device / pinned-host memory of the buffers below. With the pool off (default) these are hipMalloc / hipFree and
    // hipHostMalloc / hipHostFree. With it on (sdhip_pool_enable), freed blocks are parked per (device, size) and handed to
    // the next request of the same size: a caller that destroys and re-creates handles for every recording (one cold start
    // per chunk, bench.py --gpus N) then pays no allocation after the first one. sdhip_pool_trim() releases what is parked.
    void *dev_alloc(size_t bytes);
    void dev_free(void *p, size_t bytes);
    void *pin_alloc(size_t bytes);
    void pin_free(void *p, size_t bytes);

    // Simple owning device buffer (grow-only).
    template <class T>
    struct DevBuf
    {
        T *p = nullptr;
        size_t cap = 0;
        ~DevBuf() { release(); }
        void release()
        {
            if (p)
                dev_free(p, cap * sizeof(T));
            p = nullptr;
            cap = 0;
        }
        void reserve(size_t n)
        {
            if (n <= cap)
                return;
            release();
            size_t want = n + n / 8 + 64;
            p = (T *)dev_alloc(want * sizeof(T));
            cap = want;
        }
        void swap(DevBuf &o)
        {
            std::swap(p, o.p);
            std::swap(cap, o.cap);
        }
        DevBuf() = default;
        DevBuf(const DevBuf &) = delete;
        DevBuf &operator=(const DevBuf &) = delete;
    };

    // Pinned host buffer (grow-only).
    template <class T>
    struct PinBuf
    {
        T *p = nullptr;
        size_t cap = 0;
        ~PinBuf()
        {
            if (p)
                pin_free(p, cap * sizeof(T));
        }
        void reserve(size_t n)
        {
            if (n <= cap)
                return;
            if (p)
                pin_free(p, cap * sizeof(T));
            p = nullptr;
            cap = 0;
            size_t want = n + n / 8 + 64;
            p = (T *)pin_alloc(want * sizeof(T));
            cap = want;
        }
        void swap(PinBuf &o)
        {
            std::swap(p, o.p);
            std::swap(cap, o.cap);
        }
        PinBuf() = default;
        PinBuf(const PinBuf &) = delete;
        PinBuf &operator=(const PinBuf &) = delete;
    };

    // ---- device-side wave helpers (wave64) ---------------------------------------------------
#ifdef __HIPCC__
    __device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

    __device__ __forceinline__ unsigned rotl6(unsigned x, unsigned n)
    {
        n %= 6;
        return ((x << n) | (x >> (6 - n))) & 63u;
    }
    __device__ __forceinline__ unsigned rotr6(unsigned x, unsigned n)
    {
        n %= 6;
        return ((x >> n) | (x << (6 - n))) & 63u;
    }
    __device__ __forceinline__ unsigned parity32(unsigned x) { return (unsigned)__popc(x) & 1u; }

    // DPP move: result[lane] = src[perm(lane)] (all rows/banks enabled, out-of-range lanes keep `src`).
    template <int CTRL>
    __device__ __forceinline__ unsigned dpp_mov(unsigned src)
    {
        return (unsigned)__builtin_amdgcn_update_dpp((int)src, (int)src, CTRL, 0xF, 0xF, false);
    }

    // min over the 64 lanes of a wave; every lane returns the same (wave-uniform) value.
    __device__ __forceinline__ unsigned wave_min_u32(unsigned v)
    {
        v = min(v, dpp_mov<0xB1>(v));  // quad_perm [1,0,3,2]  (xor 1)
        v = min(v, dpp_mov<0x4E>(v));  // quad_perm [2,3,0,1]  (xor 2)
        v = min(v, dpp_mov<0x141>(v)); // row_half_mirror      (joins the two quads of each 8)
        v = min(v, dpp_mov<0x140>(v)); // row_mirror           (joins the two halves of each 16)
        unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
        unsigned b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
        unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32);
        unsigned d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
        return min(min(a, b), min(c, d));
    }
    __device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
    {
        v += dpp_mov<0xB1>(v);
        v += dpp_mov<0x4E>(v);
        v += dpp_mov<0x141>(v);
        v += dpp_mov<0x140>(v);
        unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
        unsigned b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
        unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32);
        unsigned d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
        return a + b + c + d;
    }
#endif
} // namespace sdhip
