// tools/ubench/malloc_shim.cpp -- EXPERIMENT TOOL (LD_PRELOAD), not part of the product: interposes hipMalloc for the allocations libsdhip.so makes (the caller's
// return address lies in a file whose name contains "libsdhip"), to find out what k_mm's two launch-time modes (13.0 / 14.9 ms on the same virtual
// addresses, constant on recycled blocks) have to do with WHERE in device memory a handle's large buffers land. Everything else (PyTorch's allocator) passes through.
//   SHIM_MODE=0  pass through, log the engine's large allocations
//   SHIM_MODE=1  one spacer of SHIM_MB MiB (default 1024) in front of the engine's first large allocation, kept
//   SHIM_MODE=2  every large engine allocation twice, the FIRST one freed again: the block handed out is the second one
//   SHIM_MODE=3  as 2, the first one kept (never freed)
// build: g++ -O2 -fPIC -shared -o tools/ubench/libmalloc_shim.so tools/ubench/malloc_shim.cpp -ldl
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstddef>

typedef int (*malloc_fn)(void **, size_t);
typedef int (*free_fn)(void *);

// the HIP runtime the process has loaded (PyTorch brings its own copy, in a local scope RTLD_NEXT does not reach): found by name among the loaded objects
#include <link.h>
static int find_hip(struct dl_phdr_info *i, size_t, void *out)
{
    if (i->dlpi_name && strstr(i->dlpi_name, "libamdhip64"))
    {
        *(void **)out = dlopen(i->dlpi_name, RTLD_LAZY | RTLD_NOLOAD);
        return 1;
    }
    return 0;
}
static void *hip_rt()
{
    static void *h = nullptr;
    if (!h)
        dl_iterate_phdr(find_hip, &h);
    if (!h)
    {
        fprintf(stderr, "[shim] no libamdhip64 among the loaded objects\n");
        abort();
    }
    return h;
}
static malloc_fn real_malloc()
{
    static malloc_fn f = (malloc_fn)dlsym(hip_rt(), "hipMalloc");
    return f;
}
static free_fn real_free()
{
    static free_fn f = (free_fn)dlsym(hip_rt(), "hipFree");
    return f;
}
static int mode()
{
    static int m = getenv("SHIM_MODE") ? atoi(getenv("SHIM_MODE")) : 0;
    return m;
}

extern "C" int hipMalloc(void **p, size_t n)
{
    const size_t LARGE = (size_t)256 << 20;
    Dl_info info;
    const bool engine = dladdr(__builtin_return_address(0), &info) && info.dli_fname && strstr(info.dli_fname, "libsdhip");
    if (!engine || n < LARGE || mode() == 0)
    {
        const int rc = real_malloc()(p, n);
        if (engine && n >= LARGE)
            fprintf(stderr, "[shim] engine alloc %zu MiB -> %p\n", n >> 20, *p);
        return rc;
    }
    if (mode() == 1)
    {
        static bool done = false;
        if (!done)
        {
            done = true;
            void *sp = nullptr;
            const size_t mb = getenv("SHIM_MB") ? (size_t)atol(getenv("SHIM_MB")) : 1024;
            const int rc = real_malloc()(&sp, mb << 20);
            fprintf(stderr, "[shim] spacer %zu MiB -> %p (rc %d)\n", mb, sp, rc);
        }
        const int rc = real_malloc()(p, n);
        fprintf(stderr, "[shim] engine alloc %zu MiB -> %p\n", n >> 20, *p);
        return rc;
    }
    void *first = nullptr;
    int rc = real_malloc()(&first, n);
    if (rc != 0)
        return rc;
    rc = real_malloc()(p, n);
    if (rc != 0)
    { // no room for the second block: hand out the first
        *p = first;
        fprintf(stderr, "[shim] engine alloc %zu MiB -> %p (second block refused)\n", n >> 20, *p);
        return 0;
    }
    if (mode() == 2)
        real_free()(first);
    fprintf(stderr, "[shim] engine alloc %zu MiB: first %p %s, handed out %p\n", n >> 20, first, mode() == 2 ? "freed" : "kept", *p);
    return 0;
}
