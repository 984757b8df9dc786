// tools/ubench/host_copy.hip -- what the host path of sdhip_demod_push can hope for on a box: pageable -> pinned staging copy with 1..32 threads,
// pinned -> device and pageable -> device hipMemcpy, device -> pinned. Build + run: hipcc -O2 -o /tmp/host_copy tools/ubench/host_copy.hip -lpthread && /tmp/host_copy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = (size_t)2 << 30;
    char *page = (char *)malloc(n), *pin = nullptr, *dev = nullptr;
    memset(page, 1, n);
    double t0 = now();
    if (hipHostMalloc((void **)&pin, n, hipHostMallocDefault) != hipSuccess || hipMalloc((void **)&dev, n) != hipSuccess)
        return 1;
    printf("hipHostMalloc + hipMalloc of %zu MB: %.1f ms\n", n >> 20, (now() - t0) * 1e3);
    memset(pin, 2, n);
    for (int nt : {1, 2, 4, 8, 16, 32, 64})
    {
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++)
        {
            t0 = now();
            std::vector<std::thread> th;
            const size_t per = n / nt;
            for (int i = 0; i < nt; i++)
                th.emplace_back([=] { memcpy(pin + per * i, page + per * i, per); });
            for (auto &t : th)
                t.join();
            best = std::min(best, now() - t0);
        }
        printf("pageable -> pinned, %2d threads: %.1f GB/s\n", nt, n / best / 1e9);
    }
    for (int k = 0; k < 3; k++)
    {
        const char *name[3] = {"pinned -> device", "pageable -> device", "device -> pinned"};
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++)
        {
            t0 = now();
            if (k == 0)
                (void)hipMemcpy(dev, pin, n, hipMemcpyHostToDevice);
            else if (k == 1)
                (void)hipMemcpy(dev, page, n, hipMemcpyHostToDevice);
            else
                (void)hipMemcpy(pin, dev, n, hipMemcpyDeviceToHost);
            (void)hipDeviceSynchronize();
            best = std::min(best, now() - t0);
        }
        printf("%s: %.1f GB/s\n", name[k], n / best / 1e9);
    }
    return 0;
}
