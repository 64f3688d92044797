// how long does hipMalloc take right after another process has freed tens of GB? (the "burst" of the stage wall: DESIGN.md section 5)
//   alloc_probe dirty GB      : allocate GB gigabytes, touch them, exit (leaves freshly freed memory behind)
//   alloc_probe probe N MB    : time N allocations of MB megabytes one after the other, then free them
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
int main(int argc, char **argv)
{
    using C = std::chrono::steady_clock;
    if (argc < 3) return 1;
    if (!strcmp(argv[1], "dirty")) {
        const size_t gb = (size_t)atoll(argv[2]);
        void *p = nullptr;
        if (hipMalloc(&p, gb << 30) != hipSuccess) return 2;
        hipMemset(p, 1, gb << 30);
        hipDeviceSynchronize();
        return 0; // no hipFree: the process end frees it
    }
    const int n = atoi(argv[2]);
    const size_t mb = (size_t)atoll(argv[3]);
    auto t00 = C::now();
    hipFree(nullptr); // runtime init
    printf("init %.3f s\n", std::chrono::duration<double>(C::now() - t00).count());
    std::vector<void *> ps;
    auto t0 = C::now();
    for (int i = 0; i < n; i++) {
        auto t = C::now();
        void *p = nullptr;
        if (hipMalloc(&p, mb << 20) != hipSuccess) { printf("alloc %d failed\n", i); break; }
        ps.push_back(p);
        printf("alloc %2d of %zu MB: %.3f s (cumulative %.3f)\n", i, mb, std::chrono::duration<double>(C::now() - t).count(), std::chrono::duration<double>(C::now() - t0).count());
    }
    return 0;
}
