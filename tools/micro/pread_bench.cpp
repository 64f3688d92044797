// how fast do T threads pread a page-cache-resident file into (a) plain memory, (b) hipHostMalloc memory, (c) malloc + hipHostRegister?
// (the input stage's reader, disco_hip.hip ingest_read_file, is bound by this copy under the box's 16-CPU quota)
//   hipcc -O2 -o /tmp/pread_bench tools/micro/pread_bench.cpp -lpthread ;  /tmp/pread_bench FILE [THREADS=16]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double run(int fd, char *buf, size_t chunk, size_t total, int T)
{
    const double t0 = now();
    for (size_t off = 0; off < total; off += chunk) {
        const size_t len = std::min(chunk, total - off);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([=]() {
                size_t p0 = len * t / T, p1 = len * (t + 1) / T;
                while (p0 < p1) {
                    ssize_t g = pread(fd, buf + p0, p1 - p0, (off_t)(off + p0));
                    if (g <= 0) return;
                    p0 += (size_t)g;
                }
            });
        for (auto &x : th) x.join();
    }
    return now() - t0;
}
int main(int argc, char **argv)
{
    const char *path = argv[1];
    const int T = argc > 2 ? atoi(argv[2]) : 16;
    int fd = open(path, O_RDONLY);
    struct stat st;
    fstat(fd, &st);
    const size_t total = (size_t)st.st_size, chunk = 128u << 20;
    hipSetDevice(0);
    char *a = (char *)aligned_alloc(4096, chunk), *b = nullptr, *c = (char *)aligned_alloc(4096, chunk), *d = nullptr;
    for (size_t i = 0; i < chunk; i += 4096) a[i] = c[i] = 1;
    hipHostMalloc((void **)&b, chunk);
    hipHostMalloc((void **)&d, chunk, hipHostMallocNonCoherent);
    hipHostRegister(c, chunk, hipHostRegisterDefault);
    for (int rep = 0; rep < 2; rep++) {
        printf("plain %.3f s | hipHostMalloc %.3f s | hipHostMalloc(NonCoherent) %.3f s | malloc+hipHostRegister %.3f s  (%.1f GB, %d threads)\n", run(fd, a, chunk, total, T),
               run(fd, b, chunk, total, T), run(fd, d, chunk, total, T), run(fd, c, chunk, total, T), total / 1e9, T);
    }
    return 0;
}
