// Can a file's page-cache pages be registered for DMA and copied to the device without a CPU copy?  (feasibility probe)
// build: hipcc -O2 tools/native/map_probe.cpp -o tools/native/map_probe ; run: tools/native/map_probe <file> [chunk MB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* path = argv[1];
    const size_t chunk = (argc > 2 ? atol(argv[2]) : 48) << 20;
    int fd = open(path, O_RDONLY);
    struct stat sb;
    fstat(fd, &sb);
    const size_t n = (size_t)sb.st_size & ~((size_t)(2 << 20) - 1);
    // warm the page cache
    { std::vector<char> b(8 << 20); size_t o = 0; while (o < n) { ssize_t r = pread(fd, b.data(), b.size(), o); if (r <= 0) break; o += r; } }
    void* d = nullptr;
    hipMalloc(&d, chunk * 2);
    hipStream_t s;
    hipStreamCreate(&s);
    for (int mode = 0; mode < 3; ++mode) {
        // 0: MAP_SHARED file mapping registered whole; 1: MAP_PRIVATE|MAP_POPULATE registered; 2: pread into a pinned buffer (today)
        double t0 = now();
        void* m = nullptr;
        hipError_t e = hipSuccess;
        if (mode < 2) {
            m = mmap(nullptr, n, PROT_READ, (mode == 0 ? MAP_SHARED : MAP_PRIVATE) | MAP_POPULATE, fd, 0);
            if (m == MAP_FAILED) { printf("mode %d: mmap failed\n", mode); continue; }
            const double t1 = now();
            e = hipHostRegister(m, n, hipHostRegisterDefault | hipHostRegisterReadOnly);
            if (e != hipSuccess) { (void)hipGetLastError(); e = hipHostRegister(m, n, hipHostRegisterDefault); }
            const double t2 = now();
            printf("mode %d: mmap+populate %.1f ms, hipHostRegister %.1f ms (%s) for %.2f GB\n", mode, (t1 - t0) * 1e3, (t2 - t1) * 1e3, hipGetErrorName(e), n / 1e9);
            if (e != hipSuccess) { (void)hipGetLastError(); munmap(m, n); continue; }
        } else {
            hipHostMalloc(&m, chunk, 0);
        }
        for (int rep = 0; rep < 2; ++rep) {
            const double c0 = now();
            for (size_t o = 0; o + chunk <= n; o += chunk) {
                if (mode < 2) hipMemcpyAsync((char*)d + ((o / chunk) & 1) * chunk, (char*)m + o, chunk, hipMemcpyHostToDevice, s);
                else {
                    size_t a = 0;
                    while (a < chunk) { ssize_t r = pread(fd, (char*)m + a, chunk - a, o + a); if (r <= 0) break; a += r; }
                    hipMemcpyAsync((char*)d + ((o / chunk) & 1) * chunk, m, chunk, hipMemcpyHostToDevice, s);
                    hipStreamSynchronize(s);
                }
            }
            hipStreamSynchronize(s);
            const double c1 = now();
            printf("  mode %d pass %d: %.2f GB to the device in %.1f ms = %.1f GB/s\n", mode, rep, n / 1e9, (c1 - c0) * 1e3, n / 1e9 / (c1 - c0));
        }
        const double u0 = now();
        if (mode < 2) { hipHostUnregister(m); munmap(m, n); } else hipHostFree(m);
        printf("  mode %d: released in %.1f ms\n", mode, (now() - u0) * 1e3);
    }
    return 0;
}
