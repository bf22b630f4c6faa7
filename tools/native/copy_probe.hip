// Which engine does a large device -> page-locked host copy take?  (rocprofv3 --kernel-trace --memory-copy-trace:
// an SDMA copy shows in the memory-copy trace, a blit copy as the kernel __amd_rocclr_copyBuffer.)
//   hipcc --offload-arch=gfx950 -O2 -o copy_probe copy_probe.hip && rocprofv3 ... -- ./copy_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void touch(unsigned long long* p, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : size_t(1) << 30, n = bytes / 8;
    unsigned long long *d, *h;
    CK(hipMalloc(&d, bytes));
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    auto timed = [&](const char* what, auto&& fn) {
        hipDeviceSynchronize();
        const auto t0 = std::chrono::steady_clock::now();
        fn();
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("%-70s %7.2f ms  %5.1f GB/s\n", what, ms, bytes / ms / 1e6);
    };
    touch<<<(unsigned)((n + 255) / 256), 256, 0, a>>>(d, n);
    timed("warm-up copy", [&] { hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, a); });
    timed("1: copy on an idle stream", [&] { hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, a); });
    timed("2: kernel, then copy on the same stream", [&] {
        touch<<<(unsigned)((n + 255) / 256), 256, 0, a>>>(d, n);
        hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, a);
    });
    timed("3: kernel on a, event, copy on b behind the event", [&] {
        touch<<<(unsigned)((n + 255) / 256), 256, 0, a>>>(d, n);
        hipEventRecord(ev, a);
        hipStreamWaitEvent(b, ev, 0);
        hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, b);
    });
    timed("4: like 3, in 16 chunks of 64 MB", [&] {
        touch<<<(unsigned)((n + 255) / 256), 256, 0, a>>>(d, n);
        hipEventRecord(ev, a);
        hipStreamWaitEvent(b, ev, 0);
        for (int k = 0; k < 16; ++k)
            hipMemcpyAsync((char*)h + (size_t)k * (bytes / 16), (char*)d + (size_t)k * (bytes / 16), bytes / 16, hipMemcpyDeviceToHost, b);
    });
    timed("5: hipMemcpyDtoHAsync on an idle stream", [&] { hipMemcpyDtoHAsync(h, (hipDeviceptr_t)d, bytes, b); });
    // do kernels on another stream slow down while the copy runs?
    unsigned long long* d2;
    CK(hipMalloc(&d2, bytes));
    hipEvent_t k0, k1;
    CK(hipEventCreate(&k0));
    CK(hipEventCreate(&k1));
    auto kernels = [&](const char* what, bool with_copy) {
        hipDeviceSynchronize();
        if (with_copy) hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, b);
        hipEventRecord(k0, a);
        for (int r = 0; r < 20; ++r) touch<<<(unsigned)((n + 255) / 256), 256, 0, a>>>(d2, n);
        hipEventRecord(k1, a);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, k0, k1);
        printf("%-70s %7.2f ms for 20 kernels\n", what, ms);
    };
    kernels("6: 20 fill kernels alone", false);
    kernels("7: 20 fill kernels while a copy to the host runs on another stream", true);
    kernels("6 again", false);
    kernels("7 again", true);
    return 0;
}
