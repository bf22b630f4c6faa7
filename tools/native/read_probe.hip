// What does a plain streaming READ of N bytes cost on this device when the bytes come from HBM (the caches evicted
// first)?  The calibration for small read-only kernels such as gc_count_kernel (61 MB per launch): a launch that short
// is a few memory latencies long, and "8 TB/s" is not what any kernel reads 61 MB at.
//   hipcc --offload-arch=gfx950 -O2 -o read_probe read_probe.hip && ./read_probe [bytes=60819299]
// Prints, for several shapes of a sum-everything kernel (16-byte loads, k loads in flight per thread, one wave per 25 KB
// like gc_count_kernel or a grid-stride loop), the median event-timed duration and the rate.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int K>
__global__ __launch_bounds__(256) void sum_stride(const uint4* __restrict__ p, size_t n16, unsigned long long* out) {
    unsigned long long acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    for (; i + (K - 1) * stride < n16; i += K * stride) {
        uint4 v[K];
#pragma unroll
        for (int u = 0; u < K; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < K; ++u) acc += __popc(v[u].x) + __popc(v[u].y) + __popc(v[u].z) + __popc(v[u].w);
    }
    for (; i < n16; i += stride) { const uint4 v = p[i]; acc += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0 && acc == 0x123456789abcdefull) *out = acc;  // (keeps the loads alive)
}

// one wave per `per` consecutive 16-byte words, K rows of 64 words in flight (gc_count_kernel's shape)
template <int K>
__global__ __launch_bounds__(256) void sum_wave_per_bin(const uint4* __restrict__ p, size_t n16, size_t per, unsigned long long* out) {
    const size_t w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t a = w * per, b = std::min(a + per, n16);
    unsigned long long acc = 0;
    size_t i = a + lane;
    for (; i + (K - 1) * 64 < b; i += K * 64) {
        uint4 v[K];
#pragma unroll
        for (int u = 0; u < K; ++u) v[u] = p[i + u * 64];
#pragma unroll
        for (int u = 0; u < K; ++u) acc += __popc(v[u].x) + __popc(v[u].y) + __popc(v[u].z) + __popc(v[u].w);
    }
    for (; i < b; i += 64) { const uint4 v = p[i]; acc += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0 && acc == 0x123456789abcdefull) *out = acc;
}

__global__ void evict(const uint4* __restrict__ p, size_t n16, unsigned long long* out) {
    unsigned long long acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) acc += p[i].x;
    if (acc == 0x123456789abcdefull) *out = acc;
}

int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? (size_t)atoll(argv[1]) : 60819299, n16 = bytes / 16;
    uint4 *d, *big;
    unsigned long long* out;
    const size_t big_bytes = size_t(640) << 20;
    CK(hipMalloc(&d, n16 * 16 + 64));
    CK(hipMalloc(&big, big_bytes));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(d, 0x5a, n16 * 16));
    CK(hipMemset(big, 1, big_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](const char* what, auto&& launch) -> int {
        std::vector<float> ms;
        for (int r = 0; r < 9; ++r) {
            evict<<<2048, 256>>>(big, big_bytes / 16, out);  // 640 MB > the 256 MB Infinity Cache
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-56s median %7.2f us  best %7.2f us  %6.2f TB/s\n", what, ms[4] * 1e3, ms[0] * 1e3, bytes / (ms[4] * 1e-3) / 1e12);
        return 0;
    };
    const size_t per = 25000 / 16;  // a 100 kb bin of a 2bit image
    const unsigned n_bins = (unsigned)((n16 + per - 1) / per);
    timed("grid-stride, 1 load in flight, 2048 blocks", [&] { sum_stride<1><<<2048, 256>>>(d, n16, out); });
    timed("grid-stride, 4 loads in flight, 2048 blocks", [&] { sum_stride<4><<<2048, 256>>>(d, n16, out); });
    timed("grid-stride, 4 loads in flight, 1024 blocks", [&] { sum_stride<4><<<1024, 256>>>(d, n16, out); });
    timed("grid-stride, 8 loads in flight, 1024 blocks", [&] { sum_stride<8><<<1024, 256>>>(d, n16, out); });
    timed("grid-stride, 8 loads in flight, 512 blocks", [&] { sum_stride<8><<<512, 256>>>(d, n16, out); });
    timed("one wave per 25 KB, 1 row in flight", [&] { sum_wave_per_bin<1><<<(n_bins + 3) / 4, 256>>>(d, n16, per, out); });
    timed("one wave per 25 KB, 4 rows in flight (gc_count_kernel)", [&] { sum_wave_per_bin<4><<<(n_bins + 3) / 4, 256>>>(d, n16, per, out); });
    timed("one wave per 25 KB, 8 rows in flight", [&] { sum_wave_per_bin<8><<<(n_bins + 3) / 4, 256>>>(d, n16, per, out); });
    timed("one wave per 25 KB, 16 rows in flight", [&] { sum_wave_per_bin<16><<<(n_bins + 3) / 4, 256>>>(d, n16, per, out); });
    timed("empty launch (events around nothing but a launch)", [&] { evict<<<1, 64>>>(big, 0, out); });
    return 0;
}
