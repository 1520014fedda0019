// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against microkernels whose bytes are known (round-5 verdict, next #3).
// Development tool, not part of libscvod.so.  Every (pattern, working set) is a kernel of its own NAME, so the counter rows of a
// `rocprofv3 --pmc FETCH_SIZE` (or WRITE_SIZE) run can be matched to the JSON line this binary prints (hipEvent time + known bytes).
//   read16   16 B per lane, coalesced, grid-stride, P passes over the working set           known: W x P bytes read
//   read4     4 B per lane, coalesced                                                        known: W x P bytes read
//   write16  16 B per lane, coalesced stores                                                 known: W x P bytes written
//   copy16   read16 + write16 (the achievable copy ceiling)                                  known: W x P read + W x P written
//   gather4   4-B loads at uniformly random addresses of the table                           known: 4 B useful per access, one 64-B line / 128-B request touched
//   atomic64 64-bit atomicMin (no return) at uniformly random slots of the table             known: 8 B useful per access, RMW of one line
//   rmw16    read 16 B, add, write back, coalesced (what an in-place pass costs)             known: W x P read + W x P written
// working sets: S = 128 MiB (inside the 256 MiB Infinity Cache, beyond the 32 MiB of L2), L = 2 GiB (beyond everything).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                            \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

template <int TAG>
__global__ __launch_bounds__(256) void cal_read16(const uint4* __restrict__ p, size_t n, int passes, uint32_t* sink) {
    uint32_t acc = 0;
    for (int q = 0; q < passes; ++q)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            const uint4 v = p[i];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    if (acc == 0x12345678u) *sink = acc;
}
template <int TAG>
__global__ __launch_bounds__(256) void cal_read4(const uint32_t* __restrict__ p, size_t n, int passes, uint32_t* sink) {
    uint32_t acc = 0;
    for (int q = 0; q < passes; ++q)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) *sink = acc;
}
template <int TAG>
__global__ __launch_bounds__(256) void cal_write16(uint4* __restrict__ p, size_t n, int passes) {
    for (int q = 0; q < passes; ++q)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(q, q, q, q);
}
template <int TAG>
__global__ __launch_bounds__(256) void cal_copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n, int passes) {
    for (int q = 0; q < passes; ++q)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
template <int TAG>
__global__ __launch_bounds__(256) void cal_rmw16(uint4* __restrict__ p, size_t n, int passes) {
    for (int q = 0; q < passes; ++q)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
            uint4 v = p[i];
            v.x += 1;
            p[i] = v;
        }
}
template <int TAG>
__global__ __launch_bounds__(256) void cal_gather4(const uint32_t* __restrict__ p, size_t n_words, int per_thread, uint32_t* sink) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll 4
    for (int k = 0; k < per_thread; ++k) acc += p[mix64(t * 1315423911ull + k) % n_words];
    if (acc == 0x12345678u) *sink = acc;
}
// the same gather followed by a load of the OTHER 64-byte half of the 128-byte line: one request per pair = the first access brought the
// whole 128-B line (the raw counter tallies it at 64 B: factor 2 as for streams); two requests per pair = 64-B sector fetches (factor 1)
template <int TAG>
__global__ __launch_bounds__(256) void cal_gatherpair4(const uint32_t* __restrict__ p, size_t n_words, int per_thread, uint32_t* sink) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int k = 0; k < per_thread; ++k) {
        const size_t i = mix64(t * 1315423911ull + k) % n_words;
        const uint32_t v = p[i];
        acc += v;
        acc += p[(i ^ 16) + (v & 0)];  // (depends on the first load: issued after it returned)
    }
    if (acc == 0x12345678u) *sink = acc;
}
template <int TAG>
__global__ __launch_bounds__(256) void cal_atomic64(unsigned long long* __restrict__ p, size_t n_slots, int per_thread) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
#pragma unroll 4
    for (int k = 0; k < per_thread; ++k) {
        const uint64_t h = mix64(t * 1315423911ull + k);
        atomicMin(&p[h % n_slots], (unsigned long long)(h >> 8));
    }
}

struct Row {
    const char* name;
    const char* ws;
    double ms, read_b, write_b, accesses;
};

int main() {
    CHK(hipSetDevice(0));
    const size_t S = 128ull << 20, L = 2048ull << 20;
    void *a = nullptr, *b = nullptr;
    uint32_t* sink = nullptr;
    CHK(hipMalloc(&a, L));
    CHK(hipMalloc(&b, L));
    CHK(hipMalloc((void**)&sink, 4));
    CHK(hipMemset(a, 0x11, L));
    CHK(hipMemset(b, 0xff, L));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    std::vector<Row> rows;
    const int grid = 256 * 16;
    auto timed = [&](const char* name, const char* ws, double rb, double wb, double acc, auto launch) {
        launch();  // warm (code object, TLB)
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        CHK(hipGetLastError());
        rows.push_back({name, ws, ms, rb, wb, acc});
    };
    const int PS = 16, PL = 1;  // passes: the same 2 GiB of traffic for either working set
    timed("cal_read16<0>", "128MiB", (double)S * PS, 0, 0, [&] { hipLaunchKernelGGL(cal_read16<0>, grid, 256, 0, 0, (const uint4*)a, S / 16, PS, sink); });
    timed("cal_read16<1>", "2GiB", (double)L * PL, 0, 0, [&] { hipLaunchKernelGGL(cal_read16<1>, grid, 256, 0, 0, (const uint4*)a, L / 16, PL, sink); });
    timed("cal_read4<0>", "128MiB", (double)S * PS, 0, 0, [&] { hipLaunchKernelGGL(cal_read4<0>, grid, 256, 0, 0, (const uint32_t*)a, S / 4, PS, sink); });
    timed("cal_read4<1>", "2GiB", (double)L * PL, 0, 0, [&] { hipLaunchKernelGGL(cal_read4<1>, grid, 256, 0, 0, (const uint32_t*)a, L / 4, PL, sink); });
    timed("cal_write16<0>", "128MiB", 0, (double)S * PS, 0, [&] { hipLaunchKernelGGL(cal_write16<0>, grid, 256, 0, 0, (uint4*)b, S / 16, PS); });
    timed("cal_write16<1>", "2GiB", 0, (double)L * PL, 0, [&] { hipLaunchKernelGGL(cal_write16<1>, grid, 256, 0, 0, (uint4*)b, L / 16, PL); });
    timed("cal_copy16<0>", "128MiB", (double)S * PS, (double)S * PS, 0, [&] { hipLaunchKernelGGL(cal_copy16<0>, grid, 256, 0, 0, (const uint4*)a, (uint4*)b, S / 16, PS); });
    timed("cal_copy16<1>", "2GiB", (double)L * PL, (double)L * PL, 0, [&] { hipLaunchKernelGGL(cal_copy16<1>, grid, 256, 0, 0, (const uint4*)a, (uint4*)b, L / 16, PL); });
    timed("cal_rmw16<0>", "128MiB", (double)S * PS, (double)S * PS, 0, [&] { hipLaunchKernelGGL(cal_rmw16<0>, grid, 256, 0, 0, (uint4*)b, S / 16, PS); });
    timed("cal_rmw16<1>", "2GiB", (double)L * PL, (double)L * PL, 0, [&] { hipLaunchKernelGGL(cal_rmw16<1>, grid, 256, 0, 0, (uint4*)b, L / 16, PL); });
    const int per = 64;
    const double acc = (double)grid * 256 * per;  // 67 M accesses
    timed("cal_gather4<0>", "128MiB", 4 * acc, 0, acc, [&] { hipLaunchKernelGGL(cal_gather4<0>, grid, 256, 0, 0, (const uint32_t*)a, S / 4, per, sink); });
    timed("cal_gather4<1>", "2GiB", 4 * acc, 0, acc, [&] { hipLaunchKernelGGL(cal_gather4<1>, grid, 256, 0, 0, (const uint32_t*)a, L / 4, per, sink); });
    timed("cal_gatherpair4<0>", "128MiB", 8 * acc, 0, acc, [&] { hipLaunchKernelGGL(cal_gatherpair4<0>, grid, 256, 0, 0, (const uint32_t*)a, S / 4, per, sink); });
    timed("cal_gatherpair4<1>", "2GiB", 8 * acc, 0, acc, [&] { hipLaunchKernelGGL(cal_gatherpair4<1>, grid, 256, 0, 0, (const uint32_t*)a, L / 4, per, sink); });
    CHK(hipMemset(b, 0xff, L));
    timed("cal_atomic64<0>", "128MiB", 8 * acc, 8 * acc, acc, [&] { hipLaunchKernelGGL(cal_atomic64<0>, grid, 256, 0, 0, (unsigned long long*)b, S / 8, per); });
    timed("cal_atomic64<1>", "2GiB", 8 * acc, 8 * acc, acc, [&] { hipLaunchKernelGGL(cal_atomic64<1>, grid, 256, 0, 0, (unsigned long long*)b, L / 8, per); });
    printf("{\"rows\": [");
    for (size_t i = 0; i < rows.size(); ++i)
        printf("%s{\"kernel\": \"%s\", \"working_set\": \"%s\", \"ms\": %.4f, \"known_read_bytes\": %.0f, \"known_write_bytes\": %.0f, \"accesses\": %.0f}", i ? ", " : "",
               rows[i].name, rows[i].ws, rows[i].ms, rows[i].read_b, rows[i].write_b, rows[i].accesses);
    printf("], \"launches_per_kernel\": 2}\n");
    return 0;
}
