// scvod_dev.h -- device-side helpers shared by the kernel translation units (scvod_kernels.hip, scvod_track.hip).
#ifndef SCVOD_DEV_H_
#define SCVOD_DEV_H_
#include "scvod_kernels.h"

namespace scvod {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ int wave_incl_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one int per thread across the workgroup; returns exclusive prefix and
// the block total.  wsum: LDS int[THREADS/64 + 1].  Ends with a barrier-safe state.
template <int THREADS>
static __device__ __forceinline__ int block_excl_scan(int v, int& total, int* wsum) {
    constexpr int NW = THREADS / 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = wave_incl_scan(v);
    if (NW == 1) {
        total = __shfl(inc, 63, 64);
        return inc - v;
    }
    __syncthreads();  // protect wsum reuse
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    if (w == 0) {
        int x = (lane < NW) ? wsum[lane] : 0;
        int xi = wave_incl_scan(x);
        if (lane < NW) wsum[lane] = xi - x;
        if (lane == NW - 1) wsum[NW] = xi;
    }
    __syncthreads();
    total = wsum[NW];
    return wsum[w] + inc - v;
}


}  // namespace scvod
#endif
