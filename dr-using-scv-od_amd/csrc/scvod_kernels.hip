// scvod_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the SCV-OD hot path.
//
// Stage map (reference file:line -> kernel):
//   PatchWork::estimate_ground prologue + pc2czm   patchwork.h:277-325,416-459 -> k_pw_classify, k_pw_offsets, k_pw_scatter,
//                                                                                  k_pw_order_*, k_pw_sort_wave, k_pw_sort<...>
//   extract_piecewiseground / seeds / plane fit    patchwork.h:217-268,463-504 -> k_pw_fit_coop<16|64>, k_pw_fit
//   gating + emission order                        patchwork.h:326-391         -> fit_finish, k_pw_arrange, k_emit_offsets, k_emit
//   SSC::makeApriVec                               ssc.cpp:155-195             -> k_emit (fused, compact apri_vec), k_apri_expand,
//                                                                                  k_bin_direct
//   SSC::makeHashCloud                             ssc.cpp:253-289             -> k_vx_count, k_vx_offsets, k_vx_scatter, k_vx_order_*,
//                                                                                  k_vx_bucket<...>, k_vx_final*
//   SSC::tracking bulk part                        ssc.cpp:1274-1321           -> k_track_probe_pair, k_track_probe, k_track_unique_bits
//   SSC::clusterAndCreateFrame (next row f-1)      ssc.cpp:299-393             -> k_cc_scan (one workgroup per scan, union-find in LDS)
//   refineClusterByBoundingBox + recognize rules   ssc.cpp:437-467,849-872     -> k_cc_scan (same workgroup, boxes in LDS)
//   SSC::getCloud filter + pcl::VoxelGrid (f-3)    ssc.cpp:1063-1076,1103-1106 -> k_vg_minmax/keys/lut/outoff/final, k_vx_bucket<..., 1>
//   kd-tree look-ups of evaluate.cpp:79-145                                    -> k_nn_count/fill/query, k_nn_brute_list
//
// Design notes: the path is gather/scatter + histogramming + short serial
// fp32 chains; there is no dense contraction, so no MFMA.  Bit-exact parity with the CPU
// restatement requires (a) the z-sort order inside every patch, (b) strictly sequential
// fp32 accumulation of the 9 covariance moments in that order, (c) sequential per-voxel
// intensity sums in point order.  Parallelism therefore comes from patches x scans and
// voxels x scans (and, inside a large patch, from its nine independent sums), not from
// tree reductions of those sums.
#include <type_traits>

#include "scvod_dev.h"

namespace scvod {

// development build only (make prof, tools/kernel_phases.py): phase clocks inside the list-driven kernels.  Thread 0 of a workgroup
// adds the 100 MHz wall clock between two marks (behind a barrier) to g_prof[kernel][phase]; never in the shipped library.
#ifdef SCVOD_PROFILE
__device__ unsigned long long g_prof[8][16];
#define PROF_BEGIN() unsigned long long t_prev = wall_clock64()
#define PROF_RESET()                                   \
    do {                                               \
        __syncthreads();                               \
        if (threadIdx.x == 0) t_prev = wall_clock64(); \
    } while (0)
#define PROF_MARK(k, i)                                      \
    do {                                                     \
        __syncthreads();                                     \
        if (threadIdx.x == 0) {                              \
            const unsigned long long t_now = wall_clock64(); \
            atomicAdd(&g_prof[k][i], t_now - t_prev);        \
            t_prev = t_now;                                  \
        }                                                    \
    } while (0)
#else
#define PROF_BEGIN()
#define PROF_RESET()
#define PROF_MARK(k, i)
#endif
#define CC_MARK(i) PROF_MARK(0, i)
#define CCW_MARK(i) PROF_MARK(4, i)  // the windowed search of the generic k_cc_scan variant

// Normalised bitonic network (every comparator puts the minimum at the lower index), valid for any n:
// indices >= n act as +inf (all-ones key) and are never read or written.  Works on LDS or global (flat)
// storage.  Register blocking: runs of 8 are sorted in registers (19-comparator network = stages
// k = 2, 4, 8), and the butterfly steps of each later stage are taken three levels (8 elements per
// thread) at a time, which cuts barriers and LDS traffic ~2.3x against one level per pass.
template <typename T>
__device__ __forceinline__ void cswap(T& x, T& y) {
    const T lo = x < y ? x : y;
    const T hi = x < y ? y : x;
    x = lo;
    y = hi;
}

// PAD: LDS layout with one spare slot after every 8 elements (slot(i) = i + i/8).  With 8-byte keys the
// strided butterfly passes (runs of jl lanes every 8*jl elements) and the runs-of-8 passes would otherwise
// put a half-wave on 4-8 banks; with the pad every pass is conflict-free or at worst 2-way.
template <bool PAD>
__device__ __forceinline__ int sort_slot(int i) {
    return PAD ? i + (i >> 3) : i;
}

template <int THREADS, bool PAD, typename T>
__device__ __forceinline__ void block_bitonic_sort(T* a, int n) {
#define IX(i) sort_slot<PAD>(i)
    if (n <= 1) return;
    const T INF = ~(T)0;
    int np2 = 8;
    while (np2 < n) np2 <<= 1;
    // stage k <= 8: in-register sort of aligned runs of 8
    for (int g = threadIdx.x; g < (np2 >> 3); g += THREADS) {
        const int b = g << 3;
        if (b >= n) continue;
        T e[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) e[m] = (b + m < n) ? a[IX(b + m)] : INF;
        cswap(e[0], e[1]); cswap(e[2], e[3]); cswap(e[4], e[5]); cswap(e[6], e[7]);
        cswap(e[0], e[2]); cswap(e[1], e[3]); cswap(e[4], e[6]); cswap(e[5], e[7]);
        cswap(e[1], e[2]); cswap(e[5], e[6]);
        cswap(e[0], e[4]); cswap(e[1], e[5]); cswap(e[2], e[6]); cswap(e[3], e[7]);
        cswap(e[2], e[4]); cswap(e[3], e[5]);
        cswap(e[1], e[2]); cswap(e[3], e[4]); cswap(e[5], e[6]);
#pragma unroll
        for (int m = 0; m < 8; ++m)
            if (b + m < n) a[IX(b + m)] = e[m];
    }
    __syncthreads();
    const int half = np2 >> 1;
    for (int k = 16; k <= np2; k <<= 1) {
        const int hk = k >> 1;
        // mirror step
        for (int t = threadIdx.x; t < half; t += THREADS) {
            const int blk = t / hk, off = t - blk * hk;
            const int i = blk * k + off, j = blk * k + (k - 1 - off);
            if (j < n) {
                T x = a[IX(i)], y = a[IX(j)];
                if (x > y) {
                    a[IX(i)] = y;
                    a[IX(j)] = x;
                }
            }
        }
        __syncthreads();
        int j = k >> 2;  // first butterfly distance of this stage (>= 4 because k >= 16)
        while (j >= 1) {
            if (j >= 4) {
                const int jl = j >> 2;  // three levels: distances 4*jl, 2*jl, jl
                for (int t = threadIdx.x; t < (np2 >> 3); t += THREADS) {
                    const int b = (t / jl) * (jl << 3) + (t % jl);
                    if (b >= n) continue;
                    T e[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m) e[m] = (b + m * jl < n) ? a[IX(b + m * jl)] : INF;
                    cswap(e[0], e[4]); cswap(e[1], e[5]); cswap(e[2], e[6]); cswap(e[3], e[7]);
                    cswap(e[0], e[2]); cswap(e[1], e[3]); cswap(e[4], e[6]); cswap(e[5], e[7]);
                    cswap(e[0], e[1]); cswap(e[2], e[3]); cswap(e[4], e[5]); cswap(e[6], e[7]);
#pragma unroll
                    for (int m = 0; m < 8; ++m)
                        if (b + m * jl < n) a[IX(b + m * jl)] = e[m];
                }
                j >>= 3;
            } else if (j == 2) {
                for (int t = threadIdx.x; t < (np2 >> 2); t += THREADS) {
                    const int b = t << 2;
                    if (b >= n) continue;
                    T e[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) e[m] = (b + m < n) ? a[IX(b + m)] : INF;
                    cswap(e[0], e[2]); cswap(e[1], e[3]);
                    cswap(e[0], e[1]); cswap(e[2], e[3]);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        if (b + m < n) a[IX(b + m)] = e[m];
                }
                j = 0;
            } else {  // j == 1
                for (int t = threadIdx.x; t < half; t += THREADS) {
                    const int i = t << 1;
                    if (i + 1 < n) {
                        T x = a[IX(i)], y = a[IX(i + 1)];
                        if (x > y) {
                            a[IX(i)] = y;
                            a[IX(i + 1)] = x;
                        }
                    }
                }
                j = 0;
            }
            __syncthreads();
        }
    }
}
#undef IX

// Histogram / rank atomics on an LDS bin table when neighbouring lanes mostly hit the SAME bin (consecutive points of a
// scan fall into the same patch / key bucket): same-address LDS atomics serialise lane by lane, so the wave first groups
// its lanes by bin and one lane per distinct bin adds the group's count.  wave_bin_rank returns the old bin value + the
// lane's rank inside its group (what atomicAdd(&hist[bin], 1) would have returned for SOME serialisation); bin < 0 = idle.
__device__ __forceinline__ int wave_bin_rank(int* hist, int bin) {
    const int lane = threadIdx.x & 63;
    int rank = 0;
    bool todo = bin >= 0;
    while (__any(todo)) {
        const int first = __ffsll((long long)__ballot(todo)) - 1;
        const int b0 = __shfl(bin, first);
        const bool mine = todo && (bin == b0);
        const unsigned long long mask = __ballot(mine);
        int old = 0;
        if (lane == first) old = atomicAdd(&hist[b0], __popcll(mask));
        old = __shfl(old, first);
        if (mine) {
            rank = old + __popcll(mask & ((1ull << lane) - 1ull));
            todo = false;
        }
    }
    return rank;
}
__device__ __forceinline__ void wave_bin_add(int* hist, int bin) {
    const int lane = threadIdx.x & 63;
    bool todo = bin >= 0;
    while (__any(todo)) {
        const int first = __ffsll((long long)__ballot(todo)) - 1;
        const int b0 = __shfl(bin, first);
        const bool mine = todo && (bin == b0);
        const unsigned long long mask = __ballot(mine);
        if (lane == first) atomicAdd(&hist[b0], __popcll(mask));
        if (mine) todo = false;
    }
}

// ---- power-of-two bitonic sort for the LDS tiers -------------------------------------------------
// Classic bitonic network on np2 = 2^q slots (slots >= n hold +inf, the LDS tiers always have room
// for them), every level a butterfly whose direction is given by bit k of the element index, so ALL
// lg(k) levels of a stage can be register-blocked: a thread takes 2^t elements (t <= 4) that are
// closed under t consecutive levels, sorts/merges them in registers and writes them back.  Runs of 16
// are sorted entirely in registers first.  For np2 = 2048 this is 19 LDS passes instead of the 66 of a
// level-per-pass network; no bounds predicates anywhere.
// ascending compare-exchange of unique unsigned keys; the borrow of x - y is the x < y flag (two full-rate
// 32-bit VALU ops instead of a 64-bit compare)
// Sort keys are (32-bit primary key, 19-bit index) packed under a fixed exponent: 0x3ff << 52 | key << 19 | idx.
// Every key is then a normal double in [1, 2) whose numeric order IS the integer order of (key, idx), so a
// compare-exchange is one v_min_f64 + one v_max_f64 instead of a 64-bit compare and four selects.  The pad value
// 2.0 sorts after every key.  (Indices fit because a scan holds at most 2^19 points, include/scvod.h.)
constexpr int kKeyIdxBits = 19;
static_assert((1 << kKeyIdxBits) >= SCVOD_MAX_SCAN_POINTS, "index field of the sort keys");
constexpr unsigned long long kKeyExp = 0x3ffull << 52;
constexpr unsigned long long kKeyPad = 0x4000000000000000ull;
__device__ __forceinline__ unsigned long long pack_key(uint32_t key, uint32_t idx) {
    return kKeyExp | ((unsigned long long)key << kKeyIdxBits) | idx;
}
__device__ __forceinline__ uint32_t key_idx(unsigned long long k) { return (uint32_t)k & ((1u << kKeyIdxBits) - 1u); }
__device__ __forceinline__ uint32_t key_major(unsigned long long k) { return (uint32_t)(k >> kKeyIdxBits); }

// index triple of an apri point, packed by k_emit / k_bin_direct / k_apri_split: 11 + 11 + 10 bits, each index clamped
// to [-2, limit] and biased by 2 (indices at or beyond -2 / dim + 1 have no in-grid neighbour at all, so clamping them
// keeps both the run comparison and the neighbourhood exact)
__device__ __forceinline__ int32_t pack_idx3(int r, int s, int a) {
    r = min(max(r, -2), 2045) + 2;
    s = min(max(s, -2), 2045) + 2;
    a = min(max(a, -2), 1021) + 2;
    return r | (s << 11) | (a << 22);
}

__device__ __forceinline__ void cswap_asc(unsigned long long& x, unsigned long long& y) {
    double lo, hi;
    const double dx = __longlong_as_double((long long)x), dy = __longlong_as_double((long long)y);
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(dx), "v"(dy));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(dx), "v"(dy));
    x = (unsigned long long)__double_as_longlong(lo);
    y = (unsigned long long)__double_as_longlong(hi);
}
__device__ __forceinline__ void cswap_asc(uint32_t& x, uint32_t& y) {
    const uint32_t lo = x < y ? x : y;
    const uint32_t hi = x < y ? y : x;
    x = lo;
    y = hi;
}

template <int LT, typename T>  // ascending butterfly levels with local distances 2^(LT-1) .. 1 on 2^LT registers
__device__ __forceinline__ void reg_merge(T (&e)[1 << LT]) {
#pragma unroll
    for (int d = (1 << LT) >> 1; d >= 1; d >>= 1) {
#pragma unroll
        for (int m = 0; m < (1 << LT); ++m)
            if ((m & d) == 0) cswap_asc(e[m], e[m + d]);
    }
}

// A descending merge of a group is the ascending merge of the group read in reverse order, so the direction
// of a group only changes WHERE its registers come from / go to (m ^ rev), never the compare-exchange code.
template <int THREADS, bool PAD, int LT, typename T>
__device__ __forceinline__ void bitonic_pass(T* a, int np2, int k, int r) {
    // levels with distances 2^(r-1) .. 2^(r-LT) of stage k
    constexpr int E = 1 << LT;
    const int jl = 1 << (r - LT);
    for (int t = threadIdx.x; t < (np2 >> LT); t += THREADS) {
        const int b = ((t >> (r - LT)) << r) | (t & (jl - 1));
        const int rev = ((b & k) == 0) ? 0 : (E - 1);
        T e[E];
#pragma unroll
        for (int m = 0; m < E; ++m) e[m] = a[sort_slot<PAD>(b + (m ^ rev) * jl)];
        reg_merge<LT>(e);
#pragma unroll
        for (int m = 0; m < E; ++m) a[sort_slot<PAD>(b + (m ^ rev) * jl)] = e[m];
    }
    __syncthreads();
}

// stages k = 2 .. E of the network on a run of E elements held in registers; directions inside the run are compile-time,
// the direction of the whole run (bit E of its base b) is applied by storing it reversed
template <bool PAD, int LGE, typename T>
__device__ __forceinline__ void bitonic_run_to_lds(T (&e)[1 << LGE], T* a, int b) {
    constexpr int E = 1 << LGE;
#pragma unroll
    for (int kk = 2; kk <= E; kk <<= 1) {
#pragma unroll
        for (int d = kk >> 1; d >= 1; d >>= 1) {
#pragma unroll
            for (int m = 0; m < E; ++m)
                if ((m & d) == 0) {
                    if (kk == E || (m & kk) == 0)
                        cswap_asc(e[m], e[m + d]);
                    else
                        cswap_asc(e[m + d], e[m]);
                }
        }
    }
    const int rev = ((b & E) == 0) ? 0 : (E - 1);
#pragma unroll
    for (int m = 0; m < E; ++m) a[sort_slot<PAD>(b + (m ^ rev))] = e[m];
}

template <int THREADS, bool PAD, int LGE, typename T>
__device__ __forceinline__ void block_bitonic_merge_stages(T* a, int np2);

template <int THREADS, bool PAD, int LGE, typename T>
__device__ __forceinline__ void block_bitonic_sort_pow2(T* a, int np2) {  // np2 >= 2^LGE, power of two
    constexpr int E = 1 << LGE;
    for (int g = threadIdx.x; g < (np2 >> LGE); g += THREADS) {
        const int b = g << LGE;
        T e[E];
#pragma unroll
        for (int m = 0; m < E; ++m) e[m] = a[sort_slot<PAD>(b + m)];
        bitonic_run_to_lds<PAD, LGE>(e, a, b);
    }
    __syncthreads();
    block_bitonic_merge_stages<THREADS, PAD, LGE>(a, np2);
}

// the same sort for np2 == THREADS << LGE unsorted keys that arrive in registers (e[it] = the thread's it-th coalesced load):
// which key starts in which slot is immaterial, so thread t's loads ARE run t -- no staging pass through LDS
template <int THREADS, bool PAD, int LGE, typename T>
__device__ __forceinline__ void block_bitonic_sort_pow2_regs(T (&e)[1 << LGE], T* a) {
    bitonic_run_to_lds<PAD, LGE>(e, a, (int)threadIdx.x << LGE);
    __syncthreads();
    block_bitonic_merge_stages<THREADS, PAD, LGE>(a, THREADS << LGE);
}

template <int THREADS, bool PAD, int LGE, typename T>
__device__ __forceinline__ void block_bitonic_merge_stages(T* a, int np2) {
    constexpr int E = 1 << LGE;
    int lgk = LGE + 1;
    for (int k = 2 * E; k <= np2; k <<= 1, ++lgk) {
        int r = lgk;  // levels still to do in this stage (distances 2^(r-1) .. 1)
        const int first = (r % LGE) ? (r % LGE) : LGE;  // leading partial pass, the rest are full LGE-level passes
        if (first == 1)
            bitonic_pass<THREADS, PAD, 1>(a, np2, k, r);
        else if (first == 2)
            bitonic_pass<THREADS, PAD, 2>(a, np2, k, r);
        else if (first == 3)
            bitonic_pass<THREADS, PAD, 3>(a, np2, k, r);
        else
            bitonic_pass<THREADS, PAD, 4>(a, np2, k, r);
        r -= first;
        while (r > 0) {
            bitonic_pass<THREADS, PAD, LGE>(a, np2, k, r);
            r -= LGE;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Patchwork stage 1: per point patch id + per-(scan, patch) histogram.
// grid = (ceil(max_scan_pts / (256*ITEMS)), B), block = 256.  Coalesced float4 loads.
// ------------------------------------------------------------------------------------------
constexpr int kClsThreads = 256;
constexpr int kClsItems = 8;

__global__ __launch_bounds__(kClsThreads) void k_pw_classify(DevParams P, Arena A) {
    __shared__ int hist[kMaxPatches];
    const int s = blockIdx.y;
    const int base = A.scan_off[s];
    const int n = A.scan_off[s + 1] - base;
    const int start = blockIdx.x * (kClsThreads * kClsItems);
    if (start >= n) return;
    for (int b = threadIdx.x; b < P.n_patches; b += kClsThreads) hist[b] = 0;
    __syncthreads();
    // all loads first (8 x 16 B in flight per lane), then the fp64 zone / ring / sector arithmetic
    float4 pt[kClsItems];
#pragma unroll
    for (int it = 0; it < kClsItems; ++it) {
        const int i = start + it * kClsThreads + threadIdx.x;
        pt[it] = A.pts[base + min(i, n - 1)];
    }
#pragma unroll
    for (int it = 0; it < kClsItems; ++it) {
        const int i = start + it * kClsThreads + threadIdx.x;
        if (i < n) {
            const float4 p = pt[it];
            int pid = czm_patch_of(P.czm, p.x, p.y, p.z);
            A.pid[base + i] = (int16_t)pid;
            A.zkey[base + i] = float_sort_key(p.z);
            if (pid >= 0) atomicAdd(&hist[pid], 1);  // (grouping the wave's lanes by bin first costs more than it saves here)
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < P.n_patches; b += kClsThreads) {
        int c = hist[b];
        if (c) atomicAdd(&A.patch_count[s * kMaxPatches + b], c);
    }
}

// per scan exclusive scan of the patch histogram.  grid = B, block = 1024.
__global__ __launch_bounds__(1024) void k_pw_offsets(DevParams P, Arena A) {
    __shared__ int wsum[17];
    const int s = blockIdx.x;
    int c = (threadIdx.x < (unsigned)P.n_patches) ? A.patch_count[s * kMaxPatches + threadIdx.x] : 0;
    int total;
    int ex = block_excl_scan<1024>(c, total, wsum);
    if (threadIdx.x <= (unsigned)P.n_patches) A.patch_off[s * (kMaxPatches + 1) + threadIdx.x] = ex;
}

// scatter (z key, idx) into patch-major order.  Position inside the patch is arbitrary (the
// patch kernel sorts); block-aggregated so that there is one global atomic per (block, patch).
__global__ __launch_bounds__(kClsThreads) void k_pw_scatter(DevParams P, Arena A) {
    __shared__ int hist[kMaxPatches];
    const int s = blockIdx.y;
    const int base = A.scan_off[s];
    const int n = A.scan_off[s + 1] - base;
    const int start = blockIdx.x * (kClsThreads * kClsItems);
    if (start >= n) return;
    for (int b = threadIdx.x; b < P.n_patches; b += kClsThreads) hist[b] = 0;
    __syncthreads();
    int pid[kClsItems], rank[kClsItems];
    uint32_t zk[kClsItems];
#pragma unroll
    for (int it = 0; it < kClsItems; ++it) {
        const int i = start + it * kClsThreads + threadIdx.x;
        pid[it] = (i < n) ? (int)A.pid[base + i] : -1;
        zk[it] = A.zkey[base + min(i, n - 1)];
    }
#pragma unroll
    for (int it = 0; it < kClsItems; ++it)
        rank[it] = wave_bin_rank(hist, pid[it]);
    __syncthreads();
    for (int b = threadIdx.x; b < P.n_patches; b += kClsThreads) {
        int c = hist[b];
        if (c) hist[b] = A.patch_off[s * (kMaxPatches + 1) + b] + atomicAdd(&A.patch_cursor[s * kMaxPatches + b], c);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kClsItems; ++it) {
        int i = start + it * kClsThreads + threadIdx.x;
        if (i < n && pid[it] >= 0) A.keys[(size_t)base + hist[pid[it]] + rank[it]] = pack_key(zk[it], (uint32_t)i);
    }
}

// ------------------------------------------------------------------------------------------
// Patchwork stage 2 (extract_piecewiseground + gating), split so that every phase has the
// parallel shape that fits it:
//   k_pw_order_*  counting sort of the live patches of the whole batch by quarter-octave size class into order[];
//                 every later kernel is list-driven (persistent grids striding over their slice of the list).
//   k_pw_sort*    tiers by size class: LDS bitonic sort of the patch's (z, idx) keys, then the patch's points are
//                 written in sorted order (packed xyz + input index).
//   k_pw_fit_coop patches of >= 512 points (>= 64 for a handful of scans): GL lanes per patch; products in parallel,
//                 the nine sequential fp32 sums on nine lanes.
//   k_pw_fit      the smaller patches, ONE LANE PER PATCH (the 3x3 Jacobi SVD then costs one lane, not one wave),
//                 reads staged cooperatively through LDS.
//   k_pw_arrange  one wave per patch: final plane test, [ground part | non-ground part] arrangement
//                 and the per-patch counters the ordered emission needs.
// ------------------------------------------------------------------------------------------
// size classes (pw_size_class) that bound the sort tiers: class 40 <=> n >= 1024, class 48 <=> n >= 4096
constexpr int kClassXS = 32, kClassM = 40, kClassM2 = 44, kClassL = 48;  // n >= 256 / 1024 / 2048 / 4096
#ifndef SCVOD_FIT_COOP_CLASS
#define SCVOD_FIT_COOP_CLASS 36
#endif
constexpr int kClassFitCoop = SCVOD_FIT_COOP_CLASS;  // 36: n >= 512: plane fit by 16 lanes per patch (k_pw_fit_coop)

// order[] lists live items by descending size class; positions of classes [C_LO, C_HI]
__device__ __forceinline__ void order_range(const int32_t* off, int c_lo, int c_hi, int& lo, int& hi) {
    lo = off[c_hi];
    hi = (c_lo == 0) ? off[64] : off[c_lo - 1];
}

template <int CAP, int THREADS, int C_LO, int C_HI, int LGE>
__global__ __launch_bounds__(THREADS) void k_pw_sort(DevParams P, Arena A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int lo, hi;
    order_range(A.order_off, C_LO, C_HI, lo, hi);
    constexpr int PK = (CAP == 4096) ? 1 : 2;  // (profiling build: the two large tiers are clocked)
    PROF_BEGIN();
    for (int w = lo + blockIdx.x; w < hi; w += gridDim.x) {
        const int4 item = A.order[w];
        const int n = item.y, base = item.z, off = item.w;
        const bool in_lds = (n <= CAP);
        unsigned long long* keys;
        if (CAP >= 4096) PROF_RESET();
        if (in_lds) {
            keys = (unsigned long long*)smem;
            int np2 = 1 << LGE;
            while (np2 < n) np2 <<= 1;
            // a full tier: all of a thread's key loads in flight together (coalesced), sorted straight from the registers
            constexpr int IT = CAP / THREADS;
            const unsigned long long* gk = (const unsigned long long*)A.keys + (size_t)base + off;
            bool from_regs = false;
            if constexpr (IT == (1 << LGE) && CAP >= 4096) {  // (the small tiers are occupancy-bound: they keep their registers)
                if (np2 == CAP) {
                    from_regs = true;
                    unsigned long long tmp[IT];
#pragma unroll
                    for (int it = 0; it < IT; ++it) {
                        const int j = it * THREADS + (int)threadIdx.x;
                        tmp[it] = (j < n) ? gk[j] : kKeyPad;
                    }
                    if (CAP >= 4096) PROF_MARK(PK, 0);
                    block_bitonic_sort_pow2_regs<THREADS, true, LGE>(tmp, keys);
                }
            }
            if (!from_regs) {
                for (int j = threadIdx.x; j < np2; j += THREADS) keys[sort_slot<true>(j)] = (j < n) ? gk[j] : kKeyPad;
                __syncthreads();
                if (CAP >= 4096) PROF_MARK(PK, 0);
                block_bitonic_sort_pow2<THREADS, true, LGE>(keys, np2);
            }
            if (CAP >= 4096) PROF_MARK(PK, 1);
        } else {
            keys = (unsigned long long*)(A.keys + (size_t)base + off);  // oversize patch: sort in place in global memory
            block_bitonic_sort<THREADS, false>(keys, n);
        }
        Xyz* dst = A.sorted_xyz + (size_t)base + off;
        uint32_t* dsti = A.sorted_idx + (size_t)base + off;
        // four independent gathers in flight per thread (the index comes from LDS, the point from anywhere in the scan)
        for (int j0 = threadIdx.x; j0 < n; j0 += 4 * THREADS) {
            uint32_t id[4];
            float4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = min(j0 + u * THREADS, n - 1);
                id[u] = key_idx(keys[in_lds ? sort_slot<true>(j) : j]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = A.pts[base + id[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * THREADS;
                if (j < n) {
                    Xyz o;
                    o.x = q[u].x;
                    o.y = q[u].y;
                    o.z = q[u].z;
                    dst[j] = o;
                    dsti[j] = id[u];
                }
            }
        }
        __syncthreads();  // LDS is reused by the next item
        if (CAP >= 4096) PROF_MARK(PK, 2);
    }
}

// patches with fewer than 64 points: one key per lane, bitonic network over the wave with xor-shuffles
// (21 compare-exchange steps, no LDS, no barriers); four patches per 256-thread workgroup
constexpr int kClassWave = 24;  // pw_size_class(64)
__global__ __launch_bounds__(256) void k_pw_sort_wave(DevParams P, Arena A) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int lo, hi;
    order_range(A.order_off, 0, kClassWave - 1, lo, hi);
    for (int w = lo + blockIdx.x * 4 + wave; w < hi; w += gridDim.x * 4) {
        const int4 item = A.order[w];
        const int n = item.y, base = item.z, off = item.w;  // n < 64
        unsigned long long key = (lane < n) ? A.keys[(size_t)base + off + lane] : kKeyPad;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j >= 1; j >>= 1) {
                const unsigned long long other = __shfl_xor(key, j, 64);
                const bool take_min = (((lane & k) == 0) == ((lane & j) == 0));
                unsigned long long mn = key, mx = other;
                cswap_asc(mn, mx);
                key = take_min ? mn : mx;
            }
        }
        if (lane < n) {
            const uint32_t id = key_idx(key);
            const float4 q = A.pts[base + id];
            Xyz o;
            o.x = q.x;
            o.y = q.y;
            o.z = q.z;
            A.sorted_xyz[(size_t)base + off + lane] = o;
            A.sorted_idx[(size_t)base + off + lane] = id;
        }
    }
}

__device__ __forceinline__ int pw_size_class(int n) {  // quarter-octave classes, 0..63
    int lg = 31 - __clz(n);
    int frac = (lg >= 2) ? ((n >> (lg - 2)) & 3) : 0;
    int c = lg * 4 + frac;
    return c > 63 ? 63 : c;
}

__global__ __launch_bounds__(256) void k_pw_order_count(DevParams P, Arena A) {
    __shared__ int hist[64];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < A.n_scans * P.n_patches) {
        const int s = t / P.n_patches, p = t - s * P.n_patches;
        const int n = A.patch_count[s * kMaxPatches + p];
        if (n <= P.czm.num_min_pts) {  // skipped patch (patchwork.h:331): record only
            PatchRec r = {n, 0, 0, 0, 0};
            A.patch_rec[s * kMaxPatches + p] = r;
            scvod_patch_plane pl = {};
            pl.n_pts = n;
            A.planes[s * kMaxPatches + p] = pl;
        } else {
            atomicAdd(&hist[pw_size_class(n)], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64 && hist[threadIdx.x]) atomicAdd(&A.order_hist[threadIdx.x], hist[threadIdx.x]);
}

__global__ __launch_bounds__(64) void k_pw_order_offsets(Arena A) {
    // descending size: class 63 first
    const int c = 63 - threadIdx.x;
    const int v = A.order_hist[c];
    const int inc = wave_incl_scan(v);
    A.order_off[c] = inc - v;
    if (threadIdx.x == 63) A.order_off[64] = inc;  // number of live patches
}

__global__ __launch_bounds__(256) void k_pw_order_scatter(DevParams P, Arena A) {
    __shared__ int hist[64];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    int c = -1, rank = 0;
    int4 item = make_int4(0, 0, 0, 0);
    if (t < A.n_scans * P.n_patches) {
        const int s = t / P.n_patches, p = t - s * P.n_patches;
        const int n = A.patch_count[s * kMaxPatches + p];
        if (n > P.czm.num_min_pts) {
            c = pw_size_class(n);
            rank = atomicAdd(&hist[c], 1);
            item = make_int4(s * kMaxPatches + p, n, A.scan_off[s], A.patch_off[s * (kMaxPatches + 1) + p]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64 && hist[threadIdx.x])
        hist[threadIdx.x] = A.order_off[threadIdx.x] + atomicAdd(&A.order_cursor[threadIdx.x], hist[threadIdx.x]);
    __syncthreads();
    if (c >= 0) A.order[hist[c] + rank] = item;  // {scan*1024 + patch, n, scan base, patch offset}
}

// the same ordering for voxel buckets (counts in vb_count, min size 1)
__global__ __launch_bounds__(256) void k_vx_order_count(DevParams P, Arena A) {
    __shared__ int hist[64];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < A.n_scans * P.n_buckets) {
        const int s = t / P.n_buckets, b = t - s * P.n_buckets;
        const int m = A.vb_count[s * kMaxBuckets + b];
        if (m > 0)
            atomicAdd(&hist[pw_size_class(m)], 1);
        else
            A.vb_nvox[s * kMaxBuckets + b] = 0;
    }
    __syncthreads();
    if (threadIdx.x < 64 && hist[threadIdx.x]) atomicAdd(&A.vorder_hist[threadIdx.x], hist[threadIdx.x]);
}

__global__ __launch_bounds__(64) void k_vx_order_offsets(Arena A) {
    const int c = 63 - threadIdx.x;
    const int v = A.vorder_hist[c];
    const int inc = wave_incl_scan(v);
    A.vorder_off[c] = inc - v;
    if (threadIdx.x == 63) A.vorder_off[64] = inc;
}

__global__ __launch_bounds__(256) void k_vx_order_scatter(DevParams P, Arena A) {
    __shared__ int hist[64];
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    int c = -1, rank = 0;
    int4 item = make_int4(0, 0, 0, 0);
    if (t < A.n_scans * P.n_buckets) {
        const int s = t / P.n_buckets, b = t - s * P.n_buckets;
        const int m = A.vb_count[s * kMaxBuckets + b];
        if (m > 0) {
            c = pw_size_class(m);
            rank = atomicAdd(&hist[c], 1);
            item = make_int4(s * kMaxBuckets + b, m, A.scan_off[s], A.vb_off[s * (kMaxBuckets + 1) + b]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64 && hist[threadIdx.x])
        hist[threadIdx.x] = A.vorder_off[threadIdx.x] + atomicAdd(&A.vorder_cursor[threadIdx.x], hist[threadIdx.x]);
    __syncthreads();
    if (c >= 0) A.vorder[hist[c] + rank] = item;
}

// per-point plane residual, Eigen GEMV order: fl(fl(x*n0 + y*n1) + z*n2)
__device__ __forceinline__ float plane_res(const Xyz& q, float n0, float n1, float n2) {
    float r = q.x * n0;
    r = r + q.y * n1;
    r = r + q.z * n2;
    return r;
}

struct FitState {
    float cov[9];
    float mean0, mean1, mean2;
    float n0, n1, n2, thd;
    float sv0, sv1, sv2;
};
__device__ __forceinline__ void fit_state_init(FitState& F) {
#pragma unroll
    for (int k = 0; k < 9; ++k) F.cov[k] = 0.f;
    F.mean0 = F.mean1 = F.mean2 = 0.f;
    F.n0 = F.n1 = F.n2 = F.thd = 0.f;
    F.sv0 = F.sv1 = F.sv2 = 0.f;
}

// estimate_plane_ (patchwork.h:206-232) from the nine sums of pcl::computeMeanAndCovarianceMatrix
__device__ __forceinline__ void fit_update(FitState& F, const CzmParams& cz, float a0, float a1, float a2, float a3, float a4,
                                           float a5, float a6, float a7, float a8, int m) {
    if (m != 0) {  // an empty set leaves cov_/pc_mean_ untouched (PCL)
        const float fn = (float)m;
        a0 = a0 / fn;
        a1 = a1 / fn;
        a2 = a2 / fn;
        a3 = a3 / fn;
        a4 = a4 / fn;
        a5 = a5 / fn;
        a6 = a6 / fn;
        a7 = a7 / fn;
        a8 = a8 / fn;
        F.mean0 = a6;
        F.mean1 = a7;
        F.mean2 = a8;
        F.cov[0] = a0 - a6 * a6;
        F.cov[1] = a1 - a6 * a7;
        F.cov[2] = a2 - a6 * a8;
        F.cov[4] = a3 - a7 * a7;
        F.cov[5] = a4 - a7 * a8;
        F.cov[8] = a5 - a8 * a8;
        F.cov[3] = F.cov[1];
        F.cov[6] = F.cov[2];
        F.cov[7] = F.cov[5];
    }
    Svd3 sv;
    svd3_jacobi(F.cov, sv);
    F.n0 = sv.U[2];
    F.n1 = sv.U[5];
    F.n2 = sv.U[8];
    F.sv0 = sv.sv[0];
    F.sv1 = sv.sv[1];
    F.sv2 = sv.sv[2];
    float dot = F.n0 * F.mean0;
    dot = dot + F.n1 * F.mean1;
    dot = dot + F.n2 * F.mean2;
    const float d = -dot;
    F.thd = (float)(cz.th_dist - (double)d);
}

__device__ __forceinline__ void fit_finish(const DevParams& P, const Arena& A, int s, int p, int n, int zone, int ring,
                                           int concentric_idx, const FitState& F) {
    const float n0 = F.n0, n1 = F.n1, n2 = F.n2, thd = F.thd, mean0 = F.mean0, mean1 = F.mean1, mean2 = F.mean2;
    const float sv0 = F.sv0, sv1 = F.sv1, sv2 = F.sv2;
    // ---- gating (patchwork.h:339-384) ----
    int status;
    {
        const double ground_z_vec = (double)fabs_f(n2);
        const double ground_z_elevation = (double)mean2;
        float svmin = sv0;
        if (sv1 < svmin) svmin = sv1;
        if (sv2 < svmin) svmin = sv2;
        const double surface_variable = (double)(svmin / (sv0 + sv1 + sv2));
        if (ground_z_vec < P.czm.uprightness_thr) {
            status = 2;
        } else if (concentric_idx < P.czm.num_rings_of_interest) {
            const int e = ring + 2 * zone;
            if (ground_z_elevation > P.czm.elevation_thr[e])
                status = (P.czm.flatness_thr[e] > surface_variable) ? 1 : 3;
            else
                status = 1;
        } else {
            status = 1;
        }
    }
    scvod_patch_plane pl;
    pl.normal[0] = n0;
    pl.normal[1] = n1;
    pl.normal[2] = n2;
    pl.mean[0] = mean0;
    pl.mean[1] = mean1;
    pl.mean[2] = mean2;
    pl.sv[0] = sv0;
    pl.sv[1] = sv1;
    pl.sv[2] = sv2;
    pl.n_pts = n;
    pl.n_ground = 0;  // filled by k_pw_arrange
    pl.status = status;
    A.planes[s * kMaxPatches + p] = pl;
    A.fit_thd[s * kMaxPatches + p] = thd;
}

// Every lane owns one patch (the fp32 sums of the fit are strictly sequential, so a patch cannot be
// split across lanes), but the wave READS cooperatively: per step it copies the next kFitCh points of
// each of its 64 patches into LDS with contiguous 192-byte runs per patch, and every lane then consumes
// its own row.  A lane-per-patch walk straight from global memory touches 64 different cache lines per
// load instruction and was bound by the L1/TA, not by HBM.
constexpr int kFitCh = 16;               // points per patch per step
constexpr int kFitRow = kFitCh * 3;      // dwords per patch per step
constexpr int kFitQuads = kFitRow / 4;   // 16-byte quads per row (12)
constexpr int kFitStride = kFitRow + 4;  // row stride 52 dwords: 16-byte aligned rows, b128 row accesses of 8
                                         // consecutive lanes cover all 32 banks once
constexpr int kFitLoads = kFitQuads;     // 64 lanes x 12 quad loads cover 64 rows x 12 quads

typedef float fitq_mem __attribute__((ext_vector_type(4), aligned(4)));  // rows start on any point boundary
typedef float fitq __attribute__((ext_vector_type(4)));

struct FitTile {
    fitq row[64 * kFitStride / 4];
    uint32_t start[64];  // first point of the lane's patch in sorted_xyz (units of points)
    int n[64];
};

// fetch step `c` of all 64 patches into registers (12 quads per lane): load j covers quad f = j * 64 + lane
// = row f / 12, quad f % 12.  A quad is fetched when its first dword belongs to the patch; its tail may run up to
// 12 bytes into whatever follows (never consumed: the consumers test the point index).
__device__ __forceinline__ void fit_fetch(const FitTile& T, const float* __restrict__ src, int c, int lane, fitq (&r)[kFitLoads]) {
#pragma unroll
    for (int j = 0; j < kFitLoads; ++j) {
        const int f = j * 64 + lane;
        const int pp = f / kFitQuads;
        const int w = (f - pp * kFitQuads) * 4;
        const int pt = c * kFitCh + w / 3;
        fitq v = {0.f, 0.f, 0.f, 0.f};
        if (pt < T.n[pp]) v = *(const fitq_mem*)(src + ((size_t)T.start[pp] * 3 + (size_t)(c * kFitRow + w)));
        r[j] = v;
    }
}
__device__ __forceinline__ void fit_store(FitTile& T, int lane, const fitq (&r)[kFitLoads]) {
#pragma unroll
    for (int j = 0; j < kFitLoads; ++j) {
        const int f = j * 64 + lane;
        const int pp = f / kFitQuads;
        const int w4 = f - pp * kFitQuads;
        T.row[pp * (kFitStride / 4) + w4] = r[j];
    }
}

__global__ __launch_bounds__(64) void k_pw_fit(DevParams P, Arena A, int coop_class) {
    __shared__ FitTile T;
    const int lane = threadIdx.x;
    int lo, hi;
    order_range(A.order_off, 0, coop_class - 1, lo, hi);  // the larger patches go to k_pw_fit_coop
    const int t = lo + blockIdx.x * 64 + lane;
    if (lo + blockIdx.x * 64 >= hi) return;
    const bool live = t < hi;
    const int4 item = live ? A.order[t] : make_int4(0, 0, 0, 0);
    const int code = item.x;
    const int s = code / kMaxPatches, p = code - s * kMaxPatches;
    const int n = live ? item.y : 0;
    T.start[lane] = (uint32_t)(item.z + item.w);
    T.n[lane] = n;
    int n_max = n;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) n_max = max(n_max, __shfl_xor(n_max, d));
    const int n_steps = (n_max + kFitCh - 1) / kFitCh;
    const float* __restrict__ src = (const float*)A.sorted_xyz;
    const fitq* myq = T.row + lane * (kFitStride / 4);
    __syncthreads();

    int zone = 0;
    while (zone < 3 && p >= P.czm.patch_base[zone + 1]) ++zone;
    const int ring = (p - P.czm.patch_base[zone]) / P.czm.num_sectors[zone];
    int concentric_idx = ring;
    for (int k = 0; k < zone; ++k) concentric_idx += P.czm.num_rings[k];

    fitq regs[kFitLoads];
    float my[kFitRow];

    // ---- extract_initial_seeds_ (patchwork.h:235-268): skip the too-low prefix (zone 0), mean z of the
    // next num_lpr points ----
    double sum = 0;
    int cnt = 0;
    {
        bool skipping = (zone == 0);
        bool busy = n > 0;
        fit_fetch(T, src, 0, lane, regs);
        for (int c = 0; c < n_steps; ++c) {
            fit_store(T, lane, regs);
            __syncthreads();
            if (c + 1 < n_steps) fit_fetch(T, src, c + 1, lane, regs);
            if (busy) {
#pragma unroll
                for (int g = 0; g < kFitQuads; ++g) {
                    const fitq v = myq[g];
                    my[4 * g] = v.x;
                    my[4 * g + 1] = v.y;
                    my[4 * g + 2] = v.z;
                    my[4 * g + 3] = v.w;
                }
#pragma unroll
                for (int k = 0; k < kFitCh; ++k) {
                    const float z = my[3 * k + 2];
                    const bool valid = (c * kFitCh + k < n);
                    if (skipping && !(valid && (double)z < P.czm.seed_margin_z)) skipping = false;
                    if (valid && !skipping && cnt < P.czm.num_lpr) {
                        sum += (double)z;
                        ++cnt;
                    }
                }
                if ((!skipping && cnt >= P.czm.num_lpr) || (c + 1) * kFitCh >= n) busy = false;
            }
            __syncthreads();
            if (!__any(busy)) break;
        }
    }
    const double lpr = cnt != 0 ? sum / cnt : 0.0;
    const double seed_thr = lpr + P.czm.th_seeds;

    FitState F;
    fit_state_init(F);

    for (int iter = 0; iter < P.czm.num_iter; ++iter) {
        const float n0 = F.n0, n1 = F.n1, n2 = F.n2, thd = F.thd;
        // pcl::computeMeanAndCovarianceMatrix over the current ground set, strictly in z order.  Membership
        // is a select; adding +0.0f is exact here because the accumulators can never be -0.0f.
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f, a8 = 0.f;
        int m = 0;
        bool busy = n > 0;
        fit_fetch(T, src, 0, lane, regs);
        for (int c = 0; c < n_steps; ++c) {
            fit_store(T, lane, regs);
            __syncthreads();
            if (c + 1 < n_steps) fit_fetch(T, src, c + 1, lane, regs);  // in flight while this step is summed
            if (busy) {
#pragma unroll
                for (int g = 0; g < kFitQuads; ++g) {
                    const fitq v = myq[g];
                    my[4 * g] = v.x;
                    my[4 * g + 1] = v.y;
                    my[4 * g + 2] = v.z;
                    my[4 * g + 3] = v.w;
                }
#pragma unroll
                for (int k = 0; k < kFitCh; ++k) {
                    Xyz q;
                    q.x = my[3 * k];
                    q.y = my[3 * k + 1];
                    q.z = my[3 * k + 2];
                    bool in = (c * kFitCh + k < n);
                    if (iter == 0)
                        in = in && ((double)q.z < seed_thr);
                    else
                        in = in && (plane_res(q, n0, n1, n2) < thd);
                    const float t0 = q.x * q.x, t1 = q.x * q.y, t2 = q.x * q.z, t3 = q.y * q.y, t4 = q.y * q.z,
                                t5 = q.z * q.z;
                    a0 += in ? t0 : 0.f;
                    a1 += in ? t1 : 0.f;
                    a2 += in ? t2 : 0.f;
                    a3 += in ? t3 : 0.f;
                    a4 += in ? t4 : 0.f;
                    a5 += in ? t5 : 0.f;
                    a6 += in ? q.x : 0.f;
                    a7 += in ? q.y : 0.f;
                    a8 += in ? q.z : 0.f;
                    m += in ? 1 : 0;
                }
                if ((c + 1) * kFitCh >= n) busy = false;
                // seeds are a prefix of the z-sorted patch: once the last point of a step fails, the rest fail
                if (iter == 0 && busy && !((double)my[3 * (kFitCh - 1) + 2] < seed_thr)) busy = false;
            }
            __syncthreads();
            if (!__any(busy)) break;
        }
        fit_update(F, P.czm, a0, a1, a2, a3, a4, a5, a6, a7, a8, m);
    }
    if (!live) return;

    fit_finish(P, A, s, p, n, zone, ring, concentric_idx, F);
}

// Patches of 512 points or more (class kClassFitCoop): a lane per patch leaves the chip almost empty (a K64 scan has ~40 such
// patches holding 80 % of its points) and runs each of them as one dependent chain thousands of points long.  Here
// 16 lanes share a patch: each step they test 16 points and form the 9 products in parallel, pass them through LDS
// transposed, and lanes 0..8 of the group add "their" accumulator over the 16 points IN ORDER -- the sums are the
// same sequential fp32 sums, only the nine independent chains run on nine lanes instead of one.
// GL lanes share a patch, 64 / GL patches per wave.  GL = 16 for sequence shards (throughput: four patches keep the
// nine adding lanes of each busy); GL = 64 for a handful of scans (latency: the parallel part of the largest patch takes
// a quarter of the steps, only its sequential adds remain a chain).
template <int GL>
__global__ __launch_bounds__(64) void k_pw_fit_coop(DevParams P, Arena A, int coop_class) {
    constexpr int NG = 64 / GL;
    constexpr int ROW = GL + 4;             // floats per product row: padded so the nine row reads spread over the banks
    constexpr int PF = (GL == 16) ? 8 : 2;  // steps in flight per lane (128 points of the patch either way)
    constexpr unsigned long long FULL = (GL == 64) ? ~0ull : ((1ull << (GL & 63)) - 1ull);
    __shared__ fitq tile[NG * 9 * ROW / 4];  // [group][accumulator][GL points]
    const int lane = threadIdx.x, g = lane / GL, r = lane % GL, gbase = g * GL;
    auto group_bits = [&](unsigned long long ballot) -> unsigned long long { return (ballot >> gbase) & FULL; };
    int lo, hi;
    order_range(A.order_off, coop_class, 63, lo, hi);
    if (lo + (int)blockIdx.x * NG >= hi) return;
    const int w = lo + blockIdx.x * NG + g;
    const bool live = w < hi;
    const int4 item = live ? A.order[w] : make_int4(0, 0, 0, 0);
    const int code = item.x;
    const int s = code / kMaxPatches, p = code - s * kMaxPatches;
    const int n = live ? item.y : 0;
    const Xyz* __restrict__ sp = A.sorted_xyz + (size_t)item.z + item.w;
    int n_max = n;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) n_max = max(n_max, __shfl_xor(n_max, d));
    const int n_blocks = (n_max + GL * PF - 1) / (GL * PF);

    int zone = 0;
    while (zone < 3 && p >= P.czm.patch_base[zone + 1]) ++zone;
    const int ring = (p - P.czm.patch_base[zone]) / P.czm.num_sectors[zone];
    int concentric_idx = ring;
    for (int k = 0; k < zone; ++k) concentric_idx += P.czm.num_rings[k];

    // ---- extract_initial_seeds_ (patchwork.h:235-268) ----
    int init_idx = 0;
    {
        bool searching = live && (zone == 0);
        for (int c = 0; __any(searching); ++c) {
            if (searching) {
                const int j = c * GL + r;
                const bool low = (j < n) && ((double)sp[j].z < P.czm.seed_margin_z);
                const unsigned long long mg = group_bits(__ballot(low));
                if (mg != FULL) {
                    init_idx = c * GL + __builtin_ctzll(~mg);
                    searching = false;
                }
            }
        }
    }
    double sum = 0;
    int cnt = 0;
    for (int i0 = 0; i0 < P.czm.num_lpr; i0 += GL) {
        const int j = init_idx + i0 + r;
        const float zc = (j < n) ? sp[j].z : 0.f;
        const int kmax = min(GL, P.czm.num_lpr - i0);
        for (int k = 0; k < kmax; ++k) {
            const float z = __shfl(zc, gbase + k);
            if (init_idx + i0 + k < n) {
                sum += (double)z;
                ++cnt;
            }
        }
    }
    const double lpr = cnt != 0 ? sum / cnt : 0.0;
    const double seed_thr = lpr + P.czm.th_seeds;

    FitState F;
    fit_state_init(F);
    const int n_last = max(n - 1, 0);
    float* tf = (float*)tile;
    const int acc_row = (g * 9 + (r < 9 ? r : 8)) * (ROW / 4);  // lanes 9.. shadow accumulator 8

    for (int iter = 0; iter < P.czm.num_iter; ++iter) {
        const float n0 = F.n0, n1 = F.n1, n2 = F.n2, thd = F.thd;
        float acc = 0.f;
        int m_lane = 0;
        bool busy = n > 0;
        Xyz cur[PF], nxt[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) cur[k] = sp[min(k * GL + r, n_last)];
        for (int b = 0; b < n_blocks; ++b) {
            const int j0 = b * GL * PF;
            // unconditional (clamped) loads: a load under a branch would have to land before the branch joins, which
            // serialises the whole block behind one memory latency
#pragma unroll
            for (int k = 0; k < PF; ++k) nxt[k] = sp[min(j0 + (PF + k) * GL + r, n_last)];
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const Xyz q = cur[k];
                const int j = j0 + k * GL + r;
                bool in = busy && (j < n);
                bool fails = false;
                if (iter == 0) {
                    fails = in && !((double)q.z < seed_thr);
                    in = in && !fails;
                } else {
                    in = in && (plane_res(q, n0, n1, n2) < thd);
                }
                // non-members contribute +0.0f to every sum: zero the point, the products follow (0 * 0 = +0)
                const float zx = in ? q.x : 0.f, zy = in ? q.y : 0.f, zz = in ? q.z : 0.f;
                float* col = tf + g * 9 * ROW + r;
                col[0 * ROW] = zx * zx;
                col[1 * ROW] = zx * zy;
                col[2 * ROW] = zx * zz;
                col[3 * ROW] = zy * zy;
                col[4 * ROW] = zy * zz;
                col[5 * ROW] = zz * zz;
                col[6 * ROW] = zx;
                col[7 * ROW] = zy;
                col[8 * ROW] = zz;
                m_lane += in ? 1 : 0;
                // seeds are a prefix of the z-sorted patch: the group stops after the step in which one fails
                const bool stop = (iter == 0) && (group_bits(__ballot(fails)) != 0ull);
                // the LDS unit executes one wave's instructions in order, so the rows written above are what the reads
                // below see; only the compiler has to be kept from reordering them
                __builtin_amdgcn_wave_barrier();
                // adding +0.0f for non-members is exact: the accumulators can never be -0.0f
#pragma unroll
                for (int q4 = 0; q4 < GL / 4; ++q4) {
                    const fitq v = tile[acc_row + q4];
                    acc += v.x;
                    acc += v.y;
                    acc += v.z;
                    acc += v.w;
                }
                __builtin_amdgcn_wave_barrier();
                if (stop || j0 + (k + 1) * GL >= n) busy = false;
            }
#pragma unroll
            for (int k = 0; k < PF; ++k) cur[k] = nxt[k];
            if (!__any(busy)) break;
        }
        int m = m_lane;  // members seen by this lane -> members of the group's patch
#pragma unroll
        for (int d = GL / 2; d > 0; d >>= 1) m += __shfl_xor(m, d);
        const float a0 = __shfl(acc, gbase + 0), a1 = __shfl(acc, gbase + 1), a2 = __shfl(acc, gbase + 2),
                    a3 = __shfl(acc, gbase + 3), a4 = __shfl(acc, gbase + 4), a5 = __shfl(acc, gbase + 5),
                    a6 = __shfl(acc, gbase + 6), a7 = __shfl(acc, gbase + 7), a8 = __shfl(acc, gbase + 8);
        fit_update(F, P.czm, a0, a1, a2, a3, a4, a5, a6, a7, a8, m);
    }
    if (live && r == 0) fit_finish(P, A, s, p, n, zone, ring, concentric_idx, F);
}


// one wave per (scan, patch): final plane test of every point (patchwork.h:488-501), keeps the
// z order inside the ground part and the non-ground part, counts what k_emit_offsets needs.
// GL lanes per patch: 64 for patches of 64 points or more, 16 (four patches per wave) for the many smaller ones, whose
// cost is the per-patch latency chain, not the points.
template <int GL, int C_LO, int C_HI>
__global__ __launch_bounds__(256) void k_pw_arrange(DevParams P, Arena A) {
    constexpr int NG = 64 / GL;
    constexpr unsigned long long FULL = (GL == 64) ? ~0ull : ((1ull << (GL & 63)) - 1ull);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g_ = lane / GL, r = lane % GL, gbase = g_ * GL;
    auto group_bits = [&](unsigned long long ballot) -> unsigned long long { return (ballot >> gbase) & FULL; };
    int lo, total;
    order_range(A.order_off, C_LO, C_HI, lo, total);
    if (lo >= total) return;
    const int stride = gridDim.x * 4 * NG;
    int wi = lo + (blockIdx.x * 4 + wave) * NG + g_;
    if (lo + (int)(blockIdx.x * 4 + wave) * NG >= total) return;
    // Half of the live patches hold fewer than 64 points: a wave would spend its time in the dependent chain
    // order[] -> plane -> points.  So the chain is software-pipelined ACROSS patches: the list item two patches ahead
    // and the plane + first points of the next patch are in flight while the current patch is arranged (all loads
    // unconditional with clamped indices, so none has to land before a branch joins).
    struct Head {
        float n0, n1, n2, thd;
        int status;
        Xyz q;
        uint32_t w;
    };
    auto load_item = [&](int k) -> int4 { return A.order[min(k, total - 1)]; };
    auto load_head = [&](const int4& it) -> Head {
        const int slot = it.x;  // scan * kMaxPatches + patch
        const size_t first = (size_t)it.z + it.w + min(r, it.y - 1);
        Head h;
        h.n0 = A.planes[slot].normal[0];
        h.n1 = A.planes[slot].normal[1];
        h.n2 = A.planes[slot].normal[2];
        h.status = A.planes[slot].status;
        h.thd = A.fit_thd[slot];
        h.q = A.sorted_xyz[first];
        h.w = A.sorted_idx[first];
        return h;
    };
    int4 item = load_item(wi), item1 = load_item(wi + stride);
    Head head = load_head(item);
    for (; __any(wi < total); wi += stride) {
        const int4 item2 = load_item(wi + 2 * stride);
        const Head head1 = load_head(item1);
        const bool live = wi < total;  // uniform inside a group
        const int code = item.x;
        const int s = code / kMaxPatches, p = code - s * kMaxPatches;
        const int n = live ? item.y : 0, base = item.z, off = item.w;
        const int n_last = max(n - 1, 0);
        const Xyz* __restrict__ sp = A.sorted_xyz + (size_t)base + off;
        const uint32_t* __restrict__ si = A.sorted_idx + (size_t)base + off;
        const float n0 = head.n0, n1 = head.n1, n2 = head.n2;
        const float thd = head.thd;
        const int status = head.status;
        const bool rejected = (status >= 2);
        int n_max = n;
        if (NG > 1) {
#pragma unroll
            for (int d = 32; d >= GL; d >>= 1) n_max = max(n_max, __shfl_xor(n_max, d));
        }
        // ONE pass: ground part grows from the front in z order, the non-ground part from the back
        // (element r of the non-ground part lives at seg[n - 1 - r]; k_emit reads it that way)
        int n_g = 0, n_ng = 0, a_g = 0, a_ng = 0;
        uint32_t* seg = A.seg + (size_t)base + off;
        // the next GL points are in flight while this step is classified (clamped, unconditional loads)
        Xyz q_next = head.q;
        uint32_t w_next = head.w;
        const unsigned long long below = (1ull << r) - 1ull;
        for (int j0 = 0; j0 < n_max; j0 += GL) {
            const int j = j0 + r;
            int g = 0, keep = 0;
            const Xyz q = q_next;
            uint32_t w = w_next;
            q_next = sp[min(j + GL, n_last)];
            w_next = si[min(j + GL, n_last)];
            if (j < n) {
                g = plane_res(q, n0, n1, n2) < thd;
                // range/FOV verdict of makeApriVec, only for points that reach the non-ground stream
                if (!g || rejected) {
                    keep = keep_of_point(P.bin, P.keep, q.x, q.y, q.z);
                    w |= keep ? 0x80000000u : 0u;
                }
            }
            const unsigned long long bg = group_bits(__ballot(g));
            const unsigned long long bn = group_bits(__ballot(!g && j < n));
            if (j < n) {
                if (g)
                    seg[n_g + __popcll(bg & below)] = w;
                else
                    seg[n - 1 - (n_ng + __popcll(bn & below))] = w;
            }
            a_g += __popcll(group_bits(__ballot(g && keep)));
            a_ng += __popcll(group_bits(__ballot(!g && keep && j < n)));
            n_g += __popcll(bg);
            n_ng += __popcll(bn);
        }
        if (live && r == 0) {
            PatchRec rec;
            rec.n = n;
            rec.n_g = n_g;
            rec.status = status;
            rec.a_g = a_g;
            rec.a_ng = a_ng;
            A.patch_rec[s * kMaxPatches + p] = rec;
            A.planes[s * kMaxPatches + p].n_ground = n_g;
        }
        item = item1;
        item1 = item2;
        head = head1;
    }
}

// ------------------------------------------------------------------------------------------
// Emission: per-scan exclusive scans over patches (reference emission order = patch order),
// then one workgroup per patch writes cloud_out / cloud_nonground indices, per-point class,
// apri_vec (ordered compaction of the non-ground stream by the range/FOV verdict) and the
// rejected list.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_emit_offsets(DevParams P, Arena A) {
    __shared__ int wsum[17];
    const int s = blockIdx.x;
    const int t = threadIdx.x;
    int eg = 0, eng = 0, ea = 0, er = 0;
    if (t < P.n_patches) {
        PatchRec r = A.patch_rec[s * kMaxPatches + t];
        if (r.status == 1) {
            eg = r.n_g;
            eng = r.n - r.n_g;
            ea = r.a_ng;
        } else if (r.status >= 2) {
            eng = r.n;
            ea = r.a_g + r.a_ng;
        }
        er = eng - ea;
    }
    int tg, tng, ta, tr;
    int xg = block_excl_scan<1024>(eg, tg, wsum);
    int xng = block_excl_scan<1024>(eng, tng, wsum);
    int xa = block_excl_scan<1024>(ea, ta, wsum);
    int xr = block_excl_scan<1024>(er, tr, wsum);
    if (t < P.n_patches) {
        int* o = A.emit_off + ((size_t)s * kMaxPatches + t) * 4;
        o[0] = xg;
        o[1] = xng;
        o[2] = xa;
        o[3] = xr;
    }
    if (t == 0) {
        int n = A.scan_off[s + 1] - A.scan_off[s];
        int* c = A.counts + s * 8;
        c[0] = n;
        c[1] = tg;
        c[2] = tng;
        c[3] = n - tg - tng;
        c[4] = ta;
        c[5] = tr;
        c[6] = 0;
        c[7] = P.n_patches;
        A.scan_irr[s] = 0;
    }
}

constexpr int kEmitThreads = 256;
__global__ __launch_bounds__(kEmitThreads) void k_emit(DevParams P, Arena A) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int total = A.order_off[64];
    if ((int)(blockIdx.x * 4 + wave) >= total) return;
    // the dependent chain list item -> patch record / output offsets -> seg words is software-pipelined ACROSS patches (most
    // hold a few hundred points): the item two patches ahead and the record + offsets of the next patch are in flight while
    // the current patch is emitted (clamped, unconditional loads)
    const int stride = gridDim.x * 4;
    struct Meta {
        PatchRec r;
        int4 o;
    };
    auto load_item = [&](int k) -> int4 { return A.order[min(k, total - 1)]; };
    auto load_meta = [&](const int4& it) -> Meta {
        Meta m;
        m.r = A.patch_rec[it.x];  // it.x = scan * kMaxPatches + patch
        m.o = *reinterpret_cast<const int4*>(A.emit_off + (size_t)it.x * 4);
        return m;
    };
    int w = blockIdx.x * 4 + wave;
    int4 item = load_item(w), item1 = load_item(w + stride);
    Meta meta = load_meta(item);
    for (; w < total; w += stride) {
        const int4 item2 = load_item(w + 2 * stride);
        const Meta meta1 = load_meta(item1);
        const int code = item.x;
        const int s = code / kMaxPatches, p = code - s * kMaxPatches;
        const PatchRec r = meta.r;
        const int base = item.z, off = item.w;
        const int xg = meta.o.x, xng = meta.o.y, xa = meta.o.z, xr = meta.o.w;
        const bool kept = (r.status == 1);
        const uint32_t* seg = A.seg + (size_t)base + off;
        // ground part of a kept patch -> cloud_out
        if (kept) {
            for (int e = lane; e < r.n_g; e += 64) {
                const uint32_t id = seg[e] & 0x7fffffffu;
                A.ground_idx[(size_t)base + xg + e] = (int32_t)id;
            }
        }
        // non-ground stream of this patch: for a rejected patch the ground part (front, ascending) followed
        // by the non-ground part; the latter is stored back to front by k_pw_arrange
        const int e0 = kept ? r.n_g : 0;
        int run_keep = 0;
        auto seg_at = [&](int e) -> uint32_t {  // clamped, unconditional load (callers test e < r.n)
            const int k = (e < r.n_g) ? e : r.n - 1 - (e - r.n_g);
            return seg[min(max(k, 0), r.n - 1)];
        };
        // two steps of seg words and one step of point gathers are in flight while a step is emitted; lanes that keep
        // nothing gather the scan's first point (one shared line) so the load needs no branch
        auto gather = [&](uint32_t v, int e) -> float4 {
            const bool k = (e < r.n) && (v >> 31);
            return A.pts[base + (k ? (v & 0x7fffffffu) : 0u)];
        };
        uint32_t v0 = seg_at(e0 + lane), v1 = seg_at(e0 + 64 + lane);
        float4 q0 = gather(v0, e0 + lane);
        for (int c0 = e0; c0 < r.n; c0 += 64) {
            const int e = c0 + lane;
            const uint32_t v = v0;
            const float4 q = q0;
            const uint32_t v2 = seg_at(e + 128);
            q0 = gather(v1, e + 64);
            v0 = v1;
            v1 = v2;
            const int keep = (e < r.n) ? (int)(v >> 31) : 0;
            const unsigned long long bk = __ballot(keep);
            const int ek = __popcll(bk & ((1ull << lane) - 1ull));
            const int nk = __popcll(bk);
            const size_t dst0 = (size_t)base + xa + run_keep;  // first PointAPRI slot of this step
            if (e < r.n) {
                const uint32_t id = v & 0x7fffffffu;
                const int spos = e - e0;  // position in the non-ground stream of this patch
                A.nonground_idx[(size_t)base + xng + spos] = (int32_t)id;
                if (keep) {
                    // apri_vec is kept in its compact form (source index, voxel key, intensity, index triple); the 44-byte
                    // PointAPRI records are expanded from it on request (k_apri_expand).  Only the indices are needed here: the
                    // guarded estimate decides them away from the bin edges, the reference arithmetic next to one
                    int32_t ri, si, ai;
                    if (!idx3_fast(P.bin, P.binfast, q.x, q.y, q.z, &ri, &si, &ai)) {
                        Apri a;
                        apri_of_point(P.bin, q.x, q.y, q.z, q.w, a);
                        ri = a.range_idx;
                        si = a.sector_idx;
                        ai = a.azimuth_idx;
                    }
                    A.apri_src[dst0 + ek] = (int32_t)id;
                    A.apri_key[dst0 + ek] = ai * P.bin.range_num * P.bin.sector_num + ri * P.bin.sector_num + si;
                    A.apri_int[dst0 + ek] = q.w;
                    A.apri_idx3[dst0 + ek] = pack_idx3(ri, si, ai);
                    if ((unsigned)ri >= (unsigned)P.bin.range_num || (unsigned)si >= (unsigned)P.bin.sector_num || (unsigned)ai >= (unsigned)P.bin.azimuth_num)
                        A.scan_irr[s] = 1;  // (rare: a -1 bin; every writer stores the same value)
                } else {
                    A.rejected_src[(size_t)base + xr + (spos - (run_keep + ek))] = (int32_t)id;
                }
            }
            run_keep += nk;
        }
        item = item1;
        item1 = item2;
        meta = meta1;
    }
}

// ------------------------------------------------------------------------------------------
// the loader step in front of the hot path -- label filter + intensity scaling of SSC::getCloud
// (ssc.cpp:1063-1076) and pcl::VoxelGrid<PointXYZI> (ssc.cpp:1103-1106; PCL 1.8.1 applyFilter).  The cell index of
// every kept point becomes a key of the voxel stage above (same bucket + LDS sort machinery, keys ascending, point
// indices ascending inside a cell = the canonical order), the centroid kernel then walks each cell's point list
// sequentially (CentroidPoint's fp32 running sums).  Filtered points get the key INT_MAX and form one trailing cell
// that is dropped.
// ------------------------------------------------------------------------------------------
constexpr int32_t kVgDropped = 0x7fffffff;
__device__ __forceinline__ bool vg_kept(const VgJob& J, int gi) {
    if (!J.labels) return true;
    const uint32_t l = J.labels[gi] & 0xFFFFu;
    return !(l == 0u || l == 1u);
}

// one workgroup per scan: getMinMax3D over the kept points, then PCL's bounding box / divisions / overflow test
__global__ __launch_bounds__(1024) void k_vg_minmax(Arena A, VgJob J) {
    __shared__ float red[6][16];
    __shared__ int cnt[16];
    const int s = blockIdx.x;
    const int base = A.scan_off[s];
    const int n = A.scan_off[s + 1] - base;
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    int kept = 0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        if (!vg_kept(J, base + i)) continue;
        const float4 p = A.pts[base + i];
        mn[0] = p.x < mn[0] ? p.x : mn[0];
        mn[1] = p.y < mn[1] ? p.y : mn[1];
        mn[2] = p.z < mn[2] ? p.z : mn[2];
        mx[0] = p.x > mx[0] ? p.x : mx[0];
        mx[1] = p.y > mx[1] ? p.y : mx[1];
        mx[2] = p.z > mx[2] ? p.z : mx[2];
        ++kept;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float o = __shfl_xor(mn[a], d), q = __shfl_xor(mx[a], d);
            mn[a] = o < mn[a] ? o : mn[a];
            mx[a] = q > mx[a] ? q : mx[a];
        }
        kept += __shfl_xor(kept, d);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        for (int a = 0; a < 3; ++a) {
            red[a][wave] = mn[a];
            red[3 + a][wave] = mx[a];
        }
        cnt[wave] = kept;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        kept = 0;
        for (int w = 0; w < 16; ++w) {
            for (int a = 0; a < 3; ++a) {
                mn[a] = red[a][w] < mn[a] ? red[a][w] : mn[a];
                mx[a] = red[3 + a][w] > mx[a] ? red[3 + a][w] : mx[a];
            }
            kept += cnt[w];
        }
        int32_t* par = A.vg_par + s * 16;
        int overflow = 0;
        int min_b[3] = {0, 0, 0}, div_b[3] = {1, 1, 1};
        long long range = 1;
        if (kept > 0) {
            const long long dx = (long long)((mx[0] - mn[0]) * J.inv_leaf[0]) + 1;
            const long long dy = (long long)((mx[1] - mn[1]) * J.inv_leaf[1]) + 1;
            const long long dz = (long long)((mx[2] - mn[2]) * J.inv_leaf[2]) + 1;
            overflow = (dx * dy * dz > 2147483647ll) ? 1 : 0;
            for (int a = 0; a < 3; ++a) {
                min_b[a] = (int)floor_f(mn[a] * J.inv_leaf[a]);
                const int max_b = (int)floor_f(mx[a] * J.inv_leaf[a]);
                div_b[a] = max_b - min_b[a] + 1;
            }
            range = overflow ? (long long)n : (long long)div_b[0] * div_b[1] * div_b[2];
            if (range > 2147483646ll) range = 2147483646ll;
        }
        par[0] = min_b[0];
        par[1] = min_b[1];
        par[2] = min_b[2];
        par[3] = 1;
        par[4] = div_b[0];
        par[5] = div_b[0] * div_b[1];
        par[6] = overflow;
        par[7] = kept;
        atomicMax(A.vg_range, (int)range);
        A.counts[s * 8 + 4] = n;  // the voxel stage sorts every input point (dropped ones under kVgDropped)
    }
}

__global__ __launch_bounds__(256) void k_vg_keys(Arena A, VgJob J) {
    const int s = blockIdx.y;
    const int base = A.scan_off[s];
    const int n = A.scan_off[s + 1] - base;
    const int32_t* par = A.vg_par + s * 16;
    const int mb0 = par[0], mb1 = par[1], mb2 = par[2], m1 = par[4], m2 = par[5], overflow = par[6];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        int32_t key = kVgDropped;
        if (vg_kept(J, base + i)) {
            if (overflow) {
                key = i;  // PCL returns the input cloud: every point its own cell, input order
            } else {
                const float4 p = A.pts[base + i];
                const int i0 = (int)(floor_f(p.x * J.inv_leaf[0]) - (float)mb0);
                const int i1 = (int)(floor_f(p.y * J.inv_leaf[1]) - (float)mb1);
                const int i2 = (int)(floor_f(p.z * J.inv_leaf[2]) - (float)mb2);
                key = i0 + i1 * m1 + i2 * m2;
            }
        }
        A.apri_key[(size_t)base + i] = key;
    }
}

// Cell indices are anything but uniform (the ground layers near the sensor hold most of a scan), so equal index ranges
// make buckets of tens of thousands of points.  One workgroup per scan histograms the keys into kVgLutBins equal
// ranges (≈ a 1 m band of one z layer for a KITTI scan at 0.08 m) and cuts the running count into <= 1022 buckets of about equal population; the table is monotone, so bucket
// order is still key order.
__global__ __launch_bounds__(1024) void k_vg_lut(Arena A, int32_t* shift_out) {
    extern __shared__ int hist[];  // kVgLutBins counters (64 KB)
    __shared__ int wsum[17];
    const int s = blockIdx.x;
    const int base = A.scan_off[s];
    const int n = A.scan_off[s + 1] - base;
    // bin width from the largest cell-index range of the batch (every workgroup derives the same value)
    const long long range = A.vg_range[0];
    int shift = 0;
    while ((range >> shift) > kVgLutBins - 1) ++shift;
    if (s == 0 && threadIdx.x == 0) *shift_out = shift;
    for (int b = threadIdx.x; b < kVgLutBins; b += 1024) hist[b] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int32_t key = A.apri_key[(size_t)base + i];
        if (key != kVgDropped) atomicAdd(&hist[key >> shift], 1);
    }
    __syncthreads();
    const int kept = A.vg_par[s * 16 + 7];
    const int target = max(kept / (kMaxBuckets - 2) + 1, 1536);  // ~1.5 k points per bucket: the 2048-key LDS tier
    int run = 0;
    for (int b0 = 0; b0 < kVgLutBins; b0 += 1024) {
        const int c = hist[b0 + threadIdx.x];
        int total;
        const int ex = block_excl_scan<1024>(c, total, wsum);
        const int bucket = min((run + ex) / target, kMaxBuckets - 2);
        A.vb_lut[(size_t)s * kVgLutBins + b0 + threadIdx.x] = (uint16_t)bucket;
        run += total;
        __syncthreads();
    }
}

// output offsets = exclusive scan over the scans of their cell counts (one workgroup)
__global__ __launch_bounds__(1024) void k_vg_outoff(Arena A) {
    __shared__ int wsum[17];
    int run = 0;
    for (int s0 = 0; s0 < A.n_scans; s0 += 1024) {
        const int s = s0 + threadIdx.x;
        const int c = (s < A.n_scans) ? A.counts[s * 8 + 6] : 0;
        int total;
        const int ex = block_excl_scan<1024>(c, total, wsum);
        if (s < A.n_scans) A.vg_outoff[s] = run + ex;
        run += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) A.vg_outoff[A.n_scans] = run;
}

// compaction of the per-bucket centroid runs into the caller's buffer (ascending cell index = bucket order)
__global__ __launch_bounds__(256) void k_vg_final(Arena A, VgJob J, long long out_capacity) {
    const int b = blockIdx.x, s = blockIdx.y;
    const int nv = A.vb_nvox[s * kMaxBuckets + b];
    if (nv == 0) return;
    const int base = A.scan_off[s];
    const int src = A.vb_off[s * (kMaxBuckets + 1) + b];
    const long long dst = (long long)A.vg_outoff[s] + A.vox_off[s * (kMaxBuckets + 1) + b];
    const float4* tmp = (const float4*)A.apri;
    for (int v = threadIdx.x; v < nv; v += 256)
        if (dst + v < out_capacity) J.out[dst + v] = tmp[(size_t)base + src + v];
}

// PointAPRI records (ssc.cpp:176-193) of scans [s0, s0 + gridDim.y), rebuilt from the compact apri_vec: the same spec
// function on the same point gives the same bits k_emit saw when it derived key and intensity.
__global__ __launch_bounds__(256) void k_apri_expand(DevParams P, Arena A, int s0) {
    const int s = s0 + blockIdx.y;
    const int base = A.scan_off[s];
    const int n = A.counts[s * 8 + 4];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float4 q = A.pts[base + A.apri_src[(size_t)base + i]];
        Apri a;
        apri_of_point(P.bin, q.x, q.y, q.z, q.w, a);
        scvod_apri out;
        out.x = a.x;
        out.y = a.y;
        out.z = a.z;
        out.range = a.range;
        out.angle = a.angle;
        out.azimuth = a.azimuth;
        out.intensity = a.intensity;
        out.range_idx = a.range_idx;
        out.sector_idx = a.sector_idx;
        out.azimuth_idx = a.azimuth_idx;
        out.voxel_idx = a.voxel_idx;
        A.apri[(size_t)base + i] = out;
    }
}

// Per-point class array of ONE scan, built on request from the two index lists (the reference holds the two clouds,
// never a class array; materialising it for every scan of a batch cost 0.9 ms per sequence in byte scatters).
__global__ __launch_bounds__(256) void k_cls_from_lists(Arena A, int s) {
    const int base = A.scan_off[s];
    const int n_g = A.counts[s * 8 + 1], n_ng = A.counts[s * 8 + 2];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_g + n_ng; i += gridDim.x * 256) {
        if (i < n_g)
            A.cls[base + A.ground_idx[(size_t)base + i]] = SCVOD_CLS_GROUND;
        else
            A.cls[base + A.nonground_idx[(size_t)base + (i - n_g)]] = SCVOD_CLS_NONGROUND;
    }
}

// makeApriVec on an arbitrary cloud in input order (no Patchwork): one workgroup per scan walks
// the scan in chunks, ordered compaction by block scan.  apply_filter == 0 keeps every point.
__global__ __launch_bounds__(1024) void k_bin_direct(DevParams P, Arena A, int apply_filter) {
    __shared__ int wsum[17];
    const int s = blockIdx.x;
    const int base = A.scan_off[s];
    const int n = A.scan_off[s + 1] - base;
    int run = 0;
    for (int c0 = 0; c0 < n; c0 += 1024) {
        int i = c0 + threadIdx.x;
        int keep = 0;
        Apri a;
        if (i < n) {
            float4 q = A.pts[base + i];
            keep = apri_of_point(P.bin, q.x, q.y, q.z, q.w, a);
            if (!apply_filter) keep = 1;
        }
        int tk;
        int ek = block_excl_scan<1024>(keep, tk, wsum);
        if (i < n) {
            if (keep) {
                size_t dst = (size_t)base + run + ek;
                scvod_apri out;
                out.x = a.x;
                out.y = a.y;
                out.z = a.z;
                out.range = a.range;
                out.angle = a.angle;
                out.azimuth = a.azimuth;
                out.intensity = a.intensity;
                out.range_idx = a.range_idx;
                out.sector_idx = a.sector_idx;
                out.azimuth_idx = a.azimuth_idx;
                out.voxel_idx = a.voxel_idx;
                A.apri[dst] = out;
                A.apri_src[dst] = i;
                A.apri_key[dst] = a.voxel_idx;
                A.apri_int[dst] = a.intensity;
                A.apri_idx3[dst] = pack_idx3(a.range_idx, a.sector_idx, a.azimuth_idx);
            } else {
                A.rejected_src[(size_t)base + (i - (run + ek))] = i;
            }
        }
        run += tk;
    }
    if (threadIdx.x == 0) {
        A.scan_irr[s] = 0;  // (order hint only: k_cc_scan decides regularity itself)
        int* c = A.counts + s * 8;
        c[0] = n;
        c[1] = 0;
        c[2] = 0;
        c[3] = 0;
        c[4] = run;
        c[5] = n - run;
        c[6] = 0;
        c[7] = 0;
    }
}

// apri_vec supplied by the caller (scvod_voxelize): derive the compact key / intensity arrays
__global__ __launch_bounds__(256) void k_apri_split(Arena A) {
    const int s = blockIdx.y;
    const int base = A.scan_off[s];
    const int n = A.counts[s * 8 + 4];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const scvod_apri& a = A.apri[(size_t)base + i];
        A.apri_key[(size_t)base + i] = a.voxel_idx;
        A.apri_int[(size_t)base + i] = a.intensity;
        A.apri_idx3[(size_t)base + i] = pack_idx3(a.range_idx, a.sector_idx, a.azimuth_idx);
    }
}

// ------------------------------------------------------------------------------------------
// Voxel stage (SSC::makeHashCloud): bucket by the high bits of the key, sort (key, apri idx)
// inside each bucket, per-voxel sequential intensity mean / variance.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int vx_bucket_of(const DevParams& P, const Arena& A, int s, int32_t voxel_idx) {
    if (A.vb_lut) {  // VoxelGrid run: population-balanced monotone table (cell indices are far from uniform)
        if (voxel_idx == 0x7fffffff) return kMaxBuckets - 1;
        return A.vb_lut[(size_t)s * kVgLutBins + (voxel_idx >> *A.vb_lut_shift)];
    }
    int64_t b = ((int64_t)voxel_idx + P.key_off) >> P.vb_shift;
    if (b < 0) b = 0;
    if (b > P.n_buckets - 1) b = P.n_buckets - 1;
    return (int)b;
}
__device__ __forceinline__ uint32_t vx_bias(int32_t k) { return (uint32_t)k ^ 0x80000000u; }
__device__ __forceinline__ int32_t vx_unbias(uint32_t u) { return (int32_t)(u ^ 0x80000000u); }

constexpr int kVxThreads = 256;
constexpr int kVxItems = 8;

__global__ __launch_bounds__(kVxThreads) void k_vx_count(DevParams P, Arena A) {
    __shared__ int hist[kMaxBuckets];
    const int s = blockIdx.y;
    const int base = A.scan_off[s];
    const int n = A.counts[s * 8 + 4];
    const int start = blockIdx.x * (kVxThreads * kVxItems);
    if (start >= n) return;
    for (int b = threadIdx.x; b < P.n_buckets; b += kVxThreads) hist[b] = 0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kVxItems; ++it) {
        int i = start + it * kVxThreads + threadIdx.x;
        wave_bin_add(hist, (i < n) ? vx_bucket_of(P, A, s, A.apri_key[(size_t)base + i]) : -1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < P.n_buckets; b += kVxThreads) {
        int c = hist[b];
        if (c) atomicAdd(&A.vb_count[s * kMaxBuckets + b], c);
    }
}

__global__ __launch_bounds__(1024) void k_vx_offsets(DevParams P, Arena A) {
    __shared__ int wsum[17];
    const int s = blockIdx.x;
    int c = (threadIdx.x < (unsigned)P.n_buckets) ? A.vb_count[s * kMaxBuckets + threadIdx.x] : 0;
    int total;
    int ex = block_excl_scan<1024>(c, total, wsum);
    if (threadIdx.x <= (unsigned)P.n_buckets) A.vb_off[s * (kMaxBuckets + 1) + threadIdx.x] = ex;
}

__global__ __launch_bounds__(kVxThreads) void k_vx_scatter(DevParams P, Arena A) {
    __shared__ int hist[kMaxBuckets];
    const int s = blockIdx.y;
    const int base = A.scan_off[s];
    const int n = A.counts[s * 8 + 4];
    const int start = blockIdx.x * (kVxThreads * kVxItems);
    if (start >= n) return;
    for (int b = threadIdx.x; b < P.n_buckets; b += kVxThreads) hist[b] = 0;
    __syncthreads();
    int bk[kVxItems], rank[kVxItems];
    int32_t key[kVxItems];
#pragma unroll
    for (int it = 0; it < kVxItems; ++it) {
        int i = start + it * kVxThreads + threadIdx.x;
        bk[it] = -1;
        if (i < n) {
            key[it] = A.apri_key[(size_t)base + i];
            bk[it] = vx_bucket_of(P, A, s, key[it]);
            rank[it] = atomicAdd(&hist[bk[it]], 1);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < P.n_buckets; b += kVxThreads) {
        int c = hist[b];
        if (c) hist[b] = A.vb_off[s * (kMaxBuckets + 1) + b] + atomicAdd(&A.vb_cursor[s * kMaxBuckets + b], c);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kVxItems; ++it) {
        int i = start + it * kVxThreads + threadIdx.x;
        if (bk[it] >= 0) {
            const size_t at = (size_t)base + hist[bk[it]] + rank[it];
            if (A.vx_k32)  // key relative to its bucket's first key | apri index: fits 32 bits (see vx_k32 in scvod_kernels.h)
                ((uint32_t*)A.vkeys)[at] = (uint32_t)((((int64_t)key[it] + P.key_off) & ((1ll << P.vb_shift) - 1)) << A.vx_idx_bits) | (uint32_t)i;
            else
                A.vkeys[at] = pack_key(vx_bias(key[it]), (uint32_t)i);
        }
    }
}

// MODE 0: SSC::makeHashCloud (intensity mean / variance per voxel).  MODE 1: pcl::VoxelGrid run (keys = cell indices):
// per cell the CentroidPoint sums of its points in ascending input index, straight from the sorted keys in LDS; no
// point lists, no intensity statistics; the bucket of the dropped points is skipped.
// KT = unsigned long long: (biased voxel key, apri index) under the double-encoded exponent (any key, VoxelGrid cells).
// KT = uint32_t: hot path of a filtered batch -- (key - first key of the bucket) << idx_bits | apri index; a bucket spans
// 2^vb_shift keys and a scan 2^idx_bits points, vb_shift + idx_bits <= 32 (checked on the host): half the LDS traffic, and
// the compare-exchange is a full-rate v_min_u32 + v_max_u32.
template <int CAP, int THREADS, int C_LO, int C_HI, int LGE, int MODE = 0, typename KT = unsigned long long>
__global__ __launch_bounds__(THREADS) void k_vx_bucket(DevParams P, Arena A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool K32 = sizeof(KT) == 4;
    constexpr bool PADK = true;  // both key types: unpadded 4-byte keys let the compiler fuse neighbouring LDS accesses into
                                 // wide ones that fault on the network's unaligned groups (seen on the 8192-key tier)
    constexpr int SLOTS = CAP + CAP / 8;
    KT* l_keys = (KT*)smem;   // padded layout (8-byte keys)
    int* l_vbeg = (int*)(smem + (size_t)SLOTS * 8);           // [CAP]  (the key area keeps its 8-byte size for both key types)
    float* l_int = (float*)(smem + (size_t)SLOTS * 8 + (size_t)CAP * 4);  // [CAP] intensity in sorted order
    int* wsum = (int*)(smem + (size_t)SLOTS * 8 + (size_t)CAP * 8);
    int lo, hi;
    order_range(A.vorder_off, C_LO, C_HI, lo, hi);
    constexpr bool PV = (CAP == 4096 && MODE == 0);  // (profiling build: the dominant tier is clocked)
    PROF_BEGIN();
    for (int w = lo + blockIdx.x; w < hi; w += gridDim.x) {
    if (PV) PROF_RESET();
    const int4 item = A.vorder[w];
    const int code = item.x;
    const int s = code / kMaxBuckets, b = code - s * kMaxBuckets;
    const int m = item.y;
    const int base = item.z, off = item.w;
    if (MODE == 1 && b == kMaxBuckets - 1) {  // dropped points (label filter): no cells
        if (threadIdx.x == 0) A.vb_nvox[s * kMaxBuckets + b] = 0;
        continue;
    }
    const bool in_lds = (m <= CAP);
    KT* keys;
    int* vbeg;
    float* ints;
    KT* gkeys = (KT*)A.vkeys + (size_t)base + off;
    const KT kpad = K32 ? (KT)0xffffffffu : (KT)kKeyPad;
    const uint32_t imask = (1u << A.vx_idx_bits) - 1u;
    auto k_major = [&](KT k) -> uint32_t { return K32 ? (uint32_t)k >> A.vx_idx_bits : key_major((unsigned long long)k); };
    auto k_idx = [&](KT k) -> uint32_t { return K32 ? (uint32_t)k & imask : key_idx((unsigned long long)k); };
    if (in_lds) {
        int np2 = 1 << LGE;
        while (np2 < m) np2 <<= 1;
        // a full tier: all of a thread's key loads in flight together (coalesced), sorted straight from the registers
        constexpr int IT = CAP / THREADS;
        keys = l_keys;
        vbeg = l_vbeg;
        ints = l_int;
        bool from_regs = false;
        if constexpr (IT == (1 << LGE) && CAP >= 4096) {  // (the small tiers are occupancy-bound: they keep their registers)
            if (np2 == CAP) {
                from_regs = true;
                KT tmp[IT];
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    const int j = it * THREADS + (int)threadIdx.x;
                    tmp[it] = (j < m) ? gkeys[j] : kpad;
                }
                if (PV) PROF_MARK(3, 0);
                block_bitonic_sort_pow2_regs<THREADS, PADK, LGE>(tmp, keys);
            }
        }
        if (!from_regs) {
            for (int j = threadIdx.x; j < np2; j += THREADS) l_keys[sort_slot<PADK>(j)] = (j < m) ? gkeys[j] : kpad;
            __syncthreads();
            if (PV) PROF_MARK(3, 0);
            block_bitonic_sort_pow2<THREADS, PADK, LGE>(keys, np2);
        }
        if (PV) PROF_MARK(3, 1);
    } else {
        keys = gkeys;
        vbeg = A.tmp_vox_begin + (size_t)base + off;  // rewritten below with final values
        ints = A.tmp_vox_av + (size_t)base + off;     // m >= nv entries: used as staging, rewritten below
        block_bitonic_sort<THREADS, false>(keys, m);
    }
#define KX(j) (in_lds ? sort_slot<PADK>(j) : (j))
    // head flags + compaction of voxel starts; stage the intensities in sorted order
    int run = 0;
    for (int c0 = 0; c0 < m; c0 += THREADS) {
        int j = c0 + threadIdx.x;
        int head = 0;
        if (j < m) {
            // never form keys[-1]: with flat addressing that leaves the LDS aperture
            const KT cur = keys[KX(j)];
            const KT prev = keys[KX(j > 0 ? j - 1 : 0)];
            head = (j == 0) || (k_major(cur) != k_major(prev));
            const uint32_t idx = k_idx(cur);
            if (MODE == 0) A.vox_pts[(size_t)base + off + j] = (int32_t)idx;
        }
        int th;
        int eh = block_excl_scan<THREADS>(head, th, wsum);
        if (head) vbeg[run + eh] = j;
        run += th;
    }
    __syncthreads();
    if (PV) PROF_MARK(3, 2);
    const int nv = run;
    if (MODE == 1) {
        // CentroidPoint (PCL 1.8.1 accumulators.hpp): fp32 running sums of x, y, z, intensity in ascending input index,
        // each divided by float(count); the intensity is the loader's scaled one when labels are given
        float4* tmp = (float4*)A.apri;  // idle in a VoxelGrid run: 16 of its 44 bytes per point hold the centroids
        for (int v = threadIdx.x; v < nv; v += THREADS) {
            const int j0 = vbeg[v];
            const int j1 = (v + 1 < nv) ? vbeg[v + 1] : m;
            float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
            for (int j = j0; j < j1; ++j) {
                const float4 p = A.pts[base + k_idx(keys[KX(j)])];
                sx += p.x;
                sy += p.y;
                sz += p.z;
                si += A.vg_labels ? p.w * A.vg_max_intensity : p.w;
            }
            const float fn = (float)(j1 - j0);
            tmp[(size_t)base + off + v] = make_float4(sx / fn, sy / fn, sz / fn, si / fn);
        }
        if (threadIdx.x == 0) A.vb_nvox[s * kMaxBuckets + b] = nv;
        __syncthreads();  // LDS is reused by the next item
        continue;
    }
    if (in_lds) {
        for (int j = threadIdx.x; j < m; j += THREADS) ints[j] = A.apri_int[(size_t)base + k_idx(keys[KX(j)])];
        __syncthreads();
    }
    if (PV) PROF_MARK(3, 3);
    // per voxel: sequential fp32 mean, then population variance accumulated as float += double
    for (int v = threadIdx.x; v < nv; v += THREADS) {
        const int j0 = vbeg[v];
        const int j1 = (v + 1 < nv) ? vbeg[v + 1] : m;
        float av = 0.f;
        if (in_lds) {
            for (int j = j0; j < j1; ++j) av += ints[j];
        } else {
            for (int j = j0; j < j1; ++j) av += A.apri_int[(size_t)base + k_idx(keys[j])];
        }
        const float fn = (float)(j1 - j0);
        av = av / fn;
        float cov = 0.f;
        for (int j = j0; j < j1; ++j) {
            const float in = in_lds ? ints[j] : A.apri_int[(size_t)base + k_idx(keys[j])];
            const double d = (double)(in - av);
            cov = (float)((double)cov + d * d);
        }
        cov = cov / fn;
        A.tmp_vox_key[(size_t)base + off + v] = K32 ? (int32_t)(((int64_t)b << P.vb_shift) + (int64_t)k_major(keys[KX(j0)]) - P.key_off)
                                                      : vx_unbias(k_major(keys[KX(j0)]));
        A.tmp_vox_cov[(size_t)base + off + v] = cov;
        A.tmp_vox_av[(size_t)base + off + v] = av;
    }
    __syncthreads();
    if (PV) PROF_MARK(3, 4);
    // vbeg aliases tmp_vox_begin in the oversize path: every thread rewrites only its own entries
    for (int v = threadIdx.x; v < nv; v += THREADS) A.tmp_vox_begin[(size_t)base + off + v] = off + vbeg[v];
    if (threadIdx.x == 0) A.vb_nvox[s * kMaxBuckets + b] = nv;
    __syncthreads();  // LDS is reused by the next item
    if (PV) PROF_MARK(3, 5);
    }
#undef KX
}

__global__ __launch_bounds__(1024) void k_vx_final_offsets(DevParams P, Arena A) {
    __shared__ int wsum[17];
    const int s = blockIdx.x;
    int c = (threadIdx.x < (unsigned)P.n_buckets) ? A.vb_nvox[s * kMaxBuckets + threadIdx.x] : 0;
    int total;
    int ex = block_excl_scan<1024>(c, total, wsum);
    if (threadIdx.x <= (unsigned)P.n_buckets) A.vox_off[s * (kMaxBuckets + 1) + threadIdx.x] = ex;
    if (threadIdx.x == 0) {
        A.counts[s * 8 + 6] = total;
        A.vox_pt_begin[(size_t)A.scan_off[s] + s + total] = A.counts[s * 8 + 4];
    }
}

__global__ __launch_bounds__(256) void k_vx_final(DevParams P, Arena A) {
    const int b = blockIdx.x, s = blockIdx.y;
    const int nv = A.vb_nvox[s * kMaxBuckets + b];
    if (nv == 0) return;
    const int base = A.scan_off[s];
    const int src = A.vb_off[s * (kMaxBuckets + 1) + b];
    const int dst = A.vox_off[s * (kMaxBuckets + 1) + b];
    for (int v = threadIdx.x; v < nv; v += 256) {
        A.vox_key[(size_t)base + dst + v] = A.tmp_vox_key[(size_t)base + src + v];
        A.vox_pt_begin[(size_t)base + s + dst + v] = A.tmp_vox_begin[(size_t)base + src + v];
        A.vox_av[(size_t)base + dst + v] = A.tmp_vox_av[(size_t)base + src + v];
        A.vox_cov[(size_t)base + dst + v] = A.tmp_vox_cov[(size_t)base + src + v];
    }
}

// ------------------------------------------------------------------------------------------
// Curved-voxel clustering (SSC::clusterAndCreateFrame, src/ssc.cpp:299-352).
// The reference walks the points in order; point i looks its index triple's 3x3x3 neighbourhood up in hash_cloud
// (findVoxelNeighbors, ssc.cpp:395-411: clipped to the grid, no sector wrap-around) and is merged with EVERY point of the
// occupied voxels it finds (ssc.cpp:316-345); a point that finds nothing opens a cluster of its own (ssc.cpp:347-353).
// The resulting partition does not depend on the order: it is the set of connected components of that relation.
//
// One workgroup per scan, the whole union-find in LDS.  Nodes are VOXELS, not points: a voxel that appears in anybody's
// neighbourhood ("touched") has all its points merged, so it is one node, named after its first point; the points of a
// voxel are walked as RUNS of equal index triples (a run's points share one neighbourhood) and a run that does not start
// its voxel -- only possible when different triples alias onto one voxel_idx, i.e. next to the -1 bins -- is an extra
// node.  Per scan: ~10 k voxels x 9 binary searches (the three sector neighbours of one (range, azimuth) pair are
// adjacent keys) + lock-free unions, all on a 64 KB key table and a 64 KB parent array in LDS; the canonical cluster name
// is the smallest apri index of the component.  Scans with more than kCcNodes voxels or kCcSlots points (128-beam scans on a
// fine grid) run the generic variant: nodes (keys, parents, bits) on arena scratch in HBM; its search still runs in LDS, one
// window of whole z-planes of the node list at a time (cc_search_windows), the few out-of-grid points of such a scan are
// listed by the regularity pass so that run detection and the regular bits look at their voxels only.
// ------------------------------------------------------------------------------------------
constexpr int kCcNodes = 14336;  // voxels + extra run openers per scan held in LDS
constexpr int kCcSlots = 65536;  // apri points per scan whose run / voxel start bits are held in LDS
constexpr int kCcThreads = 1024;
constexpr int kCcBoxes = 2048;   // bounding boxes per scan held in LDS (7 words each, in the key table once the search is over)
constexpr int kCcBuckets = 8192; // entries of the key-bucket index (uint16 node numbers; the generic variant: kCcNodes 32-bit entries)
constexpr int kCcExactMaxNodes = 4096;  // generic variant: nodes in components with irregular runs that are re-clustered exactly
constexpr int kCcBad = 256;          // generic variant: apri points outside the grid that are listed (more: every slot / node is looked at)
constexpr int kCcBadVoxel = 4096;    // ... and the largest voxel whose slots one thread walks for them
constexpr int kCcSlotsBig = 262144;  // generic variant: the nodes live in HBM, which leaves LDS for the bit arrays of this many points

static_assert(7 * kCcBoxes <= kCcNodes, "box records must fit the released key table");
constexpr size_t kCcLdsBytes = (size_t)(2 * kCcNodes + 3 * (kCcSlots / 32) + 3 * (kCcNodes / 32) + kCcBuckets / 2 + 32) * 4;
static_assert(kCcLdsBytes <= 160 * 1024, "one workgroup per CU: all of its LDS");
static_assert((size_t)(kCcNodes + 3 * (kCcSlotsBig / 32)) * 4 <= kCcLdsBytes, "generic layout inside the same LDS");

__device__ __forceinline__ int cc_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t cc_ldu(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int cc_find(int* parent, int x) {
    int p = cc_ld(&parent[x]);
    while (p != x) {
        const int gp = cc_ld(&parent[p]);
        if (gp != p) atomicMin(&parent[x], gp);  // path halving (only ever moves towards smaller ancestors)
        x = p;
        p = gp;
    }
    return x;
}
__device__ __forceinline__ void cc_union(int* parent, int a, int b) {
    for (;;) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        if (atomicCAS(&parent[a], a, b) == a) return;  // a was still a root: now hangs under the smaller b
    }
}
__device__ __forceinline__ int cc_lower_bound(const int* keys, int nv, int key) {
    int lo = 0, hi = nv;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}
__device__ __forceinline__ bool cc_bit(const int* bits, int i) { return (cc_ld(&bits[i >> 5]) >> (i & 31)) & 1; }
__device__ __forceinline__ void cc_set(int* bits, int i) { atomicOr(&bits[i >> 5], 1 << (i & 31)); }

// sorted voxel keys of the scan (`k`: LDS in the FAST variant, HBM otherwise) + bucket index in LDS: tab[b] = first node whose
// key is >= b << bshift, for the keys a search can ask for (0 .. R*S*Az - 1).  A lower bound then costs two table reads and a
// search among the few keys of one bucket instead of log2(nv) dependent reads.
template <typename TabT>
struct CcKeys {
    const int* k;
    int nv;
    const TabT* tab;
    int bshift;
    int first = 0;  // searches start at this node or later (a window of the node list: cc_search_windows)
};
template <typename TabT>
__device__ __forceinline__ int cc_lower_bound2(const CcKeys<TabT>& K, int key) {  // key >= 0
    const int b = key >> K.bshift;
    int lo = (int)K.tab[b], hi = (int)K.tab[b + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (K.k[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// neighbourhood of triple t for node `me`: unions with every occupied voxel found, marks them touched
template <typename TabT>
__device__ __forceinline__ bool cc_search(const CcKeys<TabT>& K, int* parent, int* touched, int me, int32_t t, int R, int S, int Az) {
    const int ri = (t & 2047) - 2, si = ((t >> 11) & 2047) - 2, ai = ((t >> 22) & 1023) - 2;
    const int ylo = max(si - 1, 0), yhi = min(si + 1, S - 1);
    bool found = false;
    if (ylo > yhi) return false;
    for (int z = ai - 1; z <= ai + 1; ++z) {
        if (z > Az - 1 || z < 0) continue;
        for (int x = ri - 1; x <= ri + 1; ++x) {
            if (x > R - 1 || x < 0) continue;
            const int k0 = x * S + ylo + z * R * S, k1 = k0 + (yhi - ylo);
            for (int u = cc_lower_bound2(K, k0); u < K.nv && K.k[u] <= k1; ++u) {
                cc_set(touched, u);  // every point of u joins (ssc.cpp:316)
                if (u != me) cc_union(parent, me, u);
                found = true;
            }
        }
    }
    return found;
}

// union that hands back the surviving root (the caller keeps it as its next starting point)
__device__ __forceinline__ int cc_union_r(int* parent, int a, int b) {
    for (;;) {
        a = cc_find(parent, a);
        b = cc_find(parent, b);
        if (a == b) return a;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        if (atomicCAS(&parent[a], a, b) == a) return b;
    }
}
// A REGULAR node is a voxel whose opener's triple is in range and encodes to the voxel's own key: it finds itself (so it is
// touched and found), and two regular voxels find each other, so such a pair is joined once, by the one with the larger
// index looking backwards (cc_search_half).  The few irregular nodes (-1 bins, extra runs) run the full search around their
// own triple, and an irregular VOXEL also looks around its key's own triple on behalf of the regular voxels that find it there.
template <typename TabT>
__device__ __forceinline__ void cc_search_canon(const CcKeys<TabT>& K, int* parent, int* touched, const int* regular, int me, int key, int R,
                                                int S, int Az) {
    const int RS = R * S;
    const int ai = key / RS, rem = key - ai * RS;
    const int ri = rem / S, si = rem - ri * S;
    const int ylo = max(si - 1, 0), yhi = min(si + 1, S - 1);
    for (int z = max(ai - 1, 0); z <= min(ai + 1, Az - 1); ++z)
        for (int x = max(ri - 1, 0); x <= min(ri + 1, R - 1); ++x) {
            const int k0 = x * S + ylo + z * RS, k1 = k0 + (yhi - ylo);
            for (int u = cc_lower_bound2(K, k0); u < K.nv && K.k[u] <= k1; ++u) {
                if (u == me || !cc_bit(regular, u)) continue;
                cc_set(touched, me);  // the regular voxel u finds me here (ssc.cpp:316)
                cc_union(parent, me, u);
            }
        }
}

// Regular voxels find themselves and each other mutually, so a regular node only looks BACKWARDS in key order (z-major, then
// range, then sector): the row (z, x-1) and the three rows of plane z-1 -- four lower bounds, taken in lockstep so that their
// reads overlap, instead of nine one after the other.
//   Consecutive occupied sectors of a row form a RUN; its nodes start out pointing at the run's head (cc_link_runs: no
// atomics, flat trees).  Two runs of neighbouring rows that touch are joined as soon as ONE touching pair is, and the pair
// where the node or the candidate is the HEAD of its run always exists (the head of the run that starts later touches the
// other run): a node in the middle of a run skips the candidates in the middle of theirs -- the unions left are one or two per
// run, every find is a step or two.  Irregular voxels do not search like this: they are runs of their own (always heads).
//   RUNS (parents in LDS, runs linked up front): a candidate that continues the run of the candidate before it is in that one's
// component already and is skipped, as is the predecessor of a node inside a run -- four or five unions instead of thirteen.
template <bool RUNS = false, typename TabT>
__device__ __forceinline__ void cc_search_half(const CcKeys<TabT>& K, int* parent, const int* heads, int me, int ri, int si, int ai, int R,
                                               int S, int Az) {
    const int ylo = max(si - 1, 0), yhi = min(si + 1, S - 1);
    int k0[4], k1[4], lo[4], hi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int x = (q == 3) ? ri - 1 : ri - 1 + q;
        const int z = (q == 3) ? ai : ai - 1;
        const bool valid = z >= 0 && x >= 0 && x <= R - 1;
        k0[q] = valid ? x * S + ylo + z * R * S : -1;
        k1[q] = k0[q] + (yhi - ylo);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = max(k0[q], 0) >> K.bshift;
        lo[q] = max((int)K.tab[b], K.first);
        hi[q] = max((int)K.tab[b + 1], K.first);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (k0[q] < 0) lo[q] = hi[q] = K.first;
    bool more = true;
    while (more) {
        more = false;
        int kq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) kq[q] = K.k[min((lo[q] + hi[q]) >> 1, K.nv - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (lo[q] < hi[q]) {
                const int mid = (lo[q] + hi[q]) >> 1;
                if (kq[q] < k0[q])
                    lo[q] = mid + 1;
                else
                    hi[q] = mid;
                more |= lo[q] < hi[q];
            }
        }
    }
    int cand[12], hw[12], pu[12];  // a range holds at most three keys (sectors y-1 .. y+1), distinct and ascending
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int u = min(lo[q] + e, K.nv - 1);
            cand[q * 3 + e] = K.k[u];
            hw[q * 3 + e] = heads ? heads[u >> 5] : -1;
            pu[q * 3 + e] = cc_ld(&parent[u]);  // read together: a candidate that hangs under my root already costs nothing more
        }
    const int prev = K.k[max(me - 1, 0)];
    // heads == nullptr (parents in LDS, where a union is cheap): every node is treated as a head -- all pairs are joined
    const bool me_head = heads ? ((heads[me >> 5] >> (me & 31)) & 1) : true;
    int ra = me;
    // the predecessor in the row is found too (ssc.cpp:316); inside a linked run it is the parent already
    if (me_head && me > 0 && si >= 1 && prev == K.k[me] - 1) ra = cc_union_r(parent, ra, me - 1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int u = lo[q] + e;
            if (!(k0[q] >= 0 && u < K.nv && cand[q * 3 + e] <= k1[q])) continue;
            if (pu[q * 3 + e] == ra) continue;  // (a parent link is for good: u is in ra's component)
            const bool u_head = (hw[q * 3 + e] >> (u & 31)) & 1;
            if (RUNS) {
                if (e == 0 || u_head) ra = cc_union_r(parent, ra, u);
                continue;
            }
            if (me_head || u_head) ra = cc_union_r(parent, ra, u);
        }
    // I am a candidate of the nodes of the next row and plane: hanging straight under the root I ended at keeps their finds at a
    // step and lets most of their candidates pass the test above
    if (ra < me) atomicMin(&parent[me], ra);
}

// The occupied voxels around triple t in the order findVoxelNeighbors lists them (ssc.cpp:395-411: range outermost, azimuth
// innermost); f(u) returns false to stop
template <typename TabT, typename Fn>
__device__ __forceinline__ void cc_for_each_listed(const CcKeys<TabT>& K, int32_t t, int R, int S, int Az, Fn f) {
    const int ri = (t & 2047) - 2, si = ((t >> 11) & 2047) - 2, ai = ((t >> 22) & 1023) - 2;
    for (int x = max(ri - 1, 0); x <= min(ri + 1, R - 1); ++x)
        for (int y = max(si - 1, 0); y <= min(si + 1, S - 1); ++y)
            for (int z = max(ai - 1, 0); z <= min(ai + 1, Az - 1); ++z) {
                const int key = x * S + y + z * R * S;
                const int u = cc_lower_bound2(K, key);
                if (u < K.nv && K.k[u] == key)
                    if (!f(u)) return;
            }
}

// Run heads of the voxel list + flat initial forest: head(v) = v starts a run (v == 0, or its key does not continue the
// predecessor's inside the row, or one of the two is irregular); parent[v] = the head of v's run (the latest head at or
// before v: ballots inside a wave, one LDS word per wave across the workgroup, a carry across the 1024-node chunks).
// `regular` = nullptr: every voxel is regular.  Extra-run nodes (>= nv) are their own parents.
__device__ __forceinline__ void cc_link_runs(const int* keys, int nv, int nn, int S, const int* regular, int* heads, int* parent, int* wlast,
                                             int off = 0) {  // off: the node number of entry 0 (a window of the node list)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int carry = 0;  // latest head before this chunk
    for (int j0 = 0; j0 < nv; j0 += kCcThreads) {
        const int j = j0 + tid;
        bool head = false;
        if (j < nv) {
            const int key = keys[j], pk = keys[max(j - 1, 0)];
            head = j == 0 || pk != key - 1 || key < 0 || (key % S) == 0;
            if (regular && !head) head = !cc_bit(regular, j) || !cc_bit(regular, j - 1);
            if (regular && !cc_bit(regular, j)) head = true;
        }
        const unsigned long long b = __ballot(head);
        if ((tid & 31) == 0 && j0 + (tid & ~31) < nv) heads[j >> 5] = (int)(unsigned)((tid & 32) ? (b >> 32) : b);
        if (lane == 0) wlast[wave] = b ? j0 + (wave << 6) + 63 - __clzll((long long)b) : -1;
        __syncthreads();
        int before = carry;  // latest head in the waves before mine
        for (int w = 0; w < wave; ++w) before = max(before, wlast[w]);
        const unsigned long long upto = b & ((2ull << lane) - 1ull);
        const int mine = upto ? j0 + (wave << 6) + 63 - __clzll((long long)upto) : before;
        if (j < nv) parent[j] = mine + off;
        int last = carry;
        for (int w = 0; w < kCcThreads / 64; ++w) last = max(last, wlast[w]);
        carry = last;
        __syncthreads();
    }
    for (int j = nv + tid; j < nn; j += kCcThreads) parent[j] = j;
}

// Generic variant (nodes in HBM): the backward search of the regular nodes, one WINDOW of whole z-planes at a time with the
// window's keys and parents in LDS.  A regular node only joins nodes of its own plane and of the plane before it, so a window
// holds the plane before its first one as well; a window's forest is flattened into the scan's parent array when it is done
// (parents are global node numbers throughout: the LDS arrays are addressed through pointers shifted by the window's start),
// and what the window joined among the nodes of that earlier plane -- which belong to the window before -- is joined in HBM,
// a few unions per window.  Returns false, with nothing changed, when two consecutive planes do not fit `cap` nodes.
//   regular == nullptr: every node is regular.  Nodes outside the grid's key range (negative keys, extra runs) are left as
// their own parents for the caller's pass over the irregular nodes.
// the same for a window of the node list held in LDS: the head bits first, then every node looks its head up in the words at
// and before its own (no barrier per chunk; entry 0 is a head); parents are node numbers, entry i is node off + i
__device__ __forceinline__ void cc_link_runs_lds(const int* keys, int count, int S, const int* regular, int* heads, int* parent, int off) {
    const int tid = threadIdx.x;
    for (int j0 = 0; j0 < count; j0 += kCcThreads) {
        const int j = j0 + tid;
        bool head = false;
        if (j < count) {
            const int key = keys[j], pk = keys[max(j - 1, 0)];
            head = j == 0 || pk != key - 1 || key < 0 || (key % S) == 0;
            if (regular && !head) head = !cc_bit(regular, j) || !cc_bit(regular, j - 1);
        }
        const unsigned long long b = __ballot(head);
        if ((tid & 31) == 0 && j0 + (tid & ~31) < count) heads[j >> 5] = (int)(unsigned)((tid & 32) ? (b >> 32) : b);
    }
    __syncthreads();
    for (int j = tid; j < count; j += kCcThreads) {
        int w = j >> 5;
        unsigned m = (unsigned)heads[w] & (0xffffffffu >> (31 - (j & 31)));
        while (!m) m = (unsigned)heads[--w];
        parent[j] = off + (w << 5) + 31 - __clz(m);
    }
}

#ifdef SCVOD_PROFILE
#define CCW_PROF_PARAM , unsigned long long& t_prev
#define CCW_PROF_ARG , t_prev
#else
#define CCW_PROF_PARAM
#define CCW_PROF_ARG
#endif
template <typename TabT>
__device__ __forceinline__ bool cc_search_windows(const CcKeys<TabT>& K, int* parent_g, const int* regular, int nn, int R, int S, int Az, int* lds,
                                                  int lds_words, int* wlast CCW_PROF_PARAM) {
    const int tid = threadIdx.x, RS = R * S;
    // LDS: [plane starts Az + 1][regular bits of the window][keys cap + 1][parents cap + 1]
    if (Az + 1 + 64 > lds_words / 4) return false;
    int* pstart = lds;
    const int cap = ((lds_words - (Az + 1)) * 32 / 66) - 40;  // 2 words + 2 bits per node
    int* reg_l = pstart + Az + 1;                              // [cap / 32 + 2]
    int* heads_l = reg_l + (cap >> 5) + 2;                     // [cap / 32 + 2]
    int* keys_l = heads_l + (cap >> 5) + 2;                    // [cap + 1]
    int* par_l = keys_l + cap + 1;                             // [cap + 1]
    for (int z = tid; z <= Az; z += kCcThreads) pstart[z] = z >= Az ? K.nv : cc_lower_bound2(K, z * RS);
    __syncthreads();
    bool too_big = false;
    for (int z = tid; z < Az; z += kCcThreads) too_big |= pstart[z + 1] - (max(pstart[max(z - 1, 0)] - 1, 0) & ~31) > cap;
    if (__syncthreads_or(too_big ? 1 : 0)) return false;
    for (int j = tid; j < nn; j += kCcThreads) parent_g[j] = j;
    CCW_MARK(1);
    int z = 0;
    while (z < Az) {
        __syncthreads();
        if (tid == 0) {
            const int p0 = pstart[max(z - 1, 0)];
            const int w0 = max(p0 - 1, 0) & ~31;  // (the predecessor read of the window's first node; whole words of the regular bits)
            int lo = z + 1, hi = Az;              // the last plane end that still fits
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (pstart[mid] - w0 <= cap)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            wlast[0] = w0;
            wlast[1] = p0;
            wlast[2] = pstart[z];
            wlast[3] = pstart[lo];
            wlast[4] = lo;
        }
        __syncthreads();
        const int w0 = wlast[0], p0 = wlast[1], c0 = wlast[2], c1 = wlast[3], z_end = wlast[4];
        CCW_MARK(2);
        for (int i = w0 + tid; i < c1; i += kCcThreads) keys_l[i - w0] = K.k[i];
        if (regular)
            for (int w = (w0 >> 5) + tid; w <= ((c1 - 1) >> 5); w += kCcThreads) reg_l[w - (w0 >> 5)] = regular[w];
        __syncthreads();
        cc_link_runs_lds(keys_l, c1 - w0, S, regular ? reg_l : (const int*)nullptr, heads_l, par_l, w0);
        __syncthreads();
        CCW_MARK(3);
        CcKeys<TabT> KL = K;
        KL.k = keys_l - w0;
        KL.nv = c1;
        KL.first = w0;
        int* par = par_l - w0;
        for (int j = c0 + tid; j < c1; j += kCcThreads) {
            if (regular && !((reg_l[(j - w0) >> 5] >> (j & 31)) & 1)) continue;
            const int key = KL.k[j];
            const int ai = key / RS, rem = key - ai * RS;
            const int ri = rem / S, si = rem - ri * S;
            cc_search_half<true>(KL, par, heads_l - (w0 >> 5), j, ri, si, ai, R, S, Az);
        }
        __syncthreads();
        CCW_MARK(4);
        for (int g = c0 + tid; g < c1; g += kCcThreads) parent_g[g] = cc_find(par, g);
        for (int g = p0 + tid; g < c0; g += kCcThreads) {
            const int r = cc_find(par, g);
            // (the window before left both under its own root when it had them in one component: the usual case)
            if (r != g && cc_ld(&parent_g[g]) != cc_ld(&parent_g[r])) cc_union(parent_g, g, r);
        }
        z = z_end;
        CCW_MARK(5);
    }
    __syncthreads();
    return true;
}

__device__ __forceinline__ uint32_t f2ord(float f) { return float_sort_key(f); }
__device__ __forceinline__ float ord2f(uint32_t u) { return u2f((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, d));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d));
    return v;
}

// FAST: every table of the scan in LDS, the pointers are LDS pointers at compile time (ds_* instead of flat_* accesses);
// gives up (returns false, nothing published yet) when the extra runs push the node count over kCcNodes.  The generic
// variant picks LDS or arena scratch per table at run time.
extern __shared__ int cc_smem[];
template <bool FAST>
__device__ __forceinline__ bool cc_scan_impl(const DevParams& P, const Arena& A, int from_apri, int* wsum, int* wlast, int& n_extra_s, int* bad_s, int s,
                                             int base, int n, int nv) {
    PROF_BEGIN();
    const int tid = threadIdx.x;
    if (!FAST) {  // bad_s: [kCcBad] listed points (then their voxels), [kCcBad] = how many there are
        if (tid == 0) bad_s[kCcBad] = 0;
        __syncthreads();
    }
    const int32_t* vbeg = A.vox_pt_begin + base + s;
    const int32_t* vpts = A.vox_pts + base;
    const int32_t* idx3 = A.apri_idx3 + base;
    const int R = P.bin.range_num, S = P.bin.sector_num, Az = P.bin.azimuth_num;
    const long long span = (long long)R * S * Az;
    // A scan is REGULAR when every apri point's index triple lies inside the grid and encodes to its voxel key (no -1 bins:
    // the usual case, the range / FOV verdict keeps such points out of apri_vec).  Then equal keys mean equal triples -- no
    // extra runs --, every voxel finds itself and every neighbour pair is mutual (cc_search_half).  One coalesced pass decides.
    bool bad = !(span > 0 && span < 0x7fffffffLL);
    for (int i0 = 0; i0 < n; i0 += kCcThreads * 4) {
        int tv[4], kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * kCcThreads + tid, n - 1);
            tv[u] = idx3[i];
            kv[u] = from_apri ? A.apri_key[(size_t)base + i] : 0;  // k_emit / k_bin_direct encode the key from the triple themselves
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tv[u];
            const int ri = (t & 2047) - 2, si = ((t >> 11) & 2047) - 2, ai = ((t >> 22) & 1023) - 2;
            bool b = ri < 0 || ri >= R || si < 0 || si >= S || ai < 0 || ai >= Az;
            if (from_apri) b |= kv[u] != ri * S + si + ai * R * S;
            bad |= b;
            // the generic variant lists the offenders (a handful of returns at polar angle exactly 0 per 128-beam scan): only
            // their voxels can hold a second run or be irregular, so two passes over every slot / node shrink to these voxels
            const int i = i0 + u * kCcThreads + tid;
            if (!FAST && b && i < n) {
                const int x = atomicAdd(&bad_s[kCcBad], 1);
                if (x < kCcBad) bad_s[x] = i;
            }
        }
    }
    const bool allreg = !__syncthreads_or(bad ? 1 : 0);
    const int n_bad = FAST ? 0 : bad_s[kCcBad];
    bool sparse = !FAST && !allreg && span > 0 && span < 0x7fffffffLL && n_bad <= kCcBad;
    const int nw = (n + 31) >> 5;             // words of the per-slot bit arrays
    // storage.  FAST: everything in LDS -- [keys N][parents N][vstart][rstart][prefix][touched][found][bucket index][regular].
    // Generic: nodes (keys, parents, touched / found) in HBM, LDS = [bucket index, 32-bit][vstart][rstart][prefix] for up to
    // kCcSlotsBig points; beyond that the bit arrays move to the arena's per-point scratch (nothing of it is live here).
    const bool slots_lds = FAST || n <= kCcSlotsBig;
    constexpr int kSlotWords = FAST ? kCcSlots / 32 : kCcSlotsBig / 32;
    int* vstart = FAST ? cc_smem + 2 * kCcNodes : (slots_lds ? cc_smem + kCcNodes : A.pt_voxel + base);
    int* rstart = slots_lds ? vstart + kSlotWords : A.tk_members + base;
    int* prefix = slots_lds ? rstart + kSlotWords : A.tk_clusters + base;
    int* extras = A.tk_uniq + base;            // slots of the extra run openers (rare: global scratch in both modes)
    int* extra_of_slot = A.tk_mbegin + base;   // slot -> index in extras
    if (tid == 0) n_extra_s = 0;
    for (int w = tid; w < nw; w += kCcThreads) {
        vstart[w] = 0;
        rstart[w] = 0;
    }
    __syncthreads();
    for (int v = tid; v < nv; v += kCcThreads) {
        const int k = vbeg[v];
        cc_set(vstart, k);
        if (!allreg) cc_set(rstart, k);
    }
    __syncthreads();
    CC_MARK(0);
    // runs inside a voxel: a slot whose triple differs from the previous slot's opens one (ssc.cpp:306-330 walks the
    // voxel's points with their own triples; equal triples have equal neighbourhoods)
    if (sparse) {
        const int* vkey = A.vox_key + base;
        bool fail = false;
        for (int x = tid; x < n_bad; x += kCcThreads) {
            const int i = bad_s[x];
            const int key = A.apri_key[(size_t)base + i];
            const int v = cc_lower_bound(vkey, nv, key);
            if (v >= nv || vkey[v] != key || vbeg[v + 1] - vbeg[v] > kCcBadVoxel) {
                fail = true;
                continue;
            }
            bad_s[x] = v;  // (from here on: the voxel)
            int t_prev = idx3[vpts[vbeg[v]]];
            for (int k = vbeg[v] + 1; k < vbeg[v + 1]; ++k) {
                const int t = idx3[vpts[k]];
                if (t != t_prev && !((atomicOr(&rstart[k >> 5], 1 << (k & 31)) >> (k & 31)) & 1)) {  // (two listed points of one voxel)
                    const int e = atomicAdd(&n_extra_s, 1);
                    extras[e] = k;
                    extra_of_slot[k] = e;
                }
                t_prev = t;
            }
        }
        if (__syncthreads_or(fail ? 1 : 0)) sparse = false;  // (what was found stays: the pass below skips the slots that are marked)
    }
    if (!allreg && !sparse) {
        for (int k0 = 0; k0 < n; k0 += kCcThreads * 4) {
            int pa[4], pb[4], ta[4], tb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // the two dependent gathers of four slots in flight together
                const int k = min(k0 + u * kCcThreads + tid, n - 1);
                pa[u] = vpts[k];
                pb[u] = vpts[max(k - 1, 0)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ta[u] = idx3[pa[u]];
                tb[u] = idx3[pb[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * kCcThreads + tid;
                if (k >= n || cc_bit(vstart, k)) continue;
                if (ta[u] != tb[u] && !((atomicOr(&rstart[k >> 5], 1 << (k & 31)) >> (k & 31)) & 1)) {
                    const int e = atomicAdd(&n_extra_s, 1);
                    extras[e] = k;
                    extra_of_slot[k] = e;
                }
            }
        }
    }
    CC_MARK(1);
    // prefix[w] = voxel starts in the words before w: voxel of slot k = prefix[k >> 5] + popc(vstart word up to k) - 1
    {
        int run = 0;
        for (int w0 = 0; w0 < nw; w0 += kCcThreads) {
            const int w = w0 + tid;
            const int c = (w < nw) ? __popc((unsigned)vstart[w]) : 0;
            int total;
            const int ex = block_excl_scan<kCcThreads>(c, total, wsum);
            if (w < nw) prefix[w] = run + ex;
            run += total;
        }
    }
    __syncthreads();
    const int n_extra = n_extra_s;
    const int nn = nv + n_extra;
    if (FAST && nn > kCcNodes) return false;
    int* lkeys = cc_smem;  // FAST only
    int* parent = FAST ? cc_smem + kCcNodes : A.cc_parent + base;
    int* touched = FAST ? cc_smem + 2 * kCcNodes + 3 * (kCcSlots / 32) : A.tk_npairs + base;
    int* found = FAST ? touched + kCcNodes / 32 : A.tk_hit + base;
    using TabT = typename std::conditional<FAST, uint16_t, uint32_t>::type;
    constexpr int kTabEntries = FAST ? kCcBuckets : kCcNodes;
    TabT* tab = FAST ? (TabT*)(cc_smem + 2 * kCcNodes + 3 * (kCcSlots / 32) + 2 * (kCcNodes / 32)) : (TabT*)cc_smem;
    CcKeys<TabT> K;
    K.nv = nv;
    K.k = FAST ? lkeys : A.vox_key + base;
    K.tab = tab;
    K.bshift = 0;
    const int kspan = (span > 0 && span < 0x7fffffffLL) ? (int)span : 0x7ffffffe;  // (an overflowing grid: all int keys searchable)
    while (((kspan - 1) >> K.bshift) + 2 > kTabEntries) ++K.bshift;
    const int nb = ((kspan - 1) >> K.bshift) + 1;
    if (FAST)
        for (int v = tid; v < nv; v += kCcThreads) lkeys[v] = A.vox_key[(size_t)base + v];
    __syncthreads();
    if (FAST) {
        for (int b = tid; b <= nb; b += kCcThreads) tab[b] = (TabT)cc_lower_bound(lkeys, nv, (int)min((long long)b << K.bshift, 0x7fffffffLL));
    } else {
        // keys in HBM: the nodes fill the table -- node v owns the buckets after its predecessor's up to its own
        auto bucket_of = [&](int key) -> int { return key < 0 ? -1 : min(key >> K.bshift, nb); };
        for (int v = tid; v < nv; v += kCcThreads) {
            const int bc = bucket_of(K.k[v]);
            const int bp = (v > 0) ? bucket_of(K.k[v - 1]) : -1;
            for (int b = bp + 1; b <= bc; ++b) tab[b] = (TabT)v;
            if (v == nv - 1)
                for (int b = bc + 1; b <= nb; ++b) tab[b] = (TabT)nv;
        }
    }
    __syncthreads();
    CC_MARK(2);
    auto voxel_of_slot = [&](int k) -> int {
        const unsigned m = (unsigned)vstart[k >> 5] & (0xffffffffu >> (31 - (k & 31)));
        return prefix[k >> 5] + __popc(m) - 1;
    };
    // the run that starts a voxel is node v, an extra run is node nv + e
    // parents in HBM (generic variant): runs linked up front and the head rule, so that few unions with short finds remain;
    // parents in LDS: plain forest, every neighbour pair joined (measured: the rule costs more than LDS unions save)
    int* heads = FAST ? nullptr : A.cl_count + base;  // [nv bits]
    if (FAST) {
        for (int j = tid; j < nn; j += kCcThreads) parent[j] = j;
        __syncthreads();
    }
    // generic variant: the bit arrays of the slots leave the LDS for the search (they are not read by it) and the regular
    // nodes are joined window by window on LDS copies of their keys and parents (cc_search_windows)
    constexpr int kWinWords = (int)(kCcLdsBytes / 4) - kCcNodes - 64;
    int* spill[3] = {A.pt_voxel + base, A.tk_members + base, A.tk_clusters + base};
    auto windows = [&](const int* regular) -> bool {
        if (FAST || kspan != (int)span) return false;
        if (slots_lds) {
            for (int w = tid; w < nw; w += kCcThreads) {
                spill[0][w] = vstart[w];
                spill[1][w] = rstart[w];
                spill[2][w] = prefix[w];
            }
            __syncthreads();
        }
        const bool done = cc_search_windows(K, parent, regular, nn, R, S, Az, cc_smem + kCcNodes, kWinWords, wlast CCW_PROF_ARG);
        if (slots_lds) {
            __syncthreads();
            for (int w = tid; w < nw; w += kCcThreads) {
                vstart[w] = spill[0][w];
                rstart[w] = spill[1][w];
                prefix[w] = spill[2][w];
            }
            __syncthreads();
        }
        return done;
    };
    if (allreg) {
        const bool windowed = windows(nullptr);
        if (!FAST && !windowed && tid == 0) atomicAdd(&A.cc_stats[3], 1);  // planes too large for a window: forest in HBM
        if (!FAST && !windowed) {
            cc_link_runs(K.k, nv, nn, S, (const int*)nullptr, heads, parent, wlast);
            __syncthreads();
        }
        if (FAST) {  // runs linked in LDS, their head bits in the (unused: every voxel is touched) touched words
            cc_link_runs_lds(lkeys, nv, S, (const int*)nullptr, touched, parent, 0);
            __syncthreads();
        }
        const int RS = R * S;
        int k_next = K.k[min(tid, nv - 1)];
        for (int j = tid; j < nv && !windowed; j += kCcThreads) {
            const int key = k_next;
            k_next = K.k[min(j + kCcThreads, nv - 1)];
            const int ai = key / RS, rem = key - ai * RS;  // the triple IS the key's decomposition
            const int ri = rem / S, si = rem - ri * S;
            if (FAST)
                cc_search_half<true>(K, parent, touched, j, ri, si, ai, R, S, Az);
            else
                cc_search_half(K, parent, heads, j, ri, si, ai, R, S, Az);
        }
    } else if (FAST && n <= 65535) {
        // ---- a scan with index triples outside the grid, all tables in LDS: the visiting order of clusterAndCreateFrame
        // (ssc.cpp:303-350) decides which of the asymmetric "finds" stick, so it is modelled exactly.
        // The loop reduces to a voxel-level state machine: a point is labelled once it was visited or once its voxel became
        // FULLY labelled at time T(k).  Visiting point i of voxel v with the listed voxels k1 .. km: if T(v) < i every kj joins
        // i and becomes fully labelled; otherwise with q = the first kj holding a labelled point (T(kj) < i or its first point
        // < i) the voxels kq .. km join i (kq becomes fully labelled only through a visited first point), the ones before q are
        // left alone; without any q all of them join.  T is well-founded in visiting order: Jacobi sweeps over all points
        // reach its fixed point in a handful of rounds; then one pass takes q per point and one pass does the unions.
        uint16_t* T16 = (uint16_t*)(cc_smem + kCcNodes);  // [nv] (the parents' LDS until the unions start), 0xffff = never
        uint16_t* P1 = T16 + kCcNodes;                    // [nv] first point of every voxel
        for (int v = tid; v < nv; v += kCcThreads) {
            T16[v] = 0xffffu;
            P1[v] = (uint16_t)vpts[vbeg[v]];
        }
        __syncthreads();
        auto labelled_voxel = [&](int u, int i) -> bool { return T16[u] != 0xffffu && (int)T16[u] < i; };
        // Work goes by RUN (node): the points of a run share their list.  Of a regular run (triple in the grid, encoding to the
        // voxel's key: its own voxel is in its list) only the first three points are events -- the second visit at the latest
        // labels the voxel fully, the third labels the whole list, later ones change nothing; an irregular run visits with
        // every point.  Up to three events share one walk over the list (their q's are found together).
        struct Run {
            int o, v, len;
            int32_t t;
            bool regular;
        };
        auto run_of_node = [&](int j) -> Run {
            Run r;
            r.o = (j < nv) ? vbeg[j] : extras[j - nv];
            r.v = (j < nv) ? j : voxel_of_slot(r.o);
            int w = (r.o + 1) >> 5;
            unsigned m = (w < nw) ? ((unsigned)rstart[w] & ~((1u << ((r.o + 1) & 31)) - 1u)) : 0u;
            while (!m && ++w < nw) m = (unsigned)rstart[w];
            r.len = (m ? min((w << 5) + __ffs((int)m) - 1, n) : n) - r.o;
            r.t = idx3[vpts[r.o]];
            const int ri = (r.t & 2047) - 2, si = ((r.t >> 11) & 2047) - 2, ai = ((r.t >> 22) & 1023) - 2;
            r.regular = kspan == (int)span && ri >= 0 && ri < R && si >= 0 && si < S && ai >= 0 && ai < Az && (ri * S + si + ai * R * S == K.k[r.v]);
            return r;
        };
        // the listed voxels of every node, looked up once: 27 uint16 per node in the box scratch of the scan (free until the
        // boxes are built) when it is large enough -- the rounds then walk the table instead of repeating 27 searches
        // (rows of 32 entries = 64 bytes, 16-byte aligned: a row is four wide loads in flight, then registers)
        uint16_t* nbr = (uint16_t*)(((uintptr_t)(A.cl_bbox + 7 * (size_t)base) + 15) & ~(uintptr_t)15);
        const bool tabled = (size_t)nn * 64 + 16 <= (size_t)n * 7 * sizeof(float);
        if (tabled) {
            for (int j = tid; j < nn; j += kCcThreads) {
                const int o = (j < nv) ? vbeg[j] : extras[j - nv];
                uint16_t* row = nbr + (size_t)j * 32;
                int jj = 0;
                cc_for_each_listed(K, idx3[vpts[o]], R, S, Az, [&](int u) -> bool {
                    row[jj++] = (uint16_t)u;
                    return true;
                });
                for (; jj < 32; ++jj) row[jj] = 0xffffu;
            }
            __syncthreads();
        }
        CC_MARK(10);
        auto walk = [&](int j, int32_t t, auto f) {  // f(u) in list order; false stops
            if (tabled) {
                const uint4* row = reinterpret_cast<const uint4*>(nbr + (size_t)j * 32);
                const uint4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
                const unsigned w[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
#pragma unroll
                for (int e = 0; e < 27; ++e) {
                    const int u = (int)((w[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                    if (u == 0xffff || !f(u)) return;
                }
            } else {
                cc_for_each_listed(K, t, R, S, Az, f);
            }
        };
        // what a round needs of a run, gathered once (arena scratch): opener slot, voxel, length (bit 31: regular), triple
        int4* runs = (int4*)A.tk_pairs + ((size_t)base + 1) / 2;  // [nn] (16-byte aligned inside the scan's 8 n bytes; nn <= n / 2 ... checked)
        const bool runs_cached = (size_t)nn * 2 + 2 <= (size_t)n;
        if (runs_cached) {
            for (int j = tid; j < nn; j += kCcThreads) {
                const Run r = run_of_node(j);
                runs[j] = make_int4(r.o, r.v, r.len | (r.regular ? (int)0x80000000u : 0), r.t);
            }
            __syncthreads();
        }
        auto cached_run = [&](int j) -> Run {
            if (!runs_cached) return run_of_node(j);
            const int4 c = runs[j];
            Run r;
            r.o = c.x;
            r.v = c.y;
            r.len = c.z & 0x7fffffff;
            r.regular = c.z < 0;
            r.t = c.w;
            return r;
        };
        // the next round's times: in the key table's LDS once the lists are tabled (no search needs the keys any more)
        int* Tn = tabled && runs_cached ? lkeys : A.cc_parent + base;  // [nv]
        for (int round = 0; round < 1024; ++round) {  // (about seven in practice; the fixed point exists after at most one round per visit)
            for (int v = tid; v < nv; v += kCcThreads) Tn[v] = 0x7fffffff;
            __syncthreads();
            for (int j = tid; j < nn; j += kCcThreads) {
                const Run r = cached_run(j);
                const int nev = r.regular ? min(r.len, 3) : r.len;
                for (int p0 = 0; p0 < nev; p0 += 3) {
                    int ie[3], qe[3];
                    bool cs[3];
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        ie[e] = (p0 + e < nev) ? vpts[r.o + p0 + e] : 0x7fffffff;  // (an absent event never sets anything)
                        // round 0 starts the iteration from the optimistic end (every visit labels its whole list): any start
                        // reaches the same fixed point, this one in fewer rounds than "nothing is ever labelled"
                        cs[e] = p0 + e < nev && (round == 0 || labelled_voxel(r.v, ie[e]));
                        qe[e] = -1;
                    }
                    int jj = 0;
                    walk(j, r.t, [&](int u) -> bool {
                        const int p1 = P1[u];
#pragma unroll
                        for (int e = 0; e < 3; ++e)
                            if (qe[e] < 0 && ie[e] != 0x7fffffff && (labelled_voxel(u, ie[e]) || p1 < ie[e])) qe[e] = jj;
                        ++jj;
                        return true;
                    });
                    jj = 0;
                    walk(j, r.t, [&](int u) -> bool {
                        const int p1 = P1[u];
                        int te = 0x7fffffff;  // the earliest of the events that labels u fully
#pragma unroll
                        for (int e = 2; e >= 0; --e)
                            if (ie[e] != 0x7fffffff && (cs[e] || qe[e] < 0 || jj > qe[e] || (jj == qe[e] && p1 < ie[e]))) te = ie[e];
                        if (te != 0x7fffffff && te < cc_ld(&Tn[u])) atomicMin(&Tn[u], te);
                        ++jj;
                        return true;
                    });
                }
            }
            __syncthreads();
            bool changed = false;
            for (int v = tid; v < nv; v += kCcThreads) {
                const int tv = cc_ld(&Tn[v]);  // (written by atomics: read past the L1)
                const uint16_t t16 = tv == 0x7fffffff ? (uint16_t)0xffffu : (uint16_t)tv;
                changed |= t16 != T16[v];
                T16[v] = t16;
            }
            CC_MARK(13);
#ifdef SCVOD_PROFILE
            if (tid == 0) atomicAdd(&g_prof[0][14], 100ull);  // rounds x 100 (prints as "us": 1.0 per round and scan)
#endif
            if (!__syncthreads_or(changed ? 1 : 0)) break;
        }
        CC_MARK(11);
        // the unions: what a run's LAST point joins contains what its earlier points joined (q only moves forward in time)
        int* qnode = A.cc_parent + base;  // [nn] q of every node
        for (int j = tid; j < nn; j += kCcThreads) {
            const Run r = cached_run(j);
            const int i = vpts[r.o + r.len - 1];
            int q = -1, jj = 0;
            if (!labelled_voxel(r.v, i))
                walk(j, r.t, [&](int u) -> bool {
                    if (labelled_voxel(u, i) || (int)P1[u] < i) {
                        q = jj;
                        return false;
                    }
                    ++jj;
                    return true;
                });
            qnode[j] = max(q, 0);
        }
        __syncthreads();
        for (int j = tid; j < nn; j += kCcThreads) parent[j] = j;  // (T16 / P1 are used up)
        for (int w = tid; w < ((nn + 31) >> 5); w += kCcThreads) {
            touched[w] = 0;
            found[w] = 0;
        }
        __syncthreads();
        for (int j = tid; j < nn; j += kCcThreads) {
            const int o = (j < nv) ? vbeg[j] : extras[j - nv];
            const int q = qnode[j];
            int jj = 0;
            walk(j, idx3[vpts[o]], [&](int u) -> bool {
                if (jj >= q) {
                    cc_set(touched, u);  // all of u's points join
                    if (u != j) cc_union(parent, j, u);
                }
                ++jj;
                return true;
            });
            if (jj > 0) cc_set(found, j);
        }
        CC_MARK(12);
    } else {
        int* regular = FAST ? cc_smem + 2 * kCcNodes + 3 * (kCcSlots / 32) + 2 * (kCcNodes / 32) + kCcBuckets / 2 : A.pt_cluster + base;
        int* triple = (int*)(A.tk_pairs + base);  // [nn] opener triples (arena scratch, free until the naming pass)
        const int nwords = (nn + 31) >> 5;
        constexpr int kG = FAST ? 4 : 8;
        if (sparse) {
            // every voxel is regular except the listed ones whose OPENER is outside the grid (or does not encode to the key);
            // only irregular nodes need their triple (a regular one's is its key's decomposition: triple_of)
            for (int w = tid; w < nwords; w += kCcThreads) {
                const int c = min(max(nv - (w << 5), 0), 32);
                const int word = c >= 32 ? -1 : (int)((1u << c) - 1u);
                regular[w] = word;
                touched[w] = word;  // a regular voxel finds itself
                found[w] = word;
            }
            __syncthreads();
            for (int x = tid; x < n_bad; x += kCcThreads) {
                const int v = bad_s[x];
                const int t = idx3[vpts[vbeg[v]]];
                const int ri = (t & 2047) - 2, si = ((t >> 11) & 2047) - 2, ai = ((t >> 22) & 1023) - 2;
                const bool reg = ri >= 0 && ri < R && si >= 0 && si < S && ai >= 0 && ai < Az && (ri * S + si + ai * R * S == K.k[v]);
                if (!reg) {
                    const int m = ~(1 << (v & 31));
                    atomicAnd(&regular[v >> 5], m);
                    atomicAnd(&touched[v >> 5], m);
                    atomicAnd(&found[v >> 5], m);
                    triple[v] = t;
                }
            }
            for (int e = tid; e < n_extra; e += kCcThreads) triple[nv + e] = idx3[vpts[extras[e]]];
        }
        for (int j0 = 0; j0 < nn && !sparse; j0 += kCcThreads * kG) {  // the three dependent gathers of kG nodes per thread in flight together
            int tc[kG];
#pragma unroll
            for (int u = 0; u < kG; ++u) {
                const int j = j0 + u * kCcThreads + tid;
                tc[u] = (j < nn) ? ((j < nv) ? vbeg[j] : extras[j - nv]) : 0;
            }
#pragma unroll
            for (int u = 0; u < kG; ++u) tc[u] = vpts[tc[u]];
#pragma unroll
            for (int u = 0; u < kG; ++u) tc[u] = idx3[tc[u]];
#pragma unroll
            for (int u = 0; u < kG; ++u) {
                const int j = j0 + u * kCcThreads + tid;
                const int t = tc[u];
                const int ri = (t & 2047) - 2, si = ((t >> 11) & 2047) - 2, ai = ((t >> 22) & 1023) - 2;
                const bool reg = kspan == (int)span && j < nv && ri >= 0 && ri < R && si >= 0 && si < S && ai >= 0 && ai < Az && (ri * S + si + ai * R * S == K.k[min(j, nv - 1)]);
                const unsigned long long b = __ballot(reg);
                if ((tid & 31) == 0 && (j >> 5) < nwords) {
                    const int word = (int)(unsigned)((tid & 32) ? (b >> 32) : b);
                    regular[j >> 5] = word;
                    touched[j >> 5] = word;  // a regular voxel finds itself
                    found[j >> 5] = word;
                }
                if (j < nn) triple[j] = t;
            }
        }
        __syncthreads();
        CCW_MARK(0);
        const bool windowed = windows(regular);
        if (!FAST && !windowed && tid == 0) atomicAdd(&A.cc_stats[3], 1);
        if (!FAST && !windowed) {
            cc_link_runs(K.k, nv, nn, S, regular, heads, parent, wlast);
            __syncthreads();
        }
        if (windowed) {
            // the regular nodes are done: only the irregular ones (a handful per scan) search, straight from the bit words
            // listed, then one WAVE per node: the nine (azimuth, range) rows around its triple on lanes 0-8 (cc_search), the nine
            // around its key's own triple on lanes 16-24 (cc_search_canon) -- a thread alone walks 18 searches in HBM one
            // after the other
            int* irr = heads;  // [nn] (the run heads are not used by the windowed search)
            if (tid == 0) wlast[8] = 0;
            __syncthreads();
            for (int w = tid; w < nwords; w += kCcThreads) {
                unsigned m = ~(unsigned)regular[w];
                if (w == nwords - 1 && (nn & 31)) m &= (1u << (nn & 31)) - 1u;
                while (m) {
                    irr[atomicAdd(&wlast[8], 1)] = (w << 5) + __ffs((int)m) - 1;
                    m &= m - 1;
                }
            }
            __syncthreads();
            const int n_irr = wlast[8], lane = tid & 63, RS = R * S;
            for (int x = tid >> 6; x < n_irr; x += kCcThreads / 64) {
                const int j = cc_ld(&irr[x]);
                const int q = lane & 15, canon = lane >> 4;  // canon: 0 = around the triple, 1 = around the key's triple
                int ri, si, ai;
                bool on = q < 9 && canon < 2;
                if (canon == 0) {
                    const int t = triple[j];
                    ri = (t & 2047) - 2, si = ((t >> 11) & 2047) - 2, ai = ((t >> 22) & 1023) - 2;
                } else {
                    const int key = (j < nv) ? K.k[j] : -1;
                    on = on && key >= 0 && key < kspan;
                    ai = key / RS;
                    const int rem = key - ai * RS;
                    ri = rem / S, si = rem - ri * S;
                }
                const int z = ai - 1 + q / 3, xx = ri - 1 + q % 3;
                const int ylo = max(si - 1, 0), yhi = min(si + 1, S - 1);
                on = on && ylo <= yhi && z >= 0 && z <= Az - 1 && xx >= 0 && xx <= R - 1;
                bool hit = false;
                if (on) {
                    const int k0 = xx * S + ylo + z * RS, k1 = k0 + (yhi - ylo);
                    for (int u = cc_lower_bound2(K, k0); u < K.nv && K.k[u] <= k1; ++u) {
                        if (canon == 0) {
                            cc_set(touched, u);  // every point of u joins (ssc.cpp:316)
                            if (u != j) cc_union(parent, j, u);
                            hit = true;
                        } else if (u != j && cc_bit(regular, u)) {
                            cc_set(touched, j);  // the regular voxel u finds me here (ssc.cpp:316)
                            cc_union(parent, j, u);
                        }
                    }
                }
                if (__any(hit) && lane == 0) cc_set(found, j);
            }
        }
        for (int j = tid; j < nn && !windowed; j += kCcThreads) {
            if (cc_bit(regular, j)) {
                const int RS = R * S, key = K.k[j];  // (a regular node's triple is its key's decomposition)
                const int ai = key / RS, rem = key - ai * RS;
                cc_search_half(K, parent, heads, j, rem / S, rem % S, ai, R, S, Az);
            } else {
                const int t = triple[j];
                if (cc_search(K, parent, touched, j, t, R, S, Az)) cc_set(found, j);
                if (j < nv) {
                    const int key = K.k[j];
                    if (key >= 0 && key < kspan) cc_search_canon(K, parent, touched, regular, j, key, R, S, Az);
                }
            }
        }
    }
    __syncthreads();
    CC_MARK(3);
    // every point of a touched voxel is merged with the voxel (ssc.cpp:316-345): the extra runs inside one join it
    for (int e = tid; e < n_extra; e += kCcThreads) {
        const int v = voxel_of_slot(extras[e]);
        if (cc_bit(touched, v)) cc_union(parent, nv + e, v);
    }
    __syncthreads();
    if (!FAST) CCW_MARK(6);
    if (!FAST && !allreg && slots_lds && kspan == (int)span) {
        // ---- generic variant, scan with irregular triples: what stands now is "everything found is joined".  The visiting
        // order (the model of the all-in-LDS variant above) can only SPLIT components that hold an
        // irregular run, and its state machine never looks outside such a component (a listed voxel is in the lister's
        // component; a voxel's labelling time depends on its finders only): those components are clustered again, exactly,
        // with the times / first points / next times of their voxels in arena scratch.
        int* regular = A.pt_cluster + base;
        const int* triple = (const int*)(A.tk_pairs + base);  // [nn] opener triples (written by the search above: the irregular nodes' at least)
        auto triple_of = [&](int j) -> int32_t {
            if (j >= nv || !cc_bit(regular, j)) return triple[j];
            const int RS = R * S, key = K.k[j];
            const int ai = key / RS, rem = key - ai * RS;
            return (int32_t)((rem / S + 2) | ((rem % S + 2) << 11) | ((ai + 2) << 22));
        };
        int* aff = A.tk_members + base;                       // [nn] root flag: the component holds an irregular run
        int* La = A.tk_clusters + base;                       // [na] the nodes of those components
        int* Tg = A.cl_count + base;                          // [nv] labelling times (run heads are used up)
        int* P1g = A.tk_nuniq + base;                         // [nv] first point of a voxel
        int* Tng = A.tk_cursor + base;                        // [nv] next round's times, then [na] q per listed node
        for (int j = tid; j < nn; j += kCcThreads) aff[j] = 0;
        __syncthreads();
        for (int w = tid; w < ((nn + 31) >> 5); w += kCcThreads) {  // the irregular nodes, straight from the bit words (extra runs have no bit set)
            unsigned m = ~(unsigned)regular[w];
            if (w == ((nn + 31) >> 5) - 1 && (nn & 31)) m &= (1u << (nn & 31)) - 1u;
            while (m) {
                const int j = (w << 5) + __ffs((int)m) - 1;
                m &= m - 1;
                aff[cc_find(parent, j)] = 1;
            }
        }
        __syncthreads();
        // (a sample of every eighth node first: components far beyond the bound are not even listed)
        int sampled = 0;
        for (int j0 = 0; j0 < nn; j0 += kCcThreads * 8) {
            const int j = j0 + tid * 8;
            sampled += __syncthreads_count(j < nn && aff[cc_find(parent, j)] != 0);
        }
        int na = 0;
        const int exact_max = A.cc_exact_max;
        const bool listed = (long long)sampled * 8 <= 2ll * exact_max;
        if (listed) {
            for (int j0 = 0; j0 < nn; j0 += kCcThreads) {
                const int j = j0 + tid;
                const bool a = j < nn && aff[cc_find(parent, j)] != 0;
                int total;
                const int ex = block_excl_scan<kCcThreads>(a ? 1 : 0, total, wsum);
                if (a) La[na + ex] = j;
                na += total;
            }
        }
        __syncthreads();
        // (components of tens of thousands of nodes -- an irregular return on a facade of a 128-beam scan -- are left as they
        // are: the rounds would cost milliseconds in HBM, and around such a point every voxel has finders with three or
        // more points, whose third visit joins all they list)
        if (tid == 0 && ((!listed && sampled > 0) || na > exact_max || (na > 0 && n_extra + na > n))) {
            // this scan keeps "everything found is joined" for its affected components (reported: scvod_batch_cluster_stats)
            atomicAdd(&A.cc_stats[0], 1);
            atomicAdd(&A.cc_stats[1], listed ? na : sampled * 8);
        }
        if (na > 0 && na <= exact_max) {
            struct Run {
                int o, v, len;
                int32_t t;
                bool regular;
            };
            auto run_of_node = [&](int j) -> Run {
                Run r;
                r.o = (j < nv) ? vbeg[j] : extras[j - nv];
                r.v = (j < nv) ? j : voxel_of_slot(r.o);
                int w = (r.o + 1) >> 5;
                unsigned m = (w < nw) ? ((unsigned)rstart[w] & ~((1u << ((r.o + 1) & 31)) - 1u)) : 0u;
                while (!m && ++w < nw) m = (unsigned)rstart[w];
                r.len = (m ? min((w << 5) + __ffs((int)m) - 1, n) : n) - r.o;
                r.t = triple_of(j);
                const int ri = (r.t & 2047) - 2, si = ((r.t >> 11) & 2047) - 2, ai = ((r.t >> 22) & 1023) - 2;
                r.regular = ri >= 0 && ri < R && si >= 0 && si < S && ai >= 0 && ai < Az && (ri * S + si + ai * R * S == K.k[r.v]);
                return r;
            };
            // run records + the listed voxels of every affected node (32-bit rows of 32) in the scan's box scratch when it fits
            char* scratch = (char*)(((uintptr_t)(A.cl_bbox + 7 * (size_t)base) + 15) & ~(uintptr_t)15);
            const bool tabled = (size_t)na * (16 + 128) + 16 <= (size_t)n * 7 * sizeof(float);
            int4* runs = (int4*)scratch;
            uint32_t* nbr = (uint32_t*)(scratch + (size_t)na * 16);
            for (int x = tid; x < na; x += kCcThreads) {
                const int j = La[x];
                if (j < nv) {
                    Tg[j] = 0x7fffffff;
                    P1g[j] = vpts[vbeg[j]];
                }
                if (tabled) {
                    const Run r = run_of_node(j);
                    runs[x] = make_int4(r.o, r.v, r.len | (r.regular ? (int)0x80000000u : 0), r.t);
                    uint32_t* row = nbr + (size_t)x * 32;
                    int jj = 0;
                    cc_for_each_listed(K, r.t, R, S, Az, [&](int u) -> bool {
                        row[jj++] = (uint32_t)u;
                        return true;
                    });
                    for (; jj < 32; ++jj) row[jj] = 0xffffffffu;
                }
            }
            __syncthreads();
            auto cached_run = [&](int x) -> Run {
                if (!tabled) return run_of_node(La[x]);
                const int4 c = runs[x];
                Run r;
                r.o = c.x;
                r.v = c.y;
                r.len = c.z & 0x7fffffff;
                r.regular = c.z < 0;
                r.t = c.w;
                return r;
            };
            auto walk = [&](int x, int32_t t, auto f) {  // f(u) in list order; false stops
                if (tabled) {
                    const uint32_t* row = nbr + (size_t)x * 32;
                    for (int e = 0; e < 27; ++e) {
                        const uint32_t u = row[e];
                        if (u == 0xffffffffu || !f((int)u)) return;
                    }
                } else {
                    cc_for_each_listed(K, t, R, S, Az, f);
                }
            };
            auto labelled_voxel = [&](int u, int i) -> bool { return Tg[u] < i; };
            for (int round = 0; round < 1024; ++round) {
                for (int x = tid; x < na; x += kCcThreads)
                    if (La[x] < nv) Tng[La[x]] = 0x7fffffff;
                __syncthreads();
                for (int x = tid; x < na; x += kCcThreads) {
                    const Run r = cached_run(x);
                    const int nev = r.regular ? min(r.len, 3) : r.len;
                    for (int p0 = 0; p0 < nev; p0 += 3) {
                        int ie[3], qe[3];
                        bool cs[3];
#pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            ie[e] = (p0 + e < nev) ? vpts[r.o + p0 + e] : 0x7fffffff;
                            cs[e] = p0 + e < nev && (round == 0 || labelled_voxel(r.v, ie[e]));
                            qe[e] = -1;
                        }
                        int jj = 0;
                        walk(x, r.t, [&](int u) -> bool {
                            const int p1 = P1g[u];
#pragma unroll
                            for (int e = 0; e < 3; ++e)
                                if (qe[e] < 0 && ie[e] != 0x7fffffff && (labelled_voxel(u, ie[e]) || p1 < ie[e])) qe[e] = jj;
                            ++jj;
                            return true;
                        });
                        jj = 0;
                        walk(x, r.t, [&](int u) -> bool {
                            const int p1 = P1g[u];
                            int te = 0x7fffffff;
#pragma unroll
                            for (int e = 2; e >= 0; --e)
                                if (ie[e] != 0x7fffffff && (cs[e] || qe[e] < 0 || jj > qe[e] || (jj == qe[e] && p1 < ie[e]))) te = ie[e];
                            if (te != 0x7fffffff && te < cc_ld(&Tng[u])) atomicMin(&Tng[u], te);
                            ++jj;
                            return true;
                        });
                    }
                }
                __syncthreads();
                bool changed = false;
                for (int x = tid; x < na; x += kCcThreads) {
                    const int j = La[x];
                    if (j >= nv) continue;
                    const int tv = cc_ld(&Tng[j]);
                    changed |= tv != Tg[j];
                    Tg[j] = tv;
                }
                if (!__syncthreads_or(changed ? 1 : 0)) break;
            }
            // q of every listed node at its last point, then the unions of these components from scratch
            int* qnode = A.tk_uniq + base + n_extra;  // [na] (behind the extras list)
            const bool q_fits = n_extra + na <= n;
            for (int x = tid; x < na; x += kCcThreads) {
                const Run r = cached_run(x);
                const int i = vpts[r.o + r.len - 1];
                int q = -1, jj = 0;
                if (!labelled_voxel(r.v, i))
                    walk(x, r.t, [&](int u) -> bool {
                        if (labelled_voxel(u, i) || P1g[u] < i) {
                            q = jj;
                            return false;
                        }
                        ++jj;
                        return true;
                    });
                if (q_fits) qnode[x] = max(q, 0);
            }
            __syncthreads();
            if (q_fits) {
                for (int x = tid; x < na; x += kCcThreads) {
                    const int j = La[x];
                    parent[j] = j;
                    atomicAnd(&found[j >> 5], ~(1 << (j & 31)));
                    if (j < nv) atomicAnd(&touched[j >> 5], ~(1 << (j & 31)));
                }
                __syncthreads();
                for (int x = tid; x < na; x += kCcThreads) {
                    const int j = La[x];
                    const int q = qnode[x];
                    int jj = 0;
                    walk(x, triple_of(j), [&](int u) -> bool {
                        if (jj >= q) {
                            cc_set(touched, u);
                            if (u != j) cc_union(parent, j, u);
                        }
                        ++jj;
                        return true;
                    });
                    if (jj > 0) cc_set(found, j);
                }
                __syncthreads();
                for (int x = tid; x < na; x += kCcThreads) {
                    const int j = La[x];
                    if (j < nv) continue;
                    const int v = voxel_of_slot(extras[j - nv]);
                    if (cc_bit(touched, v)) cc_union(parent, j, v);
                }
                __syncthreads();
            }
        }
    }
    if (!FAST) CCW_MARK(7);
    // canonical name of a component: the smallest apri index among the openers of its nodes (every other member of a
    // node sits behind its opener in an ascending point list)
    int* minpt = FAST ? lkeys : A.cl_count + base;  // the key table is not needed any more
    int* flat_g = A.tk_cursor + base;               // [nn] generic variant: root of every node
    for (int j = tid; j < nn; j += kCcThreads) minpt[j] = 0x7fffffff;
    __syncthreads();
    if (FAST) {
        for (int j = tid; j < nn; j += kCcThreads) {
            const int k = (j < nv) ? vbeg[j] : extras[j - nv];
            const int r = cc_find(parent, j);
            atomicMin(&minpt[r], vpts[k]);
        }
    } else {
        // minima in HBM: neighbours in the node list mostly share their root -- one atomic per distinct root of a wave.  Four nodes
        // per thread and step, their gathers and the hops to their roots in flight together; the roots are final here and are
        // kept (flat_g) for the numbering below
        for (int j0 = 0; j0 < nn; j0 += kCcThreads * 4) {
            int kk[4], rr[4], nm[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * kCcThreads + tid;
                kk[u] = (j < nn) ? ((j < nv) ? vbeg[j] : extras[j - nv]) : 0;
                rr[u] = (j < nn) ? cc_ld(&parent[j]) : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) nm[u] = vpts[kk[u]];
            bool more = true;
            while (more) {
                more = false;
                int pp[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) pp[u] = cc_ld(&parent[rr[u]]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (pp[u] != rr[u]) {
                        rr[u] = pp[u];
                        more = true;
                    }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * kCcThreads + tid;
                const int r = (j < nn) ? rr[u] : -1;
                if (j < nn) flat_g[j] = r;
                bool todo = r >= 0;
                while (__any(todo)) {
                    const int first = __ffsll((long long)__ballot(todo)) - 1;
                    const int r0 = __shfl(r, first);
                    const bool mine = todo && r == r0;
                    int m = mine ? nm[u] : 0x7fffffff;
#pragma unroll
                    for (int d = 32; d > 0; d >>= 1) m = min(m, __shfl_xor(m, d));
                    if ((tid & 63) == first) atomicMin(&minpt[r0], m);
                    if (mine) todo = false;
                }
            }
        }
    }
    __syncthreads();
    CC_MARK(4);
    // flatten, then number the components 0 .. ncl-1 in node order: parent[j] becomes the compact id of j's component
    int* flat = A.tk_cursor + base;                 // [nn] root of every node (arena scratch, nn <= n)
    if (FAST) {
        for (int j = tid; j < nn; j += kCcThreads) {
            int r = j;
            for (int q = parent[r]; q != r; q = parent[r]) r = q;  // read-only walk: the roots are final
            flat[j] = r;
        }
        __syncthreads();
    }  // (generic variant: written with the names above)
    int* slot_cid = (int*)(A.tk_pairs + base);      // [n] compact id of every apri point's component, -1 = a cluster of its own
    int* rootcid = slot_cid + n;                    // [nn]
    int* names = A.tk_nuniq + base;                 // [ncl] canonical name per compact id
    int ncl = 0;
    for (int j0 = 0; j0 < nn; j0 += kCcThreads) {
        const int j = j0 + tid;
        const bool isroot = (j < nn) && flat[j] == j;
        int total;
        const int ex = block_excl_scan<kCcThreads>(isroot ? 1 : 0, total, wsum);
        if (isroot) {
            rootcid[j] = ncl + ex;
            names[ncl + ex] = minpt[j];
        }
        ncl += total;
    }
    __syncthreads();
    for (int j = tid; j < nn; j += kCcThreads) parent[j] = rootcid[flat[j]];
    __syncthreads();
    CC_MARK(5);
    for (int k0 = 0; k0 < n; k0 += kCcThreads * 4) {  // four slots per thread and step: their loads are in flight together
        int pv[4], cv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pv[u] = vpts[min(k0 + u * kCcThreads + tid, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * kCcThreads + tid;
            int cid = -2;
            if (k < n) {
                const int v = voxel_of_slot(k);
                if (allreg || cc_bit(touched, v)) {
                    cid = parent[v];
                } else {
                    int o = k;  // opener of this slot's run: the closest run start at or before k (a voxel start is one)
                    {
                        int w = o >> 5;
                        unsigned m = (unsigned)rstart[w] & (0xffffffffu >> (31 - (o & 31)));
                        while (!m) m = (unsigned)rstart[--w];
                        o = (w << 5) + 31 - __clz(m);
                    }
                    const int node = (o == vbeg[v]) ? v : nv + extra_of_slot[o];
                    // a point that found nothing is a cluster of its own (ssc.cpp:347-353); the members of a run that found
                    // something joined what the opener joined
                    cid = cc_bit(found, node) ? parent[node] : -1;
                }
            }
            cv[u] = cid;
        }
        int nm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) nm[u] = names[max(cv[u], 0)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (cv[u] == -2) continue;
            slot_cid[pv[u]] = cv[u];  // (indexed by apri point from here on: the passes below stream the points in order)
            A.pt_cluster[(size_t)base + pv[u]] = cv[u] >= 0 ? nm[u] : pv[u];
        }
    }
    __syncthreads();
    CC_MARK(6);
    // ---- bounding boxes + type of every cluster (refineClusterByBoundingBox ssc.cpp:437-467, recognize ssc.cpp:849-872).
    // The boxes take the key table's LDS (the search is over); components beyond kCcBoxes take arena scratch.
    uint32_t* bb = (uint32_t*)cc_smem;
    uint32_t* ov = (uint32_t*)(A.cl_bbox + 7 * (size_t)base);  // overflow records, 7 words each
    for (int c = tid; c < ncl; c += kCcThreads) {
        uint32_t* r = c < kCcBoxes ? bb + 7 * c : ov + 7 * (size_t)(c - kCcBoxes);
        r[0] = r[1] = r[2] = 0xffffffffu;  // running minima (order-preserving encoding)
        r[3] = r[4] = r[5] = 0u;           // running maxima
        r[6] = 0u;                         // members
    }
    __syncthreads();
    {
        constexpr int U = 4;  // four points per thread and step: their index and point loads are in flight together (eight: slower)
        for (int i0 = 0; i0 < n; i0 += kCcThreads * U) {
            int cidv[U], srcv[U];
            float4 qv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * kCcThreads + tid;
                cidv[u] = (i < n) ? slot_cid[i] : -2;
                srcv[u] = (i < n && !from_apri) ? A.apri_src[(size_t)base + i] : 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = min(i0 + u * kCcThreads + tid, n - 1);
                if (from_apri) {  // apri_vec supplied by the caller: no input cloud on the device
                    const scvod_apri& a = A.apri[(size_t)base + i];
                    qv[u] = make_float4(a.x, a.y, a.z, 0.f);
                } else {  // cloud_use[i] = input point apri_src[i]
                    qv[u] = A.pts[base + srcv[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * kCcThreads + tid;
                const int cid = cidv[u];
                if (cid == -1) {  // one point: z extent 0 < 0.2 m, erased whatever toBeClass is (ssc.cpp:444)
                    A.pt_type[(size_t)base + i] = 0;
                    A.cl_count[(size_t)base + i] = 1;
                }
                if (cid < 0) continue;
                // most points lie inside the box their component has so far: plain reads first, an atomic only to grow it
                uint32_t* r = cid < kCcBoxes ? bb + 7 * cid : ov + 7 * (size_t)(cid - kCcBoxes);
                const uint32_t ox = f2ord(qv[u].x), oy = f2ord(qv[u].y), oz = f2ord(qv[u].z);
                const uint32_t b0 = cc_ldu(&r[0]), b1 = cc_ldu(&r[1]), b2 = cc_ldu(&r[2]), b3 = cc_ldu(&r[3]), b4 = cc_ldu(&r[4]), b5 = cc_ldu(&r[5]);
                if (ox < b0) atomicMin(&r[0], ox);
                if (oy < b1) atomicMin(&r[1], oy);
                if (oz < b2) atomicMin(&r[2], oz);
                if (ox > b3) atomicMax(&r[3], ox);
                if (oy > b4) atomicMax(&r[4], oy);
                if (oz > b5) atomicMax(&r[5], oz);
                atomicAdd(&r[6], 1u);
            }
        }
    }
    CC_MARK(7);
    __syncthreads();
    for (int c = tid; c < ncl; c += kCcThreads) {
        uint32_t* r = c < kCcBoxes ? bb + 7 * c : ov + 7 * (size_t)(c - kCcBoxes);
        const float mnx = ord2f(r[0]), mny = ord2f(r[1]), mnz = ord2f(r[2]);
        const float mxx = ord2f(r[3]), mxy = ord2f(r[4]), mxz = ord2f(r[5]);
        const int cnt = (int)r[6];
        const float diff_zf = mxz - mnz;
        uint32_t t;
        if (mnz > 0.f || cnt < P.to_be_class || diff_zf < 0.2f) {
            t = 0;
        } else {
            const double square = (double)(mxx - mnx) * (double)(mxy - mny);
            if (square > (double)P.car_square)
                t = 1;
            else if ((double)mnz < (double)P.min_z && square < (double)P.car_square && (double)mxz < (double)P.max_z)
                t = 2;
            else
                t = 1;
        }
        r[0] = t;
        r[1] = 0;  // (the box is used up: the word counts the cluster's voxels below)
        r[4] = 0xffffffffu;  // ... and this one takes the lowest voxel slot carrying the cluster's label (its id in the tracking chain)
        A.cl_count[(size_t)base + names[c]] = cnt;  // Cluster::occupy_pts.size(), kept at the cluster's canonical name
    }
    __syncthreads();
    // ---- successor table of the scan: Voxel::label after clusterAndCreateFrame + refineClusterByBoundingBox (ssc.cpp:388-392,
    // 461-466) = the cluster of the voxel's first point, -1 when the refine erased it; |occupy_voxels| of that cluster
    // (sampleVec of its points' voxel_idx, ssc.cpp:382-384) = the number of voxels carrying its label; its type.  This is what
    // SSC::tracking probes a predecessor against and what a shard exports (scvod_batch_export_table).
    auto cid_of_voxel = [&](int v) -> int {  // the voxel's first point opens node v
        return (allreg || cc_bit(touched, v) || cc_bit(found, v)) ? parent[v] : -1;
    };
    for (int v = tid; v < nv; v += kCcThreads) {
        const int cid = cid_of_voxel(v);
        if (cid < 0) continue;
        uint32_t* r = cid < kCcBoxes ? bb + 7 * cid : ov + 7 * (size_t)(cid - kCcBoxes);
        if (r[0]) {
            atomicAdd(&r[1], 1u);
            atomicMin(&r[4], (uint32_t)v);
        }
    }
    __syncthreads();
    for (int v = tid; v < nv; v += kCcThreads) {
        const int cid = cid_of_voxel(v);
        int4 rec = make_int4(A.vox_key[(size_t)base + v], -1, 0, 0);
        int rep = -1;
        if (cid >= 0) {
            const uint32_t* r = cid < kCcBoxes ? bb + 7 * cid : ov + 7 * (size_t)(cid - kCcBoxes);
            if (r[0]) {
                rec.y = names[cid];
                rec.z = (int)r[1];
                rec.w = (int)r[0];
                rep = (int)r[4];
            }
        }
        A.vox_track[(size_t)base + v] = rec;
        A.vox_rep[(size_t)base + v] = rep;
        // for the max_name pass that follows (scvod_lastname.hip): the cluster of the voxel's first point whether the refine erased
        // it or not (-1: not decided here), and that point -- two arrays the voxel stage no longer needs
        A.tmp_vox_key[(size_t)base + v] = cid >= 0 ? names[cid] : -1;
        ((int32_t*)A.sorted_idx)[(size_t)base + v] = vpts[vbeg[v]];
    }
    // ---- member lists of the car clusters (what SSC::tracking walks, ssc.cpp:1274-1321): the car roots in ascending order
    // (tk_clusters), exclusive offsets of their sizes (tk_mbegin at the root), and -- in the per-point pass below -- every
    // car point appended to its cluster's list (tk_members; the order inside a list is immaterial).  The car components are
    // few (tens): listed in the LDS the bit arrays released, ranked by counting.
    __syncthreads();
    int ncar = 0;
    for (int c0 = 0; c0 < ncl; c0 += kCcThreads) {
        const int c = c0 + tid;
        const uint32_t* r = c < kCcBoxes ? bb + 7 * c : ov + 7 * (size_t)(c - kCcBoxes);
        ncar += __syncthreads_count(c < ncl && r[0] == 2u);
    }
    const int carcap = slots_lds ? (3 * kSlotWords) / 4 : 0;
    const bool car_lds = ncar <= carcap;  // four lists of ncar entries; arena scratch (dead by now) when they are many
    int* carname = car_lds ? vstart : A.tk_uniq + base;
    int* carcid = car_lds ? carname + ncar : A.cc_parent + base;
    int* scnt = car_lds ? carcid + ncar : A.tk_hit + base;
    int* scid = car_lds ? scnt + ncar : A.tk_npairs + base;
    {
        int run = 0;
        for (int c0 = 0; c0 < ncl; c0 += kCcThreads) {
            const int c = c0 + tid;
            const uint32_t* r = c < kCcBoxes ? bb + 7 * c : ov + 7 * (size_t)(c - kCcBoxes);
            const bool car = c < ncl && r[0] == 2u;
            int total;
            const int ex = block_excl_scan<kCcThreads>(car ? 1 : 0, total, wsum);
            if (car) {
                carname[run + ex] = names[c];
                carcid[run + ex] = c;
            }
            run += total;
        }
    }
    __syncthreads();
    for (int j = tid; j < ncar; j += kCcThreads) {
        const int mine = carname[j], cid = carcid[j];
        int rank = 0;
        for (int i = 0; i < ncar; ++i) rank += carname[i] < mine ? 1 : 0;
        const uint32_t* r = cid < kCcBoxes ? bb + 7 * cid : ov + 7 * (size_t)(cid - kCcBoxes);
        scnt[rank] = (int)r[6];
        scid[rank] = cid;
        A.tk_clusters[(size_t)base + rank] = mine;
        A.tk_crep[(size_t)base + rank] = (int)r[4];
    }
    __syncthreads();
    {
        int run = 0;
        for (int j0 = 0; j0 < ncar; j0 += kCcThreads) {
            const int j = j0 + tid;
            const int cnt = j < ncar ? scnt[j] : 0;
            int total;
            const int ex = block_excl_scan<kCcThreads>(cnt, total, wsum);
            if (j < ncar) {
                const int cid = scid[j];
                uint32_t* r = cid < kCcBoxes ? bb + 7 * cid : ov + 7 * (size_t)(cid - kCcBoxes);
                r[2] = (uint32_t)(run + ex);  // where the cluster's member list starts
                r[3] = 0u;                    // its cursor
                A.tk_mbegin[(size_t)base + names[cid]] = run + ex;
            }
            run += total;
        }
        if (tid == 0) {
            A.tk_scan[s * 4 + 0] = ncar;
            A.tk_scan[s * 4 + 1] = run;
        }
    }
    __syncthreads();
    CC_MARK(8);
    for (int i0 = 0; i0 < n; i0 += kCcThreads * 4) {
        int cv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * kCcThreads + tid;
            cv[u] = (i < n) ? slot_cid[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cid = cv[u];
            if (cid < 0) continue;
            uint32_t* r = cid < kCcBoxes ? bb + 7 * cid : ov + 7 * (size_t)(cid - kCcBoxes);
            const uint32_t t = r[0];
            const int i = i0 + u * kCcThreads + tid;
            A.pt_type[(size_t)base + i] = (uint8_t)t;
            if (t == 2u) {
                A.tk_members[(size_t)base + r[2] + atomicAdd(&r[3], 1u)] = i;
                if (!from_apri) A.pt_mapcls[(size_t)base + A.apri_src[(size_t)base + i]] = kMapCar;  // (the batch starts with clean marks)
            }
        }
    }
    CC_MARK(9);
    return true;
}

// the scans with triples outside the grid run the visiting-order model (milliseconds instead of a quarter of one): they are
// handed out first, so that they overlap the regular ones instead of trailing them.  One workgroup, stable partition.
__global__ __launch_bounds__(1024) void k_cc_order(Arena A) {
    __shared__ int wsum[17];
    const int B = A.n_scans;
    int n_irr = 0;
    for (int s0 = 0; s0 < B; s0 += 1024) n_irr += __syncthreads_count(s0 + (int)threadIdx.x < B && A.scan_irr[s0 + threadIdx.x] != 0);
    int run_i = 0, run_r = 0;
    for (int s0 = 0; s0 < B; s0 += 1024) {
        const int s = s0 + threadIdx.x;
        const bool irr = s < B && A.scan_irr[s] != 0;
        int ti, tr;
        const int ei = block_excl_scan<1024>(irr ? 1 : 0, ti, wsum);
        const int er = block_excl_scan<1024>((s < B && !irr) ? 1 : 0, tr, wsum);
        if (s < B) A.cc_perm[irr ? run_i + ei : n_irr + run_r + er] = s;
        run_i += ti;
        run_r += tr;
    }
}

__global__ __launch_bounds__(kCcThreads) void k_cc_scan(DevParams P, Arena A, int from_apri) {
    __shared__ int wsum[17];
    __shared__ int wlast[kCcThreads / 64];
    __shared__ int n_extra_s;
    __shared__ int bad_s[kCcBad + 1];
    const int s = A.cc_perm[blockIdx.x];
    const int base = A.scan_off[s];
    const int n = A.counts[s * 8 + 4];
    const int nv = A.counts[s * 8 + 6];
    if (n <= 0) {
        if (threadIdx.x < 2) A.tk_scan[s * 4 + threadIdx.x] = 0;  // no car clusters, no car points
        return;
    }
    bool done = false;
    if (n <= kCcSlots && nv <= kCcNodes && (long long)P.bin.range_num * P.bin.sector_num * P.bin.azimuth_num < 0x7fffffffLL) {
        done = cc_scan_impl<true>(P, A, from_apri, wsum, wlast, n_extra_s, bad_s, s, base, n, nv);
        __syncthreads();
    }
    if (!done) cc_scan_impl<false>(P, A, from_apri, wsum, wlast, n_extra_s, bad_s, s, base, n, nv);
}

// ------------------------------------------------------------------------------------------
// Scan-vs-next-scan differencing, bulk part of SSC::tracking (ssc.cpp:1274-1321)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int cluster_of_point(const int32_t* begin, int n_clusters, int k) {
    int lo = 0, hi = n_clusters;  // find c with begin[c] <= k < begin[c+1]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (begin[mid] <= k)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_track_probe(DevParams P, Arena A, TrackJob J, int batch_mode) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= J.n_pts) return;
    const int c = cluster_of_point(J.pt_cluster_begin, J.n_clusters, k);
    const int pair = J.cluster_pair ? J.cluster_pair[c] : 0;
    float4 q;
    const int32_t* keys;
    const int32_t* labels = J.next_labels;
    int nv;
    if (batch_mode) {
        const int sb = A.scan_off[pair];
        if (batch_mode == 1) {  // members are apri_vec indices; the point itself is read through apri_src
            q = A.pts[sb + A.apri_src[(size_t)sb + J.members[k]]];
        } else {  // apri_vec supplied by the caller (no input cloud on the device)
            const scvod_apri& a = A.apri[(size_t)sb + J.members[k]];
            q = make_float4(a.x, a.y, a.z, a.intensity);
        }
        const int nb = A.scan_off[pair + 1];
        keys = A.vox_key + nb;
        nv = A.counts[(pair + 1) * 8 + 6];
    } else {
        q = J.pts[k];
        keys = J.next_keys;
        nv = J.n_next_vox;
    }
    const float* T = J.T + 12 * pair;
    // Utility::transformCloud (utility.h:401-404): explicit fp32 dot products, no FMA
    float x = T[0] * q.x + T[1] * q.y + T[2] * q.z + T[3];
    float y = T[4] * q.x + T[5] * q.y + T[6] * q.z + T[7];
    float z = T[8] * q.x + T[9] * q.y + T[10] * q.z + T[11];
    Apri a;
    apri_of_point(P.bin, x, y, z, q.w, a);  // no range/FOV rejection, no clamping (ssc.cpp:1280-1286)
    const int key = a.voxel_idx;
    int lo = 0, hi = nv;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (keys[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    int slot = -1;
    if (lo < nv && keys[lo] == key) {
        if (!labels || labels[lo] != -1) slot = lo;
    }
    J.hit_slot[k] = slot;
}

// Batch form of the probe: blockIdx.y = scan pair, so the block can stage the next scan's voxel keys in LDS once
// and run every binary search there (eleven dependent L2 round trips per point otherwise).
constexpr int kTrackLdsKeys = 8192;
__global__ __launch_bounds__(256) void k_track_probe_pair(DevParams P, Arena A, TrackJob J, int batch_mode) {
    __shared__ int32_t skeys[kTrackLdsKeys];
    const int pair = blockIdx.y;
    const int k0 = J.pair_pt_begin[pair], k1 = J.pair_pt_begin[pair + 1];
    if (k0 + (int)blockIdx.x * 256 >= k1) return;
    const int sb = A.scan_off[pair];
    const int nb = A.scan_off[pair + 1];
    const int nv = A.counts[(pair + 1) * 8 + 6];
    const int32_t* gkeys = A.vox_key + nb;
    const bool in_lds = nv <= kTrackLdsKeys;
    if (in_lds)
        for (int i = threadIdx.x; i < nv; i += 256) skeys[i] = gkeys[i];
    __syncthreads();
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = J.T[12 * pair + i];
    for (int k = k0 + blockIdx.x * 256 + threadIdx.x; k < k1; k += gridDim.x * 256) {
        float4 q;
        if (batch_mode == 1) {  // members are apri_vec indices; the point itself is read through apri_src
            q = A.pts[sb + A.apri_src[(size_t)sb + J.members[k]]];
        } else {  // apri_vec supplied by the caller (no input cloud on the device)
            const scvod_apri& a = A.apri[(size_t)sb + J.members[k]];
            q = make_float4(a.x, a.y, a.z, a.intensity);
        }
        // Utility::transformCloud (utility.h:401-404): explicit fp32 dot products, no FMA
        float x = T[0] * q.x + T[1] * q.y + T[2] * q.z + T[3];
        float y = T[4] * q.x + T[5] * q.y + T[6] * q.z + T[7];
        float z = T[8] * q.x + T[9] * q.y + T[10] * q.z + T[11];
        Apri a;
        apri_of_point(P.bin, x, y, z, q.w, a);  // no range/FOV rejection, no clamping (ssc.cpp:1280-1286)
        const int key = a.voxel_idx;
        int lo = 0, hi = nv;
        if (in_lds) {
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (skeys[mid] < key)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            J.hit_slot[k] = (lo < nv && skeys[lo] == key) ? lo : -1;
        } else {
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (gkeys[mid] < key)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            J.hit_slot[k] = (lo < nv && gkeys[lo] == key) ? lo : -1;
        }
    }
}

// sampleVec of the hit list of every cluster (ssc.cpp:1319-1321) without a sort: the hits are slots of the next
// scan's voxel table, so a per-cluster bitset over the table (LDS) gives the sorted unique list directly.
constexpr int kTrackBitWords = 8192;  // 262144 table slots; larger tables take the sort path below
__global__ __launch_bounds__(128) void k_track_unique_bits(Arena A, TrackJob J, int batch_mode) {
    __shared__ uint32_t bits[kTrackBitWords];
    __shared__ int wsum[3];
    const int c = blockIdx.x;
    const int k0 = J.pt_cluster_begin[c], k1 = J.pt_cluster_begin[c + 1];
    const int pair = J.cluster_pair ? J.cluster_pair[c] : 0;
    const int nv = batch_mode ? A.counts[(pair + 1) * 8 + 6] : J.n_next_vox;
    const int nw = (nv + 31) >> 5;
    if (nw > kTrackBitWords) return;  // handled by k_track_unique
    for (int w = threadIdx.x; w < nw; w += 128) bits[w] = 0u;
    __syncthreads();
    for (int j = k0 + threadIdx.x; j < k1; j += 128) {
        const int slot = J.hit_slot[j];
        if (slot >= 0) atomicOr(&bits[slot >> 5], 1u << (slot & 31));
    }
    __syncthreads();
    int run = 0;
    for (int w0 = 0; w0 < nw; w0 += 128) {
        const int w = w0 + threadIdx.x;
        const uint32_t word = (w < nw) ? bits[w] : 0u;
        int total;
        const int ex = block_excl_scan<128>(__popc(word), total, wsum);
        uint32_t rest = word;
        int o = k0 + run + ex;
        while (rest) {
            const int b = __ffs(rest) - 1;
            rest &= rest - 1;
            J.uniq_slots[o++] = (w << 5) + b;
        }
        run += total;
    }
    if (threadIdx.x == 0) J.uniq_count[c] = run;
}

template <int CAP, int THREADS>
__global__ __launch_bounds__(THREADS) void k_track_unique(TrackJob J, int table_words) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* l_keys = (uint32_t*)smem;
    int* wsum = (int*)(smem + (size_t)CAP * 4);
    const int c = blockIdx.x;
    const int k0 = J.pt_cluster_begin[c], k1 = J.pt_cluster_begin[c + 1];
    const int m = k1 - k0;
    if (table_words <= kTrackBitWords) return;  // done by k_track_unique_bits
    uint32_t* keys;
    // misses sort to the end as 0xffffffff
    if (m <= CAP) {
        keys = l_keys;
        int np2 = 8;
        while (np2 < m) np2 <<= 1;
        for (int j = threadIdx.x; j < np2; j += THREADS) keys[j] = (j < m) ? (uint32_t)J.hit_slot[k0 + j] : 0xffffffffu;
        __syncthreads();
        block_bitonic_sort_pow2<THREADS, false, 3>(keys, np2);  // 32-bit keys: compare-exchange = v_min + v_max
    } else {
        keys = (uint32_t*)(J.work + k0);
        for (int j = threadIdx.x; j < m; j += THREADS) keys[j] = (uint32_t)J.hit_slot[k0 + j];
        __syncthreads();
        block_bitonic_sort<THREADS, false>(keys, m);
    }
    int run = 0;
    for (int c0 = 0; c0 < m; c0 += THREADS) {
        int j = c0 + threadIdx.x;
        int head = 0;
        uint32_t v = 0xffffffffu;
        if (j < m) {
            v = keys[j];
            const uint32_t prev = keys[j > 0 ? j - 1 : 0];
            head = (v != 0xffffffffu) && (j == 0 || prev != v);
        }
        int th;
        int eh = block_excl_scan<THREADS>(head, th, wsum);
        if (head) J.uniq_slots[k0 + run + eh] = (int32_t)v;
        run += th;
    }
    if (threadIdx.x == 0) J.uniq_count[c] = run;
}

// ------------------------------------------------------------------------------------------
// Correspondence search (north_star "GICP correspondence search"; reference analogue: kd-tree
// 1-NN / radius look-ups of src/evaluate.cpp:79-145).  LDS-tiled brute force, exact.
// ------------------------------------------------------------------------------------------
constexpr int kNnThreads = 256;
constexpr int kNnTile = 2048;
__global__ __launch_bounds__(kNnThreads) void k_nn_brute(const float* __restrict__ map_xyz, int n_map,
                                                          const float* __restrict__ q_xyz, int n_q, float r2,
                                                          int32_t* nn_idx, float* nn_sq, uint8_t* within) {
    __shared__ float tx[kNnTile], ty[kNnTile], tz[kNnTile];
    const int q = blockIdx.x * kNnThreads + threadIdx.x;
    float qx = 0, qy = 0, qz = 0;
    if (q < n_q) {
        qx = q_xyz[3 * (size_t)q];
        qy = q_xyz[3 * (size_t)q + 1];
        qz = q_xyz[3 * (size_t)q + 2];
    }
    float best = 0.f;
    int bi = -1;
    for (int t0 = 0; t0 < n_map; t0 += kNnTile) {
        int tn = min(kNnTile, n_map - t0);
        __syncthreads();
        for (int j = threadIdx.x; j < tn; j += kNnThreads) {
            tx[j] = map_xyz[3 * (size_t)(t0 + j)];
            ty[j] = map_xyz[3 * (size_t)(t0 + j) + 1];
            tz[j] = map_xyz[3 * (size_t)(t0 + j) + 2];
        }
        __syncthreads();
        if (q < n_q) {
            for (int j = 0; j < tn; ++j) {
                float dx = tx[j] - qx, dy = ty[j] - qy, dz = tz[j] - qz;
                float d = (dx * dx + dy * dy) + dz * dz;
                if (bi < 0 || d < best) {
                    best = d;
                    bi = t0 + j;
                }
            }
        }
    }
    if (q < n_q) {
        nn_idx[q] = bi;
        nn_sq[q] = best;
        within[q] = (bi >= 0 && best < r2) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------
// host-side launch sequences
// ------------------------------------------------------------------------------------------
#define TH_BEGIN(name) \
    if (th) th(tu, name, 1)
#define TH_END(name) \
    if (th) th(tu, name, 0)

#ifndef SCVOD_SORT_LGE
#define SCVOD_SORT_LGE 4
#endif
constexpr int kLGE = SCVOD_SORT_LGE;           // keys per thread = 2^kLGE in the LDS sorts
constexpr int kTS = (kLGE == 4) ? 1 : 2;      // thread multiplier: 8 keys per thread -> twice the threads
constexpr int kLGEv = 3, kTSv = 2;             // voxel-bucket sorts measured faster with 8 keys per thread
constexpr int kPersistCUs = 256;  // MI355X: 256 CUs; list-driven kernels launch a few workgroups per CU
constexpr int kSortCapS = 1024, kSortThreadsS = 64;
constexpr int kSortCapL = 8192, kSortThreadsL = 512;
constexpr int kSortCapXL = 16384, kClassXL = 52;  // n >= 8192 (pw_size_class)
constexpr size_t sort_lds_bytes(int cap) { return (size_t)(cap + cap / 8) * 8; }
constexpr int kVoxCapS = 1024, kVoxThreadsS = 64;
constexpr int kVoxCapL = 8192, kVoxThreadsL = 512;
constexpr size_t vox_lds_bytes(int cap) { return (size_t)(cap + cap / 8) * 8 + (size_t)cap * 8 + 128; }

void launch_process(const DevParams& P, const Arena& A, hipStream_t st, int do_patchwork, int apply_filter,
                    int do_voxels, TimerHook th, void* tu) {
    const int B = A.n_scans;
    if (B <= 0) return;
    if (do_patchwork == 1) {
        hipMemsetAsync(A.patch_count, 0, sizeof(int32_t) * (size_t)B * kMaxPatches, st);
        hipMemsetAsync(A.patch_cursor, 0, sizeof(int32_t) * (size_t)B * kMaxPatches, st);
        dim3 gcls((A.max_scan_pts + kClsThreads * kClsItems - 1) / (kClsThreads * kClsItems), B);
        TH_BEGIN("pw_classify");
        hipLaunchKernelGGL(k_pw_classify, gcls, dim3(kClsThreads), 0, st, P, A);
        TH_END("pw_classify");
        TH_BEGIN("pw_offsets");
        hipLaunchKernelGGL(k_pw_offsets, dim3(B), dim3(1024), 0, st, P, A);
        TH_END("pw_offsets");
        TH_BEGIN("pw_scatter");
        hipLaunchKernelGGL(k_pw_scatter, gcls, dim3(kClsThreads), 0, st, P, A);
        TH_END("pw_scatter");
        hipMemsetAsync(A.order_hist, 0, sizeof(int32_t) * 64, st);
        hipMemsetAsync(A.order_cursor, 0, sizeof(int32_t) * 64, st);
        const int n_all = B * P.n_patches;
        TH_BEGIN("pw_order");
        hipLaunchKernelGGL(k_pw_order_count, dim3((n_all + 255) / 256), dim3(256), 0, st, P, A);
        hipLaunchKernelGGL(k_pw_order_offsets, dim3(1), dim3(64), 0, st, A);
        hipLaunchKernelGGL(k_pw_order_scatter, dim3((n_all + 255) / 256), dim3(256), 0, st, P, A);
        TH_END("pw_order");
        // patches of more than 8192 points (the rings next to a 128-beam sensor: up to 11 k points, a few per scan) have their own
        // tier -- 16 384 keys in 144 KB of LDS, one workgroup per CU -- instead of the in-place sort in global memory that
        // remains for anything larger still
        hipFuncSetAttribute((const void*)k_pw_sort<kSortCapXL, 1024, kClassXL, 63, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sort_lds_bytes(kSortCapXL));
        hipFuncSetAttribute((const void*)k_pw_sort<kSortCapL, kSortThreadsL * kTS, kClassL, kClassXL - 1, kLGE>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds_bytes(kSortCapL));
        TH_BEGIN("pw_sort_8192");  // one timing label per kernel; largest first: their tail overlaps the smaller tiers' launches
        hipLaunchKernelGGL((k_pw_sort<kSortCapXL, 1024, kClassXL, 63, 4>), dim3(kPersistCUs), dim3(1024), sort_lds_bytes(kSortCapXL), st, P, A);
        hipLaunchKernelGGL((k_pw_sort<kSortCapL, kSortThreadsL * kTS, kClassL, kClassXL - 1, kLGE>), dim3(kPersistCUs * 2),
                           dim3(kSortThreadsL * kTS), sort_lds_bytes(kSortCapL), st, P, A);
        TH_END("pw_sort_8192");
        TH_BEGIN("pw_sort_4096");
        hipLaunchKernelGGL((k_pw_sort<4096, 256 * kTS, kClassM2, kClassL - 1, kLGE>), dim3(kPersistCUs * 4), dim3(256 * kTS),
                           sort_lds_bytes(4096), st, P, A);
        TH_END("pw_sort_4096");
        TH_BEGIN("pw_sort_2048");
        hipLaunchKernelGGL((k_pw_sort<2048, 128 * kTS, kClassM, kClassM2 - 1, kLGE>), dim3(kPersistCUs * 8), dim3(128 * kTS),
                           sort_lds_bytes(2048), st, P, A);
        TH_END("pw_sort_2048");
        TH_BEGIN("pw_sort_1024");
        hipLaunchKernelGGL((k_pw_sort<kSortCapS, kSortThreadsS * kTS, kClassXS, kClassM - 1, kLGE>), dim3(kPersistCUs * 16),
                           dim3(kSortThreadsS * kTS), sort_lds_bytes(kSortCapS), st, P, A);
        TH_END("pw_sort_1024");
        TH_BEGIN("pw_sort_256");
        hipLaunchKernelGGL((k_pw_sort<256, 64, kClassWave, kClassXS - 1, 3>), dim3(kPersistCUs * 32), dim3(64),
                           sort_lds_bytes(256), st, P, A);
        TH_END("pw_sort_256");
        TH_BEGIN("pw_sort_wave");
        hipLaunchKernelGGL(k_pw_sort_wave, dim3(kPersistCUs * 8), dim3(256), 0, st, P, A);
        TH_END("pw_sort_wave");
        // Throughput (a sequence shard) wants the 16-lane fit only where a lane per patch starves the chip (>= 512
        // points); a handful of scans (the per-scan host API) has too few patches to fill it either way, and the lane-per-
        // patch chain of a 500-point patch is then the latency: everything from 64 points up goes to the 16-lane kernel.
        const int coop_class = (B <= 8) ? kClassWave : kClassFitCoop;
        const int coop_min = 1 << (coop_class / 4);
        TH_BEGIN("pw_fit_large");
        if (B <= 8)
            hipLaunchKernelGGL(k_pw_fit_coop<64>, dim3((int)(A.total_pts / coop_min) + 1), dim3(64), 0, st, P, A, coop_class);
        else
            hipLaunchKernelGGL(k_pw_fit_coop<16>, dim3((int)(A.total_pts / coop_min / 4) + 1), dim3(64), 0, st, P, A, coop_class);
        TH_END("pw_fit_large");
        TH_BEGIN("pw_fit");
        hipLaunchKernelGGL(k_pw_fit, dim3((n_all + 63) / 64), dim3(64), 0, st, P, A, coop_class);
        TH_END("pw_fit");
        TH_BEGIN("pw_arrange");
        hipLaunchKernelGGL((k_pw_arrange<64, kClassWave, 63>), dim3(kPersistCUs * 8), dim3(256), 0, st, P, A);
        hipLaunchKernelGGL((k_pw_arrange<16, 0, kClassWave - 1>), dim3(kPersistCUs * 4), dim3(256), 0, st, P, A);
        TH_END("pw_arrange");
        TH_BEGIN("emit_offsets");
        hipLaunchKernelGGL(k_emit_offsets, dim3(B), dim3(1024), 0, st, P, A);
        TH_END("emit_offsets");
        TH_BEGIN("emit");
        hipLaunchKernelGGL(k_emit, dim3(kPersistCUs * 8), dim3(kEmitThreads), 0, st, P, A);
        TH_END("emit");
    } else if (do_patchwork == 0) {
        TH_BEGIN("bin_direct");
        hipLaunchKernelGGL(k_bin_direct, dim3(B), dim3(1024), 0, st, P, A, apply_filter);
        TH_END("bin_direct");
    }
    if (do_voxels) {
        if (do_patchwork == 2) {
            hipMemsetAsync(A.scan_irr, 0, sizeof(int32_t) * (size_t)B, st);  // (order hint of the clustering: unknown here)
            hipLaunchKernelGGL(k_apri_split, dim3((A.max_scan_pts + 2047) / 2048, B), dim3(256), 0, st, A);
        }
        hipMemsetAsync(A.vb_count, 0, sizeof(int32_t) * (size_t)B * kMaxBuckets, st);
        hipMemsetAsync(A.vb_cursor, 0, sizeof(int32_t) * (size_t)B * kMaxBuckets, st);
        dim3 gv((A.max_scan_pts + kVxThreads * kVxItems - 1) / (kVxThreads * kVxItems), B);
        TH_BEGIN("vx_count");
        hipLaunchKernelGGL(k_vx_count, gv, dim3(kVxThreads), 0, st, P, A);
        TH_END("vx_count");
        TH_BEGIN("vx_offsets");
        hipLaunchKernelGGL(k_vx_offsets, dim3(B), dim3(1024), 0, st, P, A);
        TH_END("vx_offsets");
        TH_BEGIN("vx_scatter");
        hipLaunchKernelGGL(k_vx_scatter, gv, dim3(kVxThreads), 0, st, P, A);
        TH_END("vx_scatter");
        dim3 gb(P.n_buckets, B);
        hipMemsetAsync(A.vorder_hist, 0, sizeof(int32_t) * 64, st);
        hipMemsetAsync(A.vorder_cursor, 0, sizeof(int32_t) * 64, st);
        const int nb_all = B * P.n_buckets;
        TH_BEGIN("vx_order");
        hipLaunchKernelGGL(k_vx_order_count, dim3((nb_all + 255) / 256), dim3(256), 0, st, P, A);
        hipLaunchKernelGGL(k_vx_order_offsets, dim3(1), dim3(64), 0, st, A);
        hipLaunchKernelGGL(k_vx_order_scatter, dim3((nb_all + 255) / 256), dim3(256), 0, st, P, A);
        TH_END("vx_order");
        if (do_patchwork == 3) {  // VoxelGrid run: centroids instead of intensity statistics
            hipFuncSetAttribute((const void*)k_vx_bucket<kVoxCapL, kVoxThreadsL * kTSv, kClassL, 63, kLGEv, 1>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)vox_lds_bytes(kVoxCapL));
            hipFuncSetAttribute((const void*)k_vx_bucket<4096, 256 * kTSv, kClassM2, kClassL - 1, kLGEv, 1>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)vox_lds_bytes(4096));
            hipLaunchKernelGGL((k_vx_bucket<kVoxCapL, kVoxThreadsL * kTSv, kClassL, 63, kLGEv, 1>), dim3(kPersistCUs), dim3(kVoxThreadsL * kTSv),
                               vox_lds_bytes(kVoxCapL), st, P, A);
            hipLaunchKernelGGL((k_vx_bucket<4096, 256 * kTSv, kClassM2, kClassL - 1, kLGEv, 1>), dim3(kPersistCUs * 2), dim3(256 * kTSv),
                               vox_lds_bytes(4096), st, P, A);
            hipLaunchKernelGGL((k_vx_bucket<2048, 128 * kTSv, kClassM, kClassM2 - 1, kLGEv, 1>), dim3(kPersistCUs * 4), dim3(128 * kTSv),
                               vox_lds_bytes(2048), st, P, A);
            hipLaunchKernelGGL((k_vx_bucket<kVoxCapS, kVoxThreadsS * kTSv, kClassXS, kClassM - 1, kLGEv, 1>), dim3(kPersistCUs * 8),
                               dim3(kVoxThreadsS * kTSv), vox_lds_bytes(kVoxCapS), st, P, A);
            hipLaunchKernelGGL((k_vx_bucket<256, 64, 0, kClassXS - 1, 3, 1>), dim3(kPersistCUs * 32), dim3(64), vox_lds_bytes(256), st,
                               P, A);
            hipLaunchKernelGGL(k_vx_final_offsets, dim3(B), dim3(1024), 0, st, P, A);
            return;
        }
        auto bucket_tiers = [&](auto kt) {
            using KT = decltype(kt);
            hipFuncSetAttribute((const void*)k_vx_bucket<kVoxCapL, kVoxThreadsL * kTSv, kClassL, 63, kLGEv, 0, KT>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)vox_lds_bytes(kVoxCapL));
            hipFuncSetAttribute((const void*)k_vx_bucket<4096, 256 * kTSv, kClassM2, kClassL - 1, kLGEv, 0, KT>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)vox_lds_bytes(4096));
            TH_BEGIN("vx_bucket_8192");
            hipLaunchKernelGGL((k_vx_bucket<kVoxCapL, kVoxThreadsL * kTSv, kClassL, 63, kLGEv, 0, KT>), dim3(kPersistCUs), dim3(kVoxThreadsL * kTSv),
                               vox_lds_bytes(kVoxCapL), st, P, A);
            TH_END("vx_bucket_8192");
            TH_BEGIN("vx_bucket_4096");
            hipLaunchKernelGGL((k_vx_bucket<4096, 256 * kTSv, kClassM2, kClassL - 1, kLGEv, 0, KT>), dim3(kPersistCUs * 2), dim3(256 * kTSv),
                               vox_lds_bytes(4096), st, P, A);
            TH_END("vx_bucket_4096");
            TH_BEGIN("vx_bucket_2048");
            hipLaunchKernelGGL((k_vx_bucket<2048, 128 * kTSv, kClassM, kClassM2 - 1, kLGEv, 0, KT>), dim3(kPersistCUs * 4), dim3(128 * kTSv),
                               vox_lds_bytes(2048), st, P, A);
            TH_END("vx_bucket_2048");
            TH_BEGIN("vx_bucket_1024");
            hipLaunchKernelGGL((k_vx_bucket<kVoxCapS, kVoxThreadsS * kTSv, kClassXS, kClassM - 1, kLGEv, 0, KT>), dim3(kPersistCUs * 8),
                               dim3(kVoxThreadsS * kTSv), vox_lds_bytes(kVoxCapS), st, P, A);
            TH_END("vx_bucket_1024");
            TH_BEGIN("vx_bucket_256");
            hipLaunchKernelGGL((k_vx_bucket<256, 64, 0, kClassXS - 1, 3, 0, KT>), dim3(kPersistCUs * 32), dim3(64), vox_lds_bytes(256), st, P, A);
            TH_END("vx_bucket_256");
        };
        if (A.vx_k32)
            bucket_tiers((uint32_t)0);
        else
            bucket_tiers((unsigned long long)0);
        TH_BEGIN("vx_final_offsets");
        hipLaunchKernelGGL(k_vx_final_offsets, dim3(B), dim3(1024), 0, st, P, A);
        TH_END("vx_final_offsets");
        TH_BEGIN("vx_final");
        hipLaunchKernelGGL(k_vx_final, gb, dim3(256), 0, st, P, A);
        TH_END("vx_final");
    }
}

void launch_apri_expand(const DevParams& P, const Arena& A, int s0, int n_scans, int max_pts, hipStream_t st) {
    if (n_scans <= 0 || max_pts <= 0) return;
    hipLaunchKernelGGL(k_apri_expand, dim3((max_pts + 1023) / 1024, n_scans), dim3(256), 0, st, P, A, s0);
}

void launch_voxelgrid_keys(const Arena& A, const VgJob& J, hipStream_t st) {
    const int B = A.n_scans;
    if (B <= 0) return;
    hipMemsetAsync(A.vg_range, 0, sizeof(int32_t), st);
    hipLaunchKernelGGL(k_vg_minmax, dim3(B), dim3(1024), 0, st, A, J);
    if (A.max_scan_pts > 0)
        hipLaunchKernelGGL(k_vg_keys, dim3((A.max_scan_pts + 2047) / 2048, B), dim3(256), 0, st, A, J);
}
void launch_voxelgrid_lut(const Arena& A, hipStream_t st) {
    if (A.n_scans <= 0 || A.max_scan_pts <= 0) return;
    hipFuncSetAttribute((const void*)k_vg_lut, hipFuncAttributeMaxDynamicSharedMemorySize, kVgLutBins * (int)sizeof(int));
    hipLaunchKernelGGL(k_vg_lut, dim3(A.n_scans), dim3(1024), kVgLutBins * sizeof(int), st, A, (int32_t*)A.vb_lut_shift);
}
void launch_voxelgrid_gather(const DevParams& P, const Arena& A, const VgJob& J, long long out_capacity, hipStream_t st) {
    if (A.n_scans <= 0 || A.max_scan_pts <= 0) return;
    hipLaunchKernelGGL(k_vg_outoff, dim3(1), dim3(1024), 0, st, A);
    hipLaunchKernelGGL(k_vg_final, dim3(P.n_buckets, A.n_scans), dim3(256), 0, st, A, J, out_capacity);
}

void launch_cls(const Arena& A, int s, size_t scan_base, int n_points, hipStream_t st) {
    if (n_points <= 0) return;
    hipMemsetAsync(A.cls + scan_base, SCVOD_CLS_DROPPED, (size_t)n_points, st);
    hipLaunchKernelGGL(k_cls_from_lists, dim3((n_points + 2047) / 2048), dim3(256), 0, st, A, s);
}

void launch_cluster(const DevParams& P, const Arena& A, int from_apri, hipStream_t st, TimerHook th, void* tu) {
    hipMemsetAsync(A.cc_stats, 0, 4 * sizeof(int32_t), st);
    const int B = A.n_scans;
    if (B <= 0 || A.max_scan_pts <= 0) return;
    hipFuncSetAttribute((const void*)k_cc_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCcLdsBytes);
    TH_BEGIN("cc_scan");
    hipLaunchKernelGGL(k_cc_order, dim3(1), dim3(1024), 0, st, A);
    hipLaunchKernelGGL(k_cc_scan, dim3(B), dim3(kCcThreads), kCcLdsBytes, st, P, A, from_apri);
    TH_END("cc_scan");
}

void launch_track(const DevParams& P, const Arena& A, const TrackJob& J, int batch_mode, hipStream_t st,
                  TimerHook th, void* tu) {
    if (J.n_pts > 0) {
        TH_BEGIN("track_probe");
        if (batch_mode && J.pair_pt_begin && !J.next_labels) {
            const int bx = (J.max_pair_pts + 1023) / 1024;  // four points per thread amortise the key staging
            hipLaunchKernelGGL(k_track_probe_pair, dim3(bx > 0 ? bx : 1, J.n_pairs), dim3(256), 0, st, P, A, J, batch_mode);
        } else {
            hipLaunchKernelGGL(k_track_probe, dim3((J.n_pts + 255) / 256), dim3(256), 0, st, P, A, J, batch_mode);
        }
        TH_END("track_probe");
    }
    if (J.n_clusters > 0) {
        TH_BEGIN("track_unique");
        hipLaunchKernelGGL(k_track_unique_bits, dim3(J.n_clusters), dim3(128), 0, st, A, J, batch_mode);
        // tables with more than 262144 voxels (never the case for the reference's grids) take the sort path
        const int table_words = batch_mode ? (A.max_scan_pts + 31) / 32 : (J.n_next_vox + 31) / 32;
        if (table_words > kTrackBitWords)
            hipLaunchKernelGGL((k_track_unique<8192, 256>), dim3(J.n_clusters), dim3(256), 8192 * 4 + 64, st, J, table_words);
        TH_END("track_unique");
    }
}

// ---- uniform-grid correspondence search ------------------------------------------------------------
// Map points are hashed by cell (edge h >= radius) into a CSR table; a query probes the 27 cells around it.
// If the best candidate is within h it is the true nearest neighbour (anything outside the 27 cells is farther
// than h); otherwise the query goes to the exact brute-force kernel.  Distances and tie-breaking (lowest map
// index) are those of the brute-force kernel, so both paths return identical results.
struct NnGrid {
    float ox, oy, oz, inv_h, h2;
    uint32_t mask;  // buckets - 1 (power of two)
};
__device__ __forceinline__ uint32_t nn_bucket(const NnGrid& g, int cx, int cy, int cz) {
    return ((uint32_t)cx * 73856093u ^ (uint32_t)cy * 19349663u ^ (uint32_t)cz * 83492791u) & g.mask;
}
__device__ __forceinline__ void nn_cell(const NnGrid& g, float x, float y, float z, int& cx, int& cy, int& cz) {
    cx = (int)floorf((x - g.ox) * g.inv_h);
    cy = (int)floorf((y - g.oy) * g.inv_h);
    cz = (int)floorf((z - g.oz) * g.inv_h);
}

__global__ __launch_bounds__(256) void k_nn_count(NnGrid g, const float* __restrict__ map_xyz, int n_map, int* count) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_map) return;
    int cx, cy, cz;
    nn_cell(g, map_xyz[3 * (size_t)i], map_xyz[3 * (size_t)i + 1], map_xyz[3 * (size_t)i + 2], cx, cy, cz);
    atomicAdd(&count[nn_bucket(g, cx, cy, cz)], 1);
}

// exclusive scan of `n` ints, three launches: per-block (1024) scans + block totals, scan of totals, add-back
__global__ __launch_bounds__(1024) void k_scan_blocks(const int* in, int* out, int* block_tot, int n) {
    __shared__ int wsum[17];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int v = (i < n) ? in[i] : 0;
    int total;
    const int ex = block_excl_scan<1024>(v, total, wsum);
    if (i < n) out[i] = ex;
    if (threadIdx.x == 0) block_tot[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void k_scan_totals(int* block_tot, int nb, int* grand_total) {
    __shared__ int wsum[17];
    int run = 0;
    for (int c0 = 0; c0 < nb; c0 += 1024) {
        const int i = c0 + threadIdx.x;
        const int v = (i < nb) ? block_tot[i] : 0;
        int total;
        const int ex = block_excl_scan<1024>(v, total, wsum);
        if (i < nb) block_tot[i] = run + ex;
        run += total;
    }
    if (threadIdx.x == 0) *grand_total = run;
}
__global__ __launch_bounds__(1024) void k_scan_add(int* out, const int* block_tot, int n) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) out[i] += block_tot[blockIdx.x];
}

__global__ __launch_bounds__(256) void k_nn_fill(NnGrid g, const float* __restrict__ map_xyz, int n_map, const int* start,
                                                 int* cursor, int* entries) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_map) return;
    int cx, cy, cz;
    nn_cell(g, map_xyz[3 * (size_t)i], map_xyz[3 * (size_t)i + 1], map_xyz[3 * (size_t)i + 2], cx, cy, cz);
    const uint32_t b = nn_bucket(g, cx, cy, cz);
    entries[start[b] + atomicAdd(&cursor[b], 1)] = i;
}

__global__ __launch_bounds__(256) void k_nn_query(NnGrid g, const float* __restrict__ map_xyz, const float* __restrict__ q_xyz,
                                                  int n_q, float r2, const int* start, const int* count, const int* entries,
                                                  int32_t* nn_idx, float* nn_sq, uint8_t* within, int* todo, int* n_todo, int bounded) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n_q) return;
    const float qx = q_xyz[3 * (size_t)q], qy = q_xyz[3 * (size_t)q + 1], qz = q_xyz[3 * (size_t)q + 2];
    int cx, cy, cz;
    nn_cell(g, qx, qy, qz, cx, cy, cz);
    float best = 0.f;
    int bi = -1;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const uint32_t b = nn_bucket(g, cx + dx, cy + dy, cz + dz);
                const int s0 = start[b], c = count[b];
                for (int k = 0; k < c; ++k) {
                    const int m = entries[s0 + k];
                    const float ex = map_xyz[3 * (size_t)m] - qx, ey = map_xyz[3 * (size_t)m + 1] - qy,
                                ez = map_xyz[3 * (size_t)m + 2] - qz;
                    const float d = (ex * ex + ey * ey) + ez * ez;
                    if (bi < 0 || d < best || (d == best && m < bi)) {
                        best = d;
                        bi = m;
                    }
                }
            }
    // several of the 27 probes may hash to the same bucket: harmless (same candidates again)
    if (bi >= 0 && best <= g.h2) {
        nn_idx[q] = bi;
        nn_sq[q] = best;
        within[q] = best < r2 ? 1 : 0;
    } else if (bounded) {
        // radius search (pcl radiusSearch, evaluate.cpp:95,104): the cell edge is >= radius, so a neighbour inside the radius
        // would have been among the candidates; accepted above whenever it is closer than 0.99 cell edges
        const bool in = bi >= 0 && best < r2;
        nn_idx[q] = in ? bi : -1;
        nn_sq[q] = in ? best : __builtin_huge_valf();
        within[q] = in ? 1 : 0;
    } else {
        todo[atomicAdd(n_todo, 1)] = q;  // exact answer needs the whole map
    }
}

__global__ __launch_bounds__(kNnThreads) void k_nn_brute_list(const float* __restrict__ map_xyz, int n_map,
                                                               const float* __restrict__ q_xyz, const int* todo,
                                                               const int* n_todo, float r2, int32_t* nn_idx, float* nn_sq,
                                                               uint8_t* within) {
    __shared__ float tx[kNnTile], ty[kNnTile], tz[kNnTile];
    const int nt = *n_todo;
    for (int t0q = blockIdx.x * kNnThreads; t0q < nt; t0q += gridDim.x * kNnThreads) {
        const int t = t0q + threadIdx.x;
        const int q = (t < nt) ? todo[t] : -1;
        float qx = 0, qy = 0, qz = 0;
        if (q >= 0) {
            qx = q_xyz[3 * (size_t)q];
            qy = q_xyz[3 * (size_t)q + 1];
            qz = q_xyz[3 * (size_t)q + 2];
        }
        float best = 0.f;
        int bi = -1;
        for (int m0 = 0; m0 < n_map; m0 += kNnTile) {
            const int tn = min(kNnTile, n_map - m0);
            __syncthreads();
            for (int j = threadIdx.x; j < tn; j += kNnThreads) {
                tx[j] = map_xyz[3 * (size_t)(m0 + j)];
                ty[j] = map_xyz[3 * (size_t)(m0 + j) + 1];
                tz[j] = map_xyz[3 * (size_t)(m0 + j) + 2];
            }
            __syncthreads();
            if (q >= 0) {
                for (int j = 0; j < tn; ++j) {
                    const float dx = tx[j] - qx, dy = ty[j] - qy, dz = tz[j] - qz;
                    const float d = (dx * dx + dy * dy) + dz * dz;
                    if (bi < 0 || d < best) {
                        best = d;
                        bi = m0 + j;
                    }
                }
            }
        }
        if (q >= 0) {
            nn_idx[q] = bi;
            nn_sq[q] = best;
            within[q] = (bi >= 0 && best < r2) ? 1 : 0;
        }
    }
}

// work: ints, size >= 3 * buckets + n_map + n_q + 4 + (buckets / 1024 + 1)
void launch_nn(const float* map_xyz, int32_t n_map, const float* q_xyz, int32_t n_q, float radius, int32_t* nn_idx,
               float* nn_sq, uint8_t* within, const float origin[3], float cell, int32_t buckets, int* work, int bounded,
               hipStream_t st) {
    if (n_q <= 0) return;
    if (n_map <= 0 || buckets <= 0) {  // empty map: brute-force kernel writes idx -1
        hipLaunchKernelGGL(k_nn_brute, dim3((n_q + kNnThreads - 1) / kNnThreads), dim3(kNnThreads), 0, st, map_xyz, n_map,
                           q_xyz, n_q, radius * radius, nn_idx, nn_sq, within);
        return;
    }
    NnGrid g;
    g.ox = origin[0];
    g.oy = origin[1];
    g.oz = origin[2];
    g.inv_h = 1.0f / cell;
    g.h2 = (0.99f * cell) * (0.99f * cell);  // acceptance radius, a hair inside the cell edge (cell rounding)
    g.mask = (uint32_t)buckets - 1u;
    int* count = work;
    int* start = count + buckets;
    int* cursor = start + buckets;
    int* entries = cursor + buckets;
    int* todo = entries + n_map;
    int* n_todo = todo + n_q;
    int* grand = n_todo + 1;
    int* block_tot = grand + 1;
    const int nb = (buckets + 1023) / 1024;
    hipMemsetAsync(count, 0, sizeof(int) * (size_t)buckets, st);
    hipMemsetAsync(cursor, 0, sizeof(int) * (size_t)buckets, st);
    hipMemsetAsync(n_todo, 0, sizeof(int), st);
    hipLaunchKernelGGL(k_nn_count, dim3((n_map + 255) / 256), dim3(256), 0, st, g, map_xyz, n_map, count);
    hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(1024), 0, st, count, start, block_tot, buckets);
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, st, block_tot, nb, grand);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(1024), 0, st, start, block_tot, buckets);
    hipLaunchKernelGGL(k_nn_fill, dim3((n_map + 255) / 256), dim3(256), 0, st, g, map_xyz, n_map, start, cursor, entries);
    hipLaunchKernelGGL(k_nn_query, dim3((n_q + 255) / 256), dim3(256), 0, st, g, map_xyz, q_xyz, n_q, radius * radius, start,
                       count, entries, nn_idx, nn_sq, within, todo, n_todo, bounded);
    if (bounded) return;
    hipLaunchKernelGGL(k_nn_brute_list, dim3(kPersistCUs * 2), dim3(kNnThreads), 0, st, map_xyz, n_map, q_xyz, todo, n_todo,
                       radius * radius, nn_idx, nn_sq, within);
}

}  // namespace scvod

#ifdef SCVOD_PROFILE
// development build only (make prof): the summed phase clocks, 100 MHz ticks, [8 kernels][16 phases]; reading resets them
extern "C" int scvod_debug_profile(unsigned long long* out128) {
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out128, HIP_SYMBOL(scvod::g_prof), sizeof(unsigned long long) * 128) != hipSuccess) return -1;
    unsigned long long z[128] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(scvod::g_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
