// scvod_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the SCV-OD hot path.
//
// Stage map (reference file:line -> kernel):
//   PatchWork::estimate_ground prologue + pc2czm   patchwork.h:277-325,416-459 -> k_pw_classify, k_pw_offsets, k_pw_scatter,
//                                                                                  k_pw_order_*, k_pw_sort_wave, k_pw_sort<...>
//   extract_piecewiseground / seeds / plane fit    patchwork.h:217-268,463-504 -> k_pw_fit_coop<16|64>, k_pw_fit
//   gating + emission order                        patchwork.h:326-391         -> fit_status / fit_finish, arrange_group (in k_pw_fit_coop), k_pw_arrange, k_emit_offsets, k_emit
//   SSC::makeApriVec                               ssc.cpp:155-195             -> k_emit (fused, compact apri_vec), k_apri_expand,
//                                                                                  k_bin_direct
//   SSC::makeHashCloud                             ssc.cpp:253-289             -> k_vx_partition (count + offsets + scatter per scan), k_vx_order_*,
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

#include "scvod_k_patchwork.inc"  // Patchwork: classify / scatter / sort tiers / plane fit / arrange / ordered emission with the fused curved-voxel binning (A1-A4)
#include "scvod_k_voxelgrid.inc"  // loader-side label filter + pcl::VoxelGrid
#include "scvod_k_voxels.inc"  // PointAPRI expansion, direct binning, the voxel stage: buckets, LDS sort, per-voxel descriptors (A4-A5)
#include "scvod_k_cluster.inc"  // curved-voxel clustering, boxes, type rules, successor tables: k_cc_scan
#include "scvod_k_probe_nn.inc"  // per-pair tracking probe of the facade path (A6 bulk part) and the brute-force correspondence search (A7)
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
#ifndef SCVOD_VOX_LGE
#define SCVOD_VOX_LGE 3
#endif
constexpr int kLGEv = SCVOD_VOX_LGE, kTSv = (kLGEv == 4) ? 1 : 2;  // voxel-bucket sorts measured faster with 8 keys per thread
constexpr int kPersistCUs = 256;  // MI355X: 256 CUs; list-driven kernels launch a few workgroups per CU
constexpr int kSortCapS = 1024, kSortThreadsS = 64;
constexpr int kSortCapL = 8192, kSortThreadsL = 512;
constexpr int kSortCapXL = 16384, kClassXL = 52;  // n >= 8192 (pw_size_class)
constexpr size_t sort_lds_bytes(int cap) { return (size_t)(cap + cap / 8) * 8; }
constexpr int kVoxCapS = 1024, kVoxThreadsS = 64;
constexpr int kVoxCapL = 8192, kVoxThreadsL = 512;

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
        // (patches of the 16-lane fit are arranged by the lanes that fitted them; what is left: the lane-per-patch fit's)
        if (coop_class > kClassWave) hipLaunchKernelGGL((k_pw_arrange<64>), dim3(kPersistCUs * 8), dim3(256), 0, st, P, A, kClassWave, coop_class - 1);
        hipLaunchKernelGGL((k_pw_arrange<16>), dim3(kPersistCUs * 4), dim3(256), 0, st, P, A, 0, kClassWave - 1);
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
            hipMemsetAsync(A.irr_list, 0xff, sizeof(int32_t) * (size_t)B * (kIrrListCap + 1), st);  // (-1: irregular points not listed)
            hipLaunchKernelGGL(k_apri_split, dim3((A.max_scan_pts + 2047) / 2048, B), dim3(256), 0, st, A);
        }
        TH_BEGIN("vx_partition");
        hipLaunchKernelGGL(k_vx_partition, dim3(B), dim3(1024), 0, st, P, A);
        TH_END("vx_partition");
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
            constexpr size_t KS = sizeof(KT);
            constexpr int kMore = (KS == 4 && SCVOD_VOX_K32_LDS) ? 3 : 2;  // workgroups per CU of the 4096 tier (x2, x4 for the next two)
            hipFuncSetAttribute((const void*)k_vx_bucket<kVoxCapL, kVoxThreadsL * kTSv, kClassL, 63, kLGEv, 0, KT>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)vox_lds_bytes(kVoxCapL, KS));
            hipFuncSetAttribute((const void*)k_vx_bucket<4096, 256 * kTSv, kClassM2, kClassL - 1, kLGEv, 0, KT>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)vox_lds_bytes(4096, KS));
            TH_BEGIN("vx_bucket_8192");
            hipLaunchKernelGGL((k_vx_bucket<kVoxCapL, kVoxThreadsL * kTSv, kClassL, 63, kLGEv, 0, KT>), dim3(kPersistCUs), dim3(kVoxThreadsL * kTSv),
                               vox_lds_bytes(kVoxCapL, KS), st, P, A);
            TH_END("vx_bucket_8192");
            TH_BEGIN("vx_bucket_4096");
            hipLaunchKernelGGL((k_vx_bucket<4096, 256 * kTSv, kClassM2, kClassL - 1, kLGEv, 0, KT>), dim3(kPersistCUs * kMore), dim3(256 * kTSv),
                               vox_lds_bytes(4096, KS), st, P, A);
            TH_END("vx_bucket_4096");
            TH_BEGIN("vx_bucket_2048");
            hipLaunchKernelGGL((k_vx_bucket<2048, 128 * kTSv, kClassM, kClassM2 - 1, kLGEv, 0, KT>), dim3(kPersistCUs * kMore * 2), dim3(128 * kTSv),
                               vox_lds_bytes(2048, KS), st, P, A);
            TH_END("vx_bucket_2048");
            TH_BEGIN("vx_bucket_1024");
            hipLaunchKernelGGL((k_vx_bucket<kVoxCapS, kVoxThreadsS * kTSv, kClassXS, kClassM - 1, kLGEv, 0, KT>), dim3(kPersistCUs * 8),
                               dim3(kVoxThreadsS * kTSv), vox_lds_bytes(kVoxCapS, KS), st, P, A);  // (twelve per CU measured slower than eight)
            TH_END("vx_bucket_1024");
            TH_BEGIN("vx_bucket_256");
            hipLaunchKernelGGL((k_vx_bucket<256, 64, 0, kClassXS - 1, 3, 0, KT>), dim3(kPersistCUs * 32), dim3(64), vox_lds_bytes(256, KS), st, P, A);
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
        hipLaunchKernelGGL(k_vx_final, dim3((P.n_buckets + 3) / 4, B), dim3(256), 0, st, P, A);
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

void launch_cluster(const DevParams& P, const Arena& A0, int from_apri, hipStream_t st, TimerHook th, void* tu) {
    Arena A = A0;
    hipMemsetAsync(A.cc_stats, 0, 8 * sizeof(int32_t), st);
    const int B = A.n_scans;
    if (B <= 0 || A.max_scan_pts <= 0) return;
    // k_cc_scan hands the scans with an unsettled irregular run over to k_cc_exact (cc_again, counted on the device); helper blocks there
    // unless the ctx asks for every such scan's workgroup to work alone (scvod_set_cluster_exact(ctx, 3))
    A.cc_help_blocks = A.cc_help_blocks_wanted > 0 ? 1 : 0;
    hipMemsetAsync(A.cc_again, 0, sizeof(int32_t), st);
    hipMemsetAsync(A.cc_help, 0, kCcHelpWords * sizeof(int32_t), st);
    hipFuncSetAttribute((const void*)k_cc_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCcLdsBytes);
    hipFuncSetAttribute((const void*)k_cc_exact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCcLdsBytes);
    TH_BEGIN("cc_scan");
    hipLaunchKernelGGL(k_cc_order, dim3(1), dim3(1024), 0, st, A);
    hipLaunchKernelGGL(k_cc_scan, dim3(B), dim3(kCcThreads), kCcLdsBytes, st, P, A, from_apri);
    TH_END("cc_scan");
    TH_BEGIN("cc_exact");
    hipLaunchKernelGGL(k_cc_exact, dim3(kCcExactBlocks), dim3(kCcThreads), kCcLdsBytes, st, P, A, from_apri);
    TH_END("cc_exact");
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

#include "scvod_k_nn_grid.inc"  // uniform-grid correspondence search (A7)
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
