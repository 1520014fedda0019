// scvod_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the SCV-OD hot path.
//
// Stage map (reference file:line -> kernel):
//   PatchWork::estimate_ground prologue + pc2czm   patchwork.h:277-325,416-459 -> k_pw_classify, k_pw_offsets, k_pw_scatter
//   extract_piecewiseground / seeds / plane fit    patchwork.h:217-268,463-504 -> k_pw_patch (one workgroup per patch)
//   gating + emission order                        patchwork.h:326-391         -> k_pw_patch epilogue, k_emit_offsets, k_emit
//   SSC::makeApriVec                               ssc.cpp:155-195             -> k_emit (fused), k_bin_direct
//   SSC::makeHashCloud                             ssc.cpp:253-289             -> k_vx_count, k_vx_offsets, k_vx_scatter, k_vx_bucket, k_vx_final*
//   SSC::tracking bulk part                        ssc.cpp:1274-1321           -> k_track_probe, k_track_unique
//   kd-tree look-ups of evaluate.cpp:79-145        -> k_nn_brute
//
// Design notes: the path is gather/scatter + histogramming + short serial
// fp32 chains; there is no dense contraction, so no MFMA.  Bit-exact parity with the CPU
// restatement requires (a) the z-sort order inside every patch, (b) strictly sequential
// fp32 accumulation of the 9 covariance moments in that order, (c) sequential per-voxel
// intensity sums in point order.  Parallelism therefore comes from patches x scans
// (504 x B workgroups) and voxels x scans, not from tree reductions of those sums.
#include "scvod_kernels.h"

namespace scvod {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v) {
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
__device__ __forceinline__ int block_excl_scan(int v, int& total, int* wsum) {
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

// Normalised bitonic network (every comparator puts the minimum at the lower index), valid
// for any n: indices >= n act as +inf.  Works on LDS or global (flat) storage.
template <int THREADS, typename T>
__device__ __forceinline__ void block_bitonic_sort(T* a, int n) {
    if (n <= 1) return;
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    const int half = np2 >> 1;
    for (int k = 2; k <= np2; k <<= 1) {
        const int hk = k >> 1;
        for (int t = threadIdx.x; t < half; t += THREADS) {
            int blk = t / hk, off = t - blk * hk;
            int i = blk * k + off, j = blk * k + (k - 1 - off);
            if (j < n) {
                T x = a[i], y = a[j];
                if (x > y) {
                    a[i] = y;
                    a[j] = x;
                }
            }
        }
        __syncthreads();
        for (int jj = k >> 2; jj >= 1; jj >>= 1) {
            for (int t = threadIdx.x; t < half; t += THREADS) {
                int i = ((t / jj) * 2 * jj) + (t % jj);
                int j = i + jj;
                if (j < n) {
                    T x = a[i], y = a[j];
                    if (x > y) {
                        a[i] = y;
                        a[j] = x;
                    }
                }
            }
            __syncthreads();
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
#pragma unroll
    for (int it = 0; it < kClsItems; ++it) {
        int i = start + it * kClsThreads + threadIdx.x;
        if (i < n) {
            float4 p = A.pts[base + i];
            int pid = czm_patch_of(P.czm, p.x, p.y, p.z);
            A.pid[base + i] = (int16_t)pid;
            A.cls[base + i] = SCVOD_CLS_DROPPED;
            if (pid >= 0) atomicAdd(&hist[pid], 1);
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
        int i = start + it * kClsThreads + threadIdx.x;
        pid[it] = -1;
        if (i < n) {
            pid[it] = A.pid[base + i];
            zk[it] = float_sort_key(A.pts[base + i].z);
            if (pid[it] >= 0) rank[it] = atomicAdd(&hist[pid[it]], 1);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < P.n_patches; b += kClsThreads) {
        int c = hist[b];
        if (c) hist[b] = A.patch_off[s * (kMaxPatches + 1) + b] + atomicAdd(&A.patch_cursor[s * kMaxPatches + b], c);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kClsItems; ++it) {
        int i = start + it * kClsThreads + threadIdx.x;
        if (i < n && pid[it] >= 0) A.keys[(size_t)base + hist[pid[it]] + rank[it]] = ((uint64_t)zk[it] << 32) | (uint32_t)i;
    }
}

// ------------------------------------------------------------------------------------------
// Patchwork stage 2: one workgroup per (scan, patch).
//   sort by (z, idx) -> seeds -> 3 x {sequential fp32 moments, mean/cov, 3x3 Jacobi SVD,
//   plane distance test} -> gating -> [ground part | non-ground part] + bin-filter bits.
// Two tiers are launched over the same grid: CAP=1024/64 threads (9 WG per CU) takes patches
// with n <= 1024, CAP=8192/512 threads (1 WG per CU) the rest; a patch larger than 8192 points
// runs the same code on global scratch.
// ------------------------------------------------------------------------------------------
struct PatchStore {
    uint64_t* keys;  // sorted in place
    float* x;
    float* y;
    float* z;
    uint32_t* idx;
    uint8_t* mask;
};

template <int THREADS>
__device__ void patch_body(const DevParams& P, const Arena& A, int s, int p, int base, int off, int n, PatchStore S,
                           int* wsum, float* sh /* >= 32 floats */) {
    const int tid = threadIdx.x;
    // ---- sort (z ascending, ties by input index) ----
    block_bitonic_sort<THREADS>(S.keys, n);
    // ---- gather sorted points (x/y alias the key storage in the LDS tiers: extract every idx
    // first, then overwrite) ----
    for (int j = tid; j < n; j += THREADS) {
        uint32_t id = (uint32_t)S.keys[j];
        S.idx[j] = id;
        S.z[j] = A.pts[base + id].z;
    }
    __syncthreads();
    for (int j = tid; j < n; j += THREADS) {
        float4 q = A.pts[base + S.idx[j]];
        S.x[j] = q.x;
        S.y[j] = q.y;
    }
    __syncthreads();

    // zone of this patch
    int zone = 0;
    while (zone < 3 && p >= P.czm.patch_base[zone + 1]) ++zone;
    const int ring = (p - P.czm.patch_base[zone]) / P.czm.num_sectors[zone];
    int concentric_idx = ring;
    for (int k = 0; k < zone; ++k) concentric_idx += P.czm.num_rings[k];

    // ---- extract_initial_seeds_ (patchwork.h:235-268) ----
    if (tid == 0) {
        int init_idx = 0;
        if (zone == 0) {
            while (init_idx < n && (double)S.z[init_idx] < P.czm.seed_margin_z) ++init_idx;
        }
        double sum = 0;
        int cnt = 0;
        for (int i = init_idx; i < n && cnt < P.czm.num_lpr; ++i) {
            sum += (double)S.z[i];
            ++cnt;
        }
        double lpr = cnt != 0 ? sum / cnt : 0.0;
        double thr = lpr + P.czm.th_seeds;
        ((double*)sh)[0] = thr;
    }
    __syncthreads();
    {
        const double thr = ((double*)sh)[0];
        for (int j = tid; j < n; j += THREADS) S.mask[j] = ((double)S.z[j] < thr) ? 1 : 0;
    }
    __syncthreads();

    // persistent plane state of this patch (sh[8..]): cov[9], mean[3], normal[3], sv[3], th_dist_d
    float* st_cov = sh + 8;      // 9
    float* st_mean = sh + 17;    // 3
    float* st_normal = sh + 20;  // 3
    float* st_sv = sh + 23;      // 3
    float* st_thd = sh + 26;     // 1
    if (tid < 19) sh[8 + tid] = 0.f;
    __syncthreads();

    for (int iter = 0; iter < P.czm.num_iter; ++iter) {
        // ---- pcl::computeMeanAndCovarianceMatrix: 9 sequential fp32 chains, lanes 0..8 of
        // wave 0 each own one accumulator; points are broadcast reads from the store ----
        if (tid < 64) {
            const int k = tid;
            // accumulator k multiplies a_k * b_k : (xx, xy, xz, yy, yz, zz, x, y, z)
            const int sa = (k < 3) ? 0 : (k < 5) ? 1 : (k == 5) ? 2 : (k - 6);
            const int sb = (k < 3) ? k : (k < 5) ? (k - 2) : (k == 5) ? 2 : 3;
            float acc = 0.f;
            int cnt = 0;
            if (k < 9) {
                for (int j = 0; j < n; ++j) {
                    float px = S.x[j], py = S.y[j], pz = S.z[j];
                    int m = S.mask[j];
                    float a = (sa == 0) ? px : (sa == 1) ? py : pz;
                    float b = (sb == 0) ? px : (sb == 1) ? py : (sb == 2) ? pz : 1.0f;
                    float term = a * b;
                    if (m) {
                        acc = acc + term;
                        ++cnt;
                    }
                }
            }
            // gather the 9 accumulators on lane 0
            float acc_all[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) acc_all[q] = __shfl(acc, q, 64);
            cnt = __shfl(cnt, 0, 64);
            if (tid == 0) {
                if (cnt != 0) {
                    float fn = (float)cnt;
                    float accu[9];
#pragma unroll
                    for (int q = 0; q < 9; ++q) accu[q] = acc_all[q] / fn;
                    st_mean[0] = accu[6];
                    st_mean[1] = accu[7];
                    st_mean[2] = accu[8];
                    float c00 = accu[0] - accu[6] * accu[6];
                    float c01 = accu[1] - accu[6] * accu[7];
                    float c02 = accu[2] - accu[6] * accu[8];
                    float c11 = accu[3] - accu[7] * accu[7];
                    float c12 = accu[4] - accu[7] * accu[8];
                    float c22 = accu[5] - accu[8] * accu[8];
                    st_cov[0] = c00;
                    st_cov[1] = c01;
                    st_cov[2] = c02;
                    st_cov[3] = c01;
                    st_cov[4] = c11;
                    st_cov[5] = c12;
                    st_cov[6] = c02;
                    st_cov[7] = c12;
                    st_cov[8] = c22;
                }
                float cov[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) cov[q] = st_cov[q];
                Svd3 sv;
                svd3_jacobi(cov, sv);
                float n0 = sv.U[2], n1 = sv.U[5], n2 = sv.U[8];
                st_normal[0] = n0;
                st_normal[1] = n1;
                st_normal[2] = n2;
                st_sv[0] = sv.sv[0];
                st_sv[1] = sv.sv[1];
                st_sv[2] = sv.sv[2];
                float dot = n0 * st_mean[0];
                dot = dot + n1 * st_mean[1];
                dot = dot + n2 * st_mean[2];
                float d = -dot;
                st_thd[0] = (float)(P.czm.th_dist - (double)d);
            }
        }
        __syncthreads();
        // ---- plane distance test on every point of the patch (Eigen GEMV order) ----
        {
            const float n0 = st_normal[0], n1 = st_normal[1], n2 = st_normal[2], thd = st_thd[0];
            for (int j = tid; j < n; j += THREADS) {
                float res = S.x[j] * n0;
                res = res + S.y[j] * n1;
                res = res + S.z[j] * n2;
                S.mask[j] = (res < thd) ? 1 : 0;
            }
        }
        __syncthreads();
    }

    // ---- gating (patchwork.h:339-384) ----
    int status;
    {
        const double ground_z_vec = (double)fabs_f(st_normal[2]);
        const double ground_z_elevation = (double)st_mean[2];
        float svmin = st_sv[0];
        if (st_sv[1] < svmin) svmin = st_sv[1];
        if (st_sv[2] < svmin) svmin = st_sv[2];
        const double surface_variable = (double)(svmin / (st_sv[0] + st_sv[1] + st_sv[2]));
        if (ground_z_vec < P.czm.uprightness_thr) {
            status = 2;
        } else if (concentric_idx < P.czm.num_rings_of_interest) {
            const int e = ring + 2 * zone;
            if (ground_z_elevation > P.czm.elevation_thr[e]) {
                status = (P.czm.flatness_thr[e] > surface_variable) ? 1 : 3;
            } else {
                status = 1;
            }
        } else {
            status = 1;
        }
    }

    // ---- arrange [ground part | non-ground part] keeping the sorted order, attach the
    // makeApriVec range/FOV verdict of every point (used by k_emit for ordered compaction) ----
    int n_g = 0, a_g = 0, a_ng = 0;
    {
        // first pass: count ground
        int run_g = 0, run_ag = 0, run_ang = 0;
        for (int j0 = 0; j0 < n; j0 += THREADS) {
            int j = j0 + tid;
            int g = 0, keep = 0;
            if (j < n) {
                g = S.mask[j];
                Apri a;
                float4 q = A.pts[base + S.idx[j]];
                keep = apri_of_point(P.bin, q.x, q.y, q.z, q.w, a);
            }
            int tg, tk;
            int eg = block_excl_scan<THREADS>(g, tg, wsum);
            int ekg = block_excl_scan<THREADS>(g & keep, tk, wsum);
            int tkn;
            int ekn = block_excl_scan<THREADS>((!g) & keep, tkn, wsum);
            (void)ekg;
            (void)ekn;
            if (j < n) {
                // stash rank and flags: low 2 bits flags, rest = rank among same class
                int rank = g ? (run_g + eg) : ((j - (run_g + eg)));
                S.keys[j] = ((uint64_t)(uint32_t)rank << 2) | (uint64_t)(g ? 1 : 0) | (uint64_t)(keep ? 2 : 0);
            }
            run_g += tg;
            run_ag += tk;
            run_ang += tkn;
            __syncthreads();
        }
        n_g = run_g;
        a_g = run_ag;
        a_ng = run_ang;
    }
    // NOTE: S.keys aliases S.x/S.y in the LDS tiers; x/y are dead from here on.
    for (int j = tid; j < n; j += THREADS) {
        uint64_t v = S.keys[j];
        int g = (int)(v & 1), keep = (int)((v >> 1) & 1);
        int rank = (int)(v >> 2);
        int dst = g ? rank : (n_g + rank);
        A.seg[(size_t)base + off + dst] = S.idx[j] | (keep ? 0x80000000u : 0u);
    }
    if (tid == 0) {
        PatchRec r;
        r.n = n;
        r.n_g = n_g;
        r.status = status;
        r.a_g = a_g;
        r.a_ng = a_ng;
        A.patch_rec[s * kMaxPatches + p] = r;
        scvod_patch_plane pl;
        for (int c = 0; c < 3; ++c) {
            pl.normal[c] = st_normal[c];
            pl.mean[c] = st_mean[c];
            pl.sv[c] = st_sv[c];
        }
        pl.n_pts = n;
        pl.n_ground = n_g;
        pl.status = status;
        A.planes[s * kMaxPatches + p] = pl;
    }
}

template <int CAP, int THREADS, int MIN_N>
__global__ __launch_bounds__(THREADS) void k_pw_patch(DevParams P, Arena A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int p = blockIdx.x, s = blockIdx.y;
    const int n = A.patch_count[s * kMaxPatches + p];
    const int base = A.scan_off[s];
    const int off = A.patch_off[s * (kMaxPatches + 1) + p];
    if (MIN_N == 0 && n <= P.czm.num_min_pts) {
        // skipped patch (patchwork.h:331): record only
        if (threadIdx.x == 0) {
            PatchRec r = {n, 0, 0, 0, 0};
            A.patch_rec[s * kMaxPatches + p] = r;
            scvod_patch_plane pl = {};
            pl.n_pts = n;
            A.planes[s * kMaxPatches + p] = pl;
        }
        return;
    }
    if (n <= MIN_N || n <= P.czm.num_min_pts) return;
    if (MIN_N == 0 && n > CAP) return;  // left to the large tier
    // carve LDS: keys (aliased by x,y) | z | idx | mask | wsum | sh
    uint64_t* l_keys = (uint64_t*)smem;
    float* l_z = (float*)(smem + (size_t)CAP * 8);
    uint32_t* l_idx = (uint32_t*)(smem + (size_t)CAP * 12);
    uint8_t* l_mask = (uint8_t*)(smem + (size_t)CAP * 16);
    int* wsum = (int*)(smem + (size_t)CAP * 17);
    float* sh = (float*)(smem + (size_t)CAP * 17 + 128);
    PatchStore S;
    if (n <= CAP) {
        // stage keys in LDS
        for (int j = threadIdx.x; j < n; j += THREADS) l_keys[j] = A.keys[(size_t)base + off + j];
        __syncthreads();
        S.keys = l_keys;
        S.x = (float*)l_keys;
        S.y = (float*)l_keys + CAP;
        S.z = l_z;
        S.idx = l_idx;
        S.mask = l_mask;
    } else {
        // oversize patch: same algorithm on global storage (rare; correctness path)
        S.keys = A.keys + (size_t)base + off;
        S.x = A.scratch_xyz + 4 * ((size_t)base + off);
        S.y = S.x + n;
        S.z = S.y + n;
        S.idx = (uint32_t*)(S.z + n);
        S.mask = A.scratch_mask + (size_t)base + off;
    }
    patch_body<THREADS>(P, A, s, p, base, off, n, S, wsum, sh);
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
    }
}

constexpr int kEmitThreads = 256;
__global__ __launch_bounds__(kEmitThreads) void k_emit(DevParams P, Arena A) {
    __shared__ int wsum[8];
    const int p = blockIdx.x, s = blockIdx.y;
    const PatchRec r = A.patch_rec[s * kMaxPatches + p];
    if (r.status == 0) return;
    const int base = A.scan_off[s];
    const int off = A.patch_off[s * (kMaxPatches + 1) + p];
    const int* o = A.emit_off + ((size_t)s * kMaxPatches + p) * 4;
    const int xg = o[0], xng = o[1], xa = o[2], xr = o[3];
    const bool kept = (r.status == 1);
    const uint32_t* seg = A.seg + (size_t)base + off;
    // ground part of a kept patch -> cloud_out
    if (kept) {
        for (int e = threadIdx.x; e < r.n_g; e += kEmitThreads) {
            uint32_t id = seg[e] & 0x7fffffffu;
            A.ground_idx[(size_t)base + xg + e] = (int32_t)id;
            A.cls[base + id] = SCVOD_CLS_GROUND;
        }
    }
    // non-ground stream of this patch: elements [e0, n)
    const int e0 = kept ? r.n_g : 0;
    int run_keep = 0;
    for (int c0 = e0; c0 < r.n; c0 += kEmitThreads) {
        int e = c0 + threadIdx.x;
        uint32_t v = 0;
        int keep = 0;
        if (e < r.n) {
            v = seg[e];
            keep = (int)(v >> 31);
        }
        int tk;
        int ek = block_excl_scan<kEmitThreads>(keep, tk, wsum);
        if (e < r.n) {
            uint32_t id = v & 0x7fffffffu;
            int spos = e - e0;  // position in the non-ground stream of this patch
            A.nonground_idx[(size_t)base + xng + spos] = (int32_t)id;
            A.cls[base + id] = SCVOD_CLS_NONGROUND;
            if (keep) {
                float4 q = A.pts[base + id];
                Apri a;
                apri_of_point(P.bin, q.x, q.y, q.z, q.w, a);
                size_t dst = (size_t)base + xa + run_keep + ek;
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
                A.apri_src[dst] = (int32_t)id;
            } else {
                A.rejected_src[(size_t)base + xr + (spos - (run_keep + ek))] = (int32_t)id;
            }
        }
        run_keep += tk;
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
            } else {
                A.rejected_src[(size_t)base + (i - (run + ek))] = i;
            }
        }
        run += tk;
    }
    if (threadIdx.x == 0) {
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

// ------------------------------------------------------------------------------------------
// Voxel stage (SSC::makeHashCloud): bucket by the high bits of the key, sort (key, apri idx)
// inside each bucket, per-voxel sequential intensity mean / variance.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int vx_bucket_of(const DevParams& P, int32_t voxel_idx) {
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
        if (i < n) atomicAdd(&hist[vx_bucket_of(P, A.apri[(size_t)base + i].voxel_idx)], 1);
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
            key[it] = A.apri[(size_t)base + i].voxel_idx;
            bk[it] = vx_bucket_of(P, key[it]);
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
        if (bk[it] >= 0) A.vkeys[(size_t)base + hist[bk[it]] + rank[it]] = ((uint64_t)vx_bias(key[it]) << 32) | (uint32_t)i;
    }
}

template <int CAP, int THREADS, int MIN_N>
__global__ __launch_bounds__(THREADS) void k_vx_bucket(DevParams P, Arena A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* l_keys = (uint64_t*)smem;
    int* l_vbeg = (int*)(smem + (size_t)CAP * 8);
    int* wsum = (int*)(smem + (size_t)CAP * 12 + 16);
    const int b = blockIdx.x, s = blockIdx.y;
    const int m = A.vb_count[s * kMaxBuckets + b];
    if (MIN_N == 0 && m == 0) {
        if (threadIdx.x == 0) A.vb_nvox[s * kMaxBuckets + b] = 0;
        return;
    }
    if (m <= MIN_N) return;
    if (MIN_N == 0 && m > CAP) return;
    const int base = A.scan_off[s];
    const int off = A.vb_off[s * (kMaxBuckets + 1) + b];
    uint64_t* keys;
    int* vbeg;
    if (m <= CAP) {
        for (int j = threadIdx.x; j < m; j += THREADS) l_keys[j] = A.vkeys[(size_t)base + off + j];
        __syncthreads();
        keys = l_keys;
        vbeg = l_vbeg;
    } else {
        keys = A.vkeys + (size_t)base + off;
        vbeg = A.tmp_vox_begin + (size_t)base + off;  // rewritten below with final values
    }
    block_bitonic_sort<THREADS>(keys, m);
    // head flags + compaction of voxel starts
    int run = 0;
    for (int c0 = 0; c0 < m; c0 += THREADS) {
        int j = c0 + threadIdx.x;
        int head = 0;
        if (j < m) {
            // never form keys[-1]: with flat addressing that leaves the LDS aperture
            const uint64_t prev = keys[j > 0 ? j - 1 : 0];
            head = (j == 0) || ((uint32_t)(keys[j] >> 32) != (uint32_t)(prev >> 32));
        }
        int th;
        int eh = block_excl_scan<THREADS>(head, th, wsum);
        if (j < m) A.vox_pts[(size_t)base + off + j] = (int32_t)(uint32_t)keys[j];
        if (head) vbeg[run + eh] = j;
        run += th;
    }
    __syncthreads();
    const int nv = run;
    // per voxel: sequential fp32 mean, then population variance accumulated as float += double
    for (int v = threadIdx.x; v < nv; v += THREADS) {
        int j0 = vbeg[v];
        int j1 = (v + 1 < nv) ? vbeg[v + 1] : m;
        float av = 0.f;
        for (int j = j0; j < j1; ++j) av += A.apri[(size_t)base + (uint32_t)keys[j]].intensity;
        const float fn = (float)(j1 - j0);
        av = av / fn;
        float cov = 0.f;
        for (int j = j0; j < j1; ++j) {
            float in = A.apri[(size_t)base + (uint32_t)keys[j]].intensity;
            double d = (double)(in - av);
            cov = (float)((double)cov + d * d);
        }
        cov = cov / fn;
        int32_t vkey = vx_unbias((uint32_t)(keys[j0] >> 32));
        A.tmp_vox_key[(size_t)base + off + v] = vkey;
        A.tmp_vox_av[(size_t)base + off + v] = av;
        A.tmp_vox_cov[(size_t)base + off + v] = cov;
    }
    __syncthreads();
    // vbeg aliases tmp_vox_begin in the oversize path: every thread rewrites only its own entries
    for (int v = threadIdx.x; v < nv; v += THREADS) A.tmp_vox_begin[(size_t)base + off + v] = off + vbeg[v];
    if (threadIdx.x == 0) A.vb_nvox[s * kMaxBuckets + b] = nv;
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
        const scvod_apri& a = A.apri[(size_t)sb + J.members[k]];
        q = make_float4(a.x, a.y, a.z, a.intensity);
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

template <int CAP, int THREADS>
__global__ __launch_bounds__(THREADS) void k_track_unique(TrackJob J) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* l_keys = (uint32_t*)smem;
    int* wsum = (int*)(smem + (size_t)CAP * 4);
    const int c = blockIdx.x;
    const int k0 = J.pt_cluster_begin[c], k1 = J.pt_cluster_begin[c + 1];
    const int m = k1 - k0;
    uint32_t* keys;
    if (m <= CAP)
        keys = l_keys;
    else
        keys = (uint32_t*)(J.work + k0);
    // misses sort to the end as 0xffffffff
    for (int j = threadIdx.x; j < m; j += THREADS) keys[j] = (uint32_t)J.hit_slot[k0 + j];
    __syncthreads();
    block_bitonic_sort<THREADS>(keys, m);
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
        within[q] = (bi >= 0 && best <= r2) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------
// host-side launch sequences
// ------------------------------------------------------------------------------------------
#define TH_BEGIN(name) \
    if (th) th(tu, name, 1)
#define TH_END(name) \
    if (th) th(tu, name, 0)

constexpr int kPatchCapS = 1024, kPatchThreadsS = 64;
constexpr int kPatchCapL = 8192, kPatchThreadsL = 512;
constexpr size_t patch_lds_bytes(int cap) { return (size_t)cap * 17 + 128 + 256; }
constexpr int kVoxCapS = 1024, kVoxThreadsS = 64;
constexpr int kVoxCapL = 8192, kVoxThreadsL = 512;
constexpr size_t vox_lds_bytes(int cap) { return (size_t)cap * 12 + 16 + 128; }

void launch_process(const DevParams& P, const Arena& A, hipStream_t st, int do_patchwork, int apply_filter,
                    int do_voxels, TimerHook th, void* tu) {
    const int B = A.n_scans;
    if (B <= 0) return;
    if (do_patchwork) {
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
        dim3 gp(P.n_patches, B);
        hipFuncSetAttribute((const void*)k_pw_patch<kPatchCapL, kPatchThreadsL, kPatchCapS>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)patch_lds_bytes(kPatchCapL));
        TH_BEGIN("pw_patch_small");
        hipLaunchKernelGGL((k_pw_patch<kPatchCapS, kPatchThreadsS, 0>), gp, dim3(kPatchThreadsS),
                           patch_lds_bytes(kPatchCapS), st, P, A);
        TH_END("pw_patch_small");
        TH_BEGIN("pw_patch_large");
        hipLaunchKernelGGL((k_pw_patch<kPatchCapL, kPatchThreadsL, kPatchCapS>), gp, dim3(kPatchThreadsL),
                           patch_lds_bytes(kPatchCapL), st, P, A);
        TH_END("pw_patch_large");
        TH_BEGIN("emit_offsets");
        hipLaunchKernelGGL(k_emit_offsets, dim3(B), dim3(1024), 0, st, P, A);
        TH_END("emit_offsets");
        TH_BEGIN("emit");
        hipLaunchKernelGGL(k_emit, gp, dim3(kEmitThreads), 0, st, P, A);
        TH_END("emit");
    } else {
        TH_BEGIN("bin_direct");
        hipLaunchKernelGGL(k_bin_direct, dim3(B), dim3(1024), 0, st, P, A, apply_filter);
        TH_END("bin_direct");
    }
    if (do_voxels) {
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
        hipFuncSetAttribute((const void*)k_vx_bucket<kVoxCapL, kVoxThreadsL, kVoxCapS>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)vox_lds_bytes(kVoxCapL));
        TH_BEGIN("vx_bucket_small");
        hipLaunchKernelGGL((k_vx_bucket<kVoxCapS, kVoxThreadsS, 0>), gb, dim3(kVoxThreadsS), vox_lds_bytes(kVoxCapS),
                           st, P, A);
        TH_END("vx_bucket_small");
        TH_BEGIN("vx_bucket_large");
        hipLaunchKernelGGL((k_vx_bucket<kVoxCapL, kVoxThreadsL, kVoxCapS>), gb, dim3(kVoxThreadsL),
                           vox_lds_bytes(kVoxCapL), st, P, A);
        TH_END("vx_bucket_large");
        TH_BEGIN("vx_final_offsets");
        hipLaunchKernelGGL(k_vx_final_offsets, dim3(B), dim3(1024), 0, st, P, A);
        TH_END("vx_final_offsets");
        TH_BEGIN("vx_final");
        hipLaunchKernelGGL(k_vx_final, gb, dim3(256), 0, st, P, A);
        TH_END("vx_final");
    }
}

void launch_track(const DevParams& P, const Arena& A, const TrackJob& J, int batch_mode, hipStream_t st,
                  TimerHook th, void* tu) {
    if (J.n_pts > 0) {
        TH_BEGIN("track_probe");
        hipLaunchKernelGGL(k_track_probe, dim3((J.n_pts + 255) / 256), dim3(256), 0, st, P, A, J, batch_mode);
        TH_END("track_probe");
    }
    if (J.n_clusters > 0) {
        TH_BEGIN("track_unique");
        hipLaunchKernelGGL((k_track_unique<8192, 256>), dim3(J.n_clusters), dim3(256), 8192 * 4 + 64, st, J);
        TH_END("track_unique");
    }
}

void launch_nn(const float* map_xyz, int32_t n_map, const float* q_xyz, int32_t n_q, float radius, int32_t* nn_idx,
               float* nn_sq, uint8_t* within, hipStream_t st) {
    if (n_q <= 0) return;
    hipLaunchKernelGGL(k_nn_brute, dim3((n_q + kNnThreads - 1) / kNnThreads), dim3(kNnThreads), 0, st, map_xyz, n_map,
                       q_xyz, n_q, radius * radius, nn_idx, nn_sq, within);
}

}  // namespace scvod
