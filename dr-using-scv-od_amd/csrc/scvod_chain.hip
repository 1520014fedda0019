// scvod_chain.hip -- the SEQUENTIAL chain of SSC::tracking on the device (gfx950, wave64).
//
// Reference: SSC::segDF runs tracking(frame i, frame i + 1) in order (/root/reference/src/ssc.cpp:1449-1451) and every call
// mutates the successor before the next call walks it as `pre`:
//   * one label, ratio >= occupancy, successor cluster a car  -> the walked cluster's TRANSFORMED cloud is appended to the
//     successor cluster's cloud (ssc.cpp:1378-1384): the next call re-bins those points too (they are transformed again);
//   * one label, ratio < occupancy, successor cluster not a car -> the hit voxels are split off into a new cluster
//     (ssc.cpp:1351-1372): later clusters of the same call see the new label and the reduced |occupy_voxels|;
//   * several labels -> the car clusters hit at or above the ratio are fused into one new car cluster (ssc.cpp:1396-1419):
//     its cloud is cloud_use of the fused members (appended clouds are dropped), it is walked by the next call after the
//     original clusters (name = max_name++).
// scvod_track.hip takes every cluster against the successor's FRESH segmentation (first order, all pairs in parallel) and
// leaves, per car cluster, the sorted unique hit list, remap_name and the first-order state.  This file replays the chain
// on top of that: a walker (one workgroup) steps through the frames of a sequence in order, carries the appended clouds
// (a pool of points re-transformed at every step with the explicit fp32 products of utility.h:401-404), evaluates only
// what the first-order pass could not know -- clusters that carry appended points, fused clusters, and clusters whose hit
// labels an earlier cluster of the same call re-labelled -- and applies the re-labelling through stamped override words
// (no table is copied or cleared per step).  cluster_set is walked in ascending canonical name, created clusters after
// the original ones in creation order (the reference's order is that of ITS names in ITS unordered_map: DESIGN.md 2).
//
// The chain is sequential per sequence (552 steps for seq 05 with skip_ 5), so it is cut into SEGMENTS walked
// concurrently: the walker of steps [a, b) starts `warm` steps earlier from the fresh state of that frame; appended
// points live a dozen frames, so its state at step a normally equals the true one.  That is CHECKED, not assumed: a
// second kernel compares the state the previous segment really ended in with the warm-up's snapshot (entries, parts,
// every carried coordinate, bit for bit) and re-walks the segment from the true state when they differ.  The result is
// the sequential chain's, whatever the warm-up length.
#include "scvod_chain.h"

namespace scvod {

namespace {

constexpr int kChThreads = 1024;
constexpr int kChWaves = kChThreads / 64;
constexpr int kChSamples = 16384;  // sampled successor keys in LDS (64 KB)

typedef unsigned long long u64;

struct Wk {  // a walker's workspace
    int32_t* hdr;
    int4* ent[3];
    int32_t* parts[3];
    float4* pool[3];
    int32_t* chit;
    int2* evr;
    int32_t* suniq;
    int2* spairs;
    u64* vlab;
    u64* lcnt;
    u64* lfwd;
    int32_t* newent;
    int4* cmeta;
    int32_t* cparts;
    int4* links;
    int32_t* dsz;
    int32_t* eidx;
    int2* rp;
};

__device__ __forceinline__ Wk wk_of(const ChainWs& S, int w) {
    unsigned char* b = S.base + (size_t)w * S.stride;
    Wk k;
    k.hdr = (int32_t*)(b + S.off_hdr);
    for (int i = 0; i < 3; ++i) {
        k.ent[i] = (int4*)(b + S.off_ent[i]);
        k.parts[i] = (int32_t*)(b + S.off_parts[i]);
        k.pool[i] = (float4*)(b + S.off_pool[i]);
    }
    k.chit = (int32_t*)(b + S.off_chit);
    k.evr = (int2*)(b + S.off_evr);
    k.suniq = (int32_t*)(b + S.off_suniq);
    k.spairs = (int2*)(b + S.off_spairs);
    k.vlab = (u64*)(b + S.off_vlab);
    k.lcnt = (u64*)(b + S.off_lcnt);
    k.lfwd = (u64*)(b + S.off_lfwd);
    k.newent = (int32_t*)(b + S.off_newent);
    k.cmeta = (int4*)(b + S.off_cmeta);
    k.cparts = (int32_t*)(b + S.off_cparts);
    k.links = (int4*)(b + S.off_links);
    k.dsz = (int32_t*)(b + S.off_dsz);
    k.eidx = (int32_t*)(b + S.off_eidx);
    k.rp = (int2*)(b + S.off_rp);
    return k;
}

// header words of a walker
enum { H_EPOCH = 0, H_END_SLOT = 1, H_HAS_SNAP = 2, H_SNAP_OK = 3, H_NENT = 4 /* [3] */, H_SPEC = 7, H_NCARRIED = 8 /* [3] */, H_NPARTS = 12 /* [3] */ };

__device__ __forceinline__ int wmin_i(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ int wmax_i(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ int wsum_i(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// volatile views: the override words are written and read back by different lanes of one wave inside a step
__device__ __forceinline__ u64 ld64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int ld32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st32(int32_t* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct StepEnv {  // the successor as the current step sees it
    const int4* tab;        // fresh records {key, label, |occupy_voxels|, type}
    const int32_t* rep;     // fresh label ids
    int nv;
    uint32_t epoch;
    u64* vlab;
    u64* lcnt;
    u64* lfwd;
};
__device__ __forceinline__ bool stamped(u64 w, uint32_t epoch) { return (uint32_t)(w >> 32) == epoch; }

// label id a voxel of the successor carries NOW (split voxels are overridden one by one, fused labels are forwarded)
__device__ __forceinline__ int cur_label(const StepEnv& E, int u) {
    const u64 o = ld64(&E.vlab[u]);
    int id = stamped(o, E.epoch) ? (int)(uint32_t)o : E.rep[u];
    while (id >= 0) {
        const u64 f = ld64(&E.lfwd[id]);
        if (!stamped(f, E.epoch)) break;
        id = (int)(uint32_t)f;
    }
    return id;
}
// |occupy_voxels| (low 28 bits) and type (bits 28..29) of a label NOW
__device__ __forceinline__ uint32_t cur_cnttype(const StepEnv& E, int id) {
    const u64 c = ld64(&E.lcnt[id]);
    if (stamped(c, E.epoch)) return (uint32_t)c;
    const int4 r = E.tab[id];  // id < nv: a voxel of that label
    return (uint32_t)r.z | ((uint32_t)r.w << 28);
}
__device__ __forceinline__ bool label_dirty(const StepEnv& E, int id) {
    return stamped(ld64(&E.lcnt[id]), E.epoch) || stamped(ld64(&E.lfwd[id]), E.epoch);
}

__device__ __forceinline__ int lower_bound_i(const int32_t* a, int n, int key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

#ifdef SCVOD_PROFILE
#define CH_MARK(i)                                                   \
    do {                                                             \
        if (threadIdx.x == 0) {                                      \
            const long long now__ = wall_clock64();                  \
            atomicAdd(&C.stats[3 + (i)], (int)(now__ - ch_t0__));    \
            ch_t0__ = now__;                                         \
        }                                                            \
    } while (0)
#define CH_T0 long long ch_t0__ = wall_clock64()
#else
#define CH_MARK(i)
#define CH_T0
#endif

constexpr int kLdsEnt = 512;      // clusters of a step whose records live in LDS (fast path)
constexpr int kSmSamples = 8192;  // sampled successor keys of the fast path (32 KB)

struct ShSmall {
    int32_t skeys[kSmSamples];
    int4 fr[kLdsEnt];           // {remap_name.size(), unique hits, id of the first label, its hit count} against the FRESH successor
    int4 fx[kLdsEnt];           // {|occupy_voxels| + type of that label (fresh), root of a one-cluster entry / -1, cloud points, own points}
    int32_t fuo[kLdsEnt];       // one-cluster entry: tk_mbegin of its root; otherwise offset of the entry's region in suniq / spairs
    int32_t nown[2 * kLdsEnt];  // successor: points of car cluster e
    int32_t ncrep[2 * kLdsEnt]; // successor: label id of car cluster e
    int32_t dsz[2 * kLdsEnt];   // points appended to successor cluster e (then its cursor)
    int32_t eidx[2 * kLdsEnt];  // successor cluster e -> entry of the next state, -1 gone
    int32_t ncb[2 * kLdsEnt];   // entry of the next state -> first carried point
    int4 lk[kLdsEnt];           // appended clouds: {walked entry, successor cluster, points, offset in the next pool / -1}
    int32_t lpre[kLdsEnt + 1];  // their exclusive point offsets (copy phase)
};
struct Shared {
    union {
        int32_t skeys[kChSamples];
        ShSmall sm;
    };
    int32_t wsum[kChWaves + 1];
    int32_t bc[8];  // broadcast words
};

// copies state `src` of walker workspace A to state `dst` of workspace B (all threads)
__device__ __forceinline__ void copy_state(const Wk& A, int src, const Wk& B, int dst) {
    const int ne = A.hdr[H_NENT + src], nc = A.hdr[H_NCARRIED + src], np = A.hdr[H_NPARTS + src];
    for (int i = threadIdx.x; i < 2 * ne; i += kChThreads) B.ent[dst][i] = A.ent[src][i];
    for (int i = threadIdx.x; i < np; i += kChThreads) B.parts[dst][i] = A.parts[src][i];
    for (int i = threadIdx.x; i < nc; i += kChThreads) B.pool[dst][i] = A.pool[src][i];
    __syncthreads();
    if (threadIdx.x == 0) {
        B.hdr[H_NENT + dst] = ne;
        B.hdr[H_NCARRIED + dst] = nc;
        B.hdr[H_NPARTS + dst] = np;
    }
    __syncthreads();
}

// bit-for-bit comparison of two states (all threads); returns true when they are equal
// The carried points of two states whose entries are equal: the same points PER ENTRY, in any order.  An entry's carried cloud is
// the concatenation of what its predecessors appended, block after block in the order of the links that fed it -- which depends
// on where a walk started (a fresh state lists the clusters in segmentation order, a state with a history has the created ones
// behind them), while nothing that reads the cloud depends on its order: every point is transformed on its own and marks the
// voxel it falls into (a bit set), and the cloud is copied on as a whole.  Compared as two 64-bit sums of mixed coordinate bits
// per entry (a multiset signature; the bit-for-bit comparison of rounds 3-4a made every shard boundary look different although
// the walk behind it was the same, and cost in-GPU segments a second walk now and then).
__device__ __forceinline__ unsigned long long pool_mix(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}
constexpr int kPdEnt = 512;   // entries whose offsets pools_differ keeps in LDS
__device__ __forceinline__ int pools_differ(const int4* ent, int ne, int nc, const float4* pa, const float4* pb) {
    int diff = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (nc <= 0) return 0;
    if (ne <= kPdEnt) {
        // flat over the points: the entry a point belongs to (offsets ascend with the entry index) salts its hash, so ONE pair of sums
        // per state stands for all the per-entry multisets
        __shared__ int32_t pd_off[kPdEnt + 1];
        __shared__ unsigned long long pd_sum[kChThreads / 64][4];
        __syncthreads();  // (the arrays may still be read by a previous call)
        for (int e = threadIdx.x; e < ne; e += kChThreads) pd_off[e] = ent[2 * e].z;
        if (threadIdx.x == 0) pd_off[ne] = nc;
        __syncthreads();
        unsigned long long a1 = 0, a2 = 0, b1 = 0, b2 = 0;
        for (int i = threadIdx.x; i < nc; i += kChThreads) {
            const float4 x = pa[i], y = pb[i];
            int lo = 0, hi = ne;  // the last entry whose offset is <= i
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (pd_off[mid] <= i)
                    lo = mid;
                else
                    hi = mid;
            }
            const unsigned long long salt = pool_mix((unsigned long long)lo + 0x51ull);
            const unsigned long long kx = (unsigned long long)__float_as_uint(x.x) | ((unsigned long long)__float_as_uint(x.y) << 32);
            const unsigned long long ky = (unsigned long long)__float_as_uint(y.x) | ((unsigned long long)__float_as_uint(y.y) << 32);
            const unsigned long long zx = __float_as_uint(x.z), zy = __float_as_uint(y.z);
            a1 += pool_mix(kx ^ (zx * 0x9e3779b97f4a7c15ull) ^ salt);
            a2 += pool_mix((kx + salt) * 0xc2b2ae3d27d4eb4full + zx + 1ull);
            b1 += pool_mix(ky ^ (zy * 0x9e3779b97f4a7c15ull) ^ salt);
            b2 += pool_mix((ky + salt) * 0xc2b2ae3d27d4eb4full + zy + 1ull);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            a1 += __shfl_xor(a1, d);
            a2 += __shfl_xor(a2, d);
            b1 += __shfl_xor(b1, d);
            b2 += __shfl_xor(b2, d);
        }
        if (lane == 0) {
            pd_sum[wave][0] = a1;
            pd_sum[wave][1] = a2;
            pd_sum[wave][2] = b1;
            pd_sum[wave][3] = b2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t[4] = {0, 0, 0, 0};
            for (int w = 0; w < kChThreads / 64; ++w)
                for (int k = 0; k < 4; ++k) t[k] += pd_sum[w][k];
            diff = (t[0] != t[2]) | (t[1] != t[3]);
        }
        return diff;
    }
    for (int e = wave; e < ne; e += kChThreads / 64) {
        const int4 r = ent[2 * e];
        const int off = r.z, cnt = r.w;
        if (cnt <= 0 || off < 0 || off + cnt > nc) continue;
        unsigned long long a1 = 0, a2 = 0, b1 = 0, b2 = 0;
        for (int i = lane; i < cnt; i += 64) {
            const float4 x = pa[off + i], y = pb[off + i];
            const unsigned long long kx = (unsigned long long)__float_as_uint(x.x) | ((unsigned long long)__float_as_uint(x.y) << 32);
            const unsigned long long ky = (unsigned long long)__float_as_uint(y.x) | ((unsigned long long)__float_as_uint(y.y) << 32);
            const unsigned long long zx = __float_as_uint(x.z), zy = __float_as_uint(y.z);
            a1 += pool_mix(kx ^ (zx * 0x9e3779b97f4a7c15ull));
            a2 += pool_mix(kx * 0xc2b2ae3d27d4eb4full + zx + 1ull);
            b1 += pool_mix(ky ^ (zy * 0x9e3779b97f4a7c15ull));
            b2 += pool_mix(ky * 0xc2b2ae3d27d4eb4full + zy + 1ull);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            a1 += __shfl_xor(a1, d);
            a2 += __shfl_xor(a2, d);
            b1 += __shfl_xor(b1, d);
            b2 += __shfl_xor(b2, d);
        }
        diff |= (a1 != b1) | (a2 != b2);
    }
    return diff;
}
__device__ __forceinline__ bool same_state(const Wk& A, int sa, const Wk& B, int sb) {
    const int ne = A.hdr[H_NENT + sa], nc = A.hdr[H_NCARRIED + sa], np = A.hdr[H_NPARTS + sa];
    int diff = (ne != B.hdr[H_NENT + sb]) || (nc != B.hdr[H_NCARRIED + sb]) || (np != B.hdr[H_NPARTS + sb]);
    if (!diff) {
        for (int i = threadIdx.x; i < 2 * ne; i += kChThreads) {
            const int4 x = A.ent[sa][i], y = B.ent[sb][i];
            diff |= (int)(x.x != y.x) | (int)(x.y != y.y) | (int)(x.z != y.z) | (int)(x.w != y.w);
        }
        for (int i = threadIdx.x; i < np; i += kChThreads) diff |= A.parts[sa][i] != B.parts[sb][i];
        diff |= pools_differ(A.ent[sa], ne, nc, A.pool[sa], B.pool[sb]);
    }
    return __syncthreads_or(diff) == 0;
}

// a state as an external record (scvod_chain.h): header {ne, nc, np, valid}, entries, parts (padded to 16 bytes), carried points
struct ExtRec {
    const int32_t* hdr;
    const int4* ent;
    const int32_t* parts;
    const float4* pool;
};
__device__ __forceinline__ ExtRec ext_of(const unsigned char* b) {
    ExtRec r;
    r.hdr = (const int32_t*)b;
    const int ne = r.hdr[0], np = r.hdr[2];
    r.ent = (const int4*)(b + 16);
    r.parts = (const int32_t*)(b + 16 + (size_t)32 * ne);
    r.pool = (const float4*)(b + 16 + (size_t)32 * ne + (((size_t)4 * np + 15) & ~(size_t)15));
    return r;
}
__device__ __forceinline__ bool same_state_ext(const Wk& A, int sa, const ExtRec& B) {
    const int ne = A.hdr[H_NENT + sa], nc = A.hdr[H_NCARRIED + sa], np = A.hdr[H_NPARTS + sa];
    int diff = (ne != B.hdr[0]) || (nc != B.hdr[1]) || (np != B.hdr[2]);
    if (!diff) {
        for (int i = threadIdx.x; i < 2 * ne; i += kChThreads) {
            const int4 x = A.ent[sa][i], y = B.ent[i];
            diff |= (int)(x.x != y.x) | (int)(x.y != y.y) | (int)(x.z != y.z) | (int)(x.w != y.w);
        }
        for (int i = threadIdx.x; i < np; i += kChThreads) diff |= A.parts[sa][i] != B.parts[i];
        diff |= pools_differ(A.ent[sa], ne, nc, A.pool[sa], B.pool);
    }
    return __syncthreads_or(diff) == 0;
}
__device__ __forceinline__ void copy_from_ext(const ExtRec& B, const Wk& K, int dst, int cap_ent, int cap_pool, int32_t* stats) {
    int ne = B.hdr[0], nc = B.hdr[1], np = B.hdr[2];
    if (ne > cap_ent || np > cap_ent || nc > cap_pool) {  // (a state of another shard that outgrows this one's workspace: flagged)
        if (threadIdx.x == 0) atomicOr(&stats[0], nc > cap_pool ? 1 : 2);
        ne = min(ne, cap_ent), np = min(np, cap_ent), nc = 0;
    }
    for (int i = threadIdx.x; i < 2 * ne; i += kChThreads) K.ent[dst][i] = B.ent[i];
    for (int i = threadIdx.x; i < np; i += kChThreads) K.parts[dst][i] = B.parts[i];
    for (int i = threadIdx.x; i < nc; i += kChThreads) K.pool[dst][i] = B.pool[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        K.hdr[H_NENT + dst] = ne;
        K.hdr[H_NCARRIED + dst] = nc;
        K.hdr[H_NPARTS + dst] = np;
    }
    __syncthreads();
}

// the freshly segmented car clusters of scan s as a state: one entry per cluster, nothing carried
__device__ __forceinline__ void fresh_state(const Arena& A, const Wk& K, int slot, int s, int cap_ent, int32_t* stats) {
    const int base = A.scan_off[s];
    int ncar = A.tk_scan[s * 4 + 0];
    if (ncar > cap_ent) {
        if (threadIdx.x == 0) atomicOr(&stats[0], 2);
        ncar = cap_ent;
    }
    __shared__ int32_t fs_wsum[kChWaves + 1];
    int run = 0;
    for (int o0 = 0; o0 < ncar; o0 += kChThreads) {
        const int o = o0 + threadIdx.x;
        int own = 0;
        if (o < ncar) own = A.cl_count[(size_t)base + A.tk_clusters[(size_t)base + o]];
        int total;
        const int ex = block_excl_scan<kChThreads>(own, total, fs_wsum);
        if (o < ncar) {
            K.ent[slot][2 * o] = make_int4(o, 1, 0, 0);
            K.ent[slot][2 * o + 1] = make_int4(own, run + ex, 0, 0);
            K.parts[slot][o] = o;
        }
        run += total;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        K.hdr[H_NENT + slot] = ncar;
        K.hdr[H_NCARRIED + slot] = 0;
        K.hdr[H_NPARTS + slot] = ncar;
    }
    __syncthreads();
}


// transform (utility.h:401-404) + unfiltered re-bin (ssc.cpp:1280-1286) + look-up of the carried points; their new
// coordinates replace the old ones.  skeys: `ns` keys of the table sampled every 2^shift records.
__device__ __forceinline__ void carried_probe(const DevParams& P, const Wk& K, float4* pool, int ncarried, const float* T, const int4* tab, int nv,
                                              const int32_t* skeys, int ns, int shift) {
#ifndef CH_U
#define CH_U 1
#endif
    constexpr int U = CH_U;  // points per thread and round (measured: 1 beats 2 and 4 -- the phase is ALU-bound, more in flight only costs registers)
    for (int c0 = threadIdx.x; c0 < ncarried; c0 += kChThreads * U) {
        float4 q[U];
        int key[U], lo[U], hi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) q[u] = pool[min(c0 + u * kChThreads, ncarried - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float x = T[0] * q[u].x + T[1] * q[u].y + T[2] * q[u].z + T[3];
            const float y = T[4] * q[u].x + T[5] * q[u].y + T[6] * q[u].z + T[7];
            const float z = T[8] * q[u].x + T[9] * q[u].y + T[10] * q[u].z + T[11];
            q[u] = make_float4(x, y, z, q[u].w);
            int32_t vi;
            if (!voxel_idx_fast(P.bin, P.binfast, x, y, z, &vi)) {  // (a few points per thousand: next to a bin edge)
                Apri a;
                apri_of_point(P.bin, x, y, z, q[u].w, a);
                vi = a.voxel_idx;
            }
            key[u] = vi;
            lo[u] = 0;  // first sample > key
            hi[u] = ns;
        }
        bool more = ns > 0;
        while (more) {
            more = false;
            int sk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) sk[u] = skeys[min((lo[u] + hi[u]) >> 1, ns - 1)];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (lo[u] < hi[u]) {
                    const int mid = (lo[u] + hi[u]) >> 1;
                    if (sk[u] <= key[u])
                        lo[u] = mid + 1;
                    else
                        hi[u] = mid;
                    more |= lo[u] < hi[u];
                }
            }
        }
        int4 first[U];
#pragma unroll
        for (int u = 0; u < U; ++u) first[u] = tab[min(max(lo[u] - 1, 0) << shift, max(nv - 1, 0))];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u * kChThreads;
            if (c >= ncarried) continue;
            int slot = -1;
            if (lo[u] > 0) {
                const int a0 = (lo[u] - 1) << shift;
                slot = tk_find_slot(tab, a0, min(a0 + (1 << shift), nv), first[u], key[u]);
            }
            pool[c] = q[u];
            K.chit[c] = slot;
        }
    }
}

// One wave: sorted unique hit list (sampleVec, ssc.cpp:1319-1321) of an entry whose cloud is more than one original cluster's
// own points -- the unique lists of its parts (scvod_track.hip) and the hits of its carried points -- through a bitset
// over the successor's table, then remap_name against the FRESH labels.  Returns {labels, unique hits}; lists at uq / pr.
__device__ __forceinline__ int2 eval_entry(const Arena& A, const Wk& K, const StepEnv& E, int base_i, const int32_t* parts, const int4 e0, uint32_t* bits,
                           int32_t* uq, int2* pr) {
    const int lane = threadIdx.x & 63;
    int wlo = 0x7fffffff, whi = -1;
    for (int p = 0; p < e0.y; ++p) {
        const int root = A.tk_clusters[(size_t)base_i + parts[e0.x + p]];
        const int mb = A.tk_mbegin[(size_t)base_i + root];
        const int nu = A.tk_nuniq[(size_t)base_i + root];
        const int32_t* src = A.tk_uniq + (size_t)base_i + mb;
        for (int j0 = lane; j0 < nu; j0 += 256) {
            int sl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) sl[u] = (j0 + 64 * u < nu) ? src[j0 + 64 * u] : -1;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (sl[u] >= 0) {
                    atomicOr(&bits[sl[u] >> 5], 1u << (sl[u] & 31));
                    wlo = min(wlo, sl[u] >> 5);
                    whi = max(whi, sl[u] >> 5);
                }
        }
    }
    {
        const int32_t* src = K.chit + e0.z;
        for (int j0 = lane; j0 < e0.w; j0 += 256) {
            int sl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) sl[u] = (j0 + 64 * u < e0.w) ? src[j0 + 64 * u] : -1;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (sl[u] >= 0) {
                    atomicOr(&bits[sl[u] >> 5], 1u << (sl[u] & 31));
                    wlo = min(wlo, sl[u] >> 5);
                    whi = max(whi, sl[u] >> 5);
                }
        }
    }
    wlo = wmin_i(wlo);
    whi = wmax_i(whi);
    wave_sync();
    // sorted unique slots over the words touched (cleared on the way)
    int U = 0;
    for (int w0 = wlo; w0 <= whi; w0 += 64) {
        const int w = w0 + lane;
        uint32_t word = 0u;
        if (w <= whi) {
            word = bits[w];
            bits[w] = 0u;
        }
        const int c = __popc(word);
        const int inc = wave_incl_scan(c);
        int o = U + inc - c;
        while (word) {
            const int b = __ffs(word) - 1;
            word &= word - 1;
            uq[o++] = (w << 5) + b;
        }
        U += __shfl(inc, 63);
    }
    wave_sync();
    // remap_name: the labels of the list, 64 slots at a time; lane t holds the t-th distinct label id met so far and its count (a
    // cluster hits a handful of labels; more than 64 fall back to the min-iteration below)
    int tid_ = -1, tcnt = 0, ntab = 0;
    bool table_ok = true;
    for (int j0 = 0; j0 < U; j0 += 64) {
        const int j = j0 + lane;
        const int id = (j < U) ? E.rep[uq[j]] : -1;
        bool todo = id >= 0;
        while (__any(todo)) {
            const int firstl = __ffsll((long long)__ballot(todo)) - 1;
            const int k0 = __shfl(id, firstl);
            const bool mine = todo && id == k0;
            const int cntk = __popcll(__ballot(mine));
            if (mine) todo = false;
            const unsigned long long hit = __ballot(tid_ == k0);
            if (hit) {
                if (tid_ == k0) tcnt += cntk;
            } else if (ntab < 64) {
                if (lane == ntab) {
                    tid_ = k0;
                    tcnt = cntk;
                }
                ++ntab;
            } else {
                table_ok = false;
            }
        }
    }
    int npairs = 0;
    if (table_ok) {  // remap_name in ascending label id
        int rank = 0;
        for (int t = 0; t < ntab; ++t) rank += (__shfl(tid_, t) < tid_) ? 1 : 0;
        if (lane < ntab) pr[rank] = make_int2(tid_, tcnt);
        npairs = ntab;
    } else {
        int curid = -1;
        for (;;) {
            int mn = 0x7fffffff;
            for (int j = lane; j < U; j += 64) {
                const int id = E.rep[uq[j]];
                if (id > curid) mn = min(mn, id);
            }
            mn = wmin_i(mn);
            if (mn == 0x7fffffff) break;
            int cnt = 0;
            for (int j = lane; j < U; j += 64) cnt += E.rep[uq[j]] == mn ? 1 : 0;
            cnt = wsum_i(cnt);
            if (lane == 0) pr[npairs] = make_int2(mn, cnt);
            ++npairs;
            curid = mn;
        }
    }
    wave_sync();
    return make_int2(npairs, U);
}

struct EntEval {        // what the walk needs of one cluster: remap_name against the fresh successor
    int np, nu;
    const int32_t* uq;  // sorted unique hit slots
    const int2* pr;     // remap_name: {label name, count} with the ids in prid (one original cluster), else {id, count}
    const int32_t* prid;
    int L0, c0;         // first label's id and count (np >= 1)
    uint32_t ct0;       // its fresh |occupy_voxels| + type, valid when have0
    bool have0;
    int size;           // points of the cluster's cloud (own + carried)
};
struct StepCounters {
    int n_dirty, nl, n_newlab, n_created, n_cparts;
    int kid;  // label id of the successor's cluster that still carries Frame::max_name (ssc.cpp:354), -1: none, or handed out already
};
// `frame_ssc.max_name = cluster_name ++` (ssc.cpp:354) keeps the LAST USED running number K, so the first
// `cluster_new.name = frame_next_.max_name ++` of a call (ssc.cpp:1357, :1401) is K again.  The label id of the cluster that
// still carries K in the successor as refineClusterByBoundingBox left it (scvod_lastname.hip found which one it is), -1 when
// no cluster does: then the number is as good as a fresh one.
__device__ __forceinline__ int max_name_label(const Arena& A, const ChainJob& C, const StepEnv& E, int sj) {
    if (!C.literal_max_name) return -1;
    const int name = A.cc_last[(size_t)sj * 4], u = A.cc_last[(size_t)sj * 4 + 1];
    if (name < 0 || u < 0 || u >= E.nv) return -1;
    return E.tab[u].y == name ? E.rep[u] : -1;
}

// SSC::tracking's decision and re-labelling for ONE walked cluster against the successor as it is now (ssc.cpp:1323-1421);
// one wave, every lane runs it with the same arguments.  Returns Cluster::state.  links / dsz_now: where an appended
// cloud is recorded (LDS on the fast path).
__device__ __forceinline__ int commit_entry(const StepEnv& E, const TrackBatch& J, const ChainJob& C, const Wk& K, EntEval v, int k, const int32_t* car_j,
                            int ncar_j, StepCounters& S, int4* links, int32_t* dsz_now) {
    const int lane = threadIdx.x & 63;
    const int nv = E.nv;
    bool stale = false;
    if (S.n_dirty) {
        if (v.np == 1) {
            stale = label_dirty(E, v.L0);
        } else {
            for (int p = lane; p < v.np; p += 64) stale |= label_dirty(E, v.prid ? v.prid[p] : v.pr[p].x);
            stale = __any(stale);
        }
    }
    if (stale) {  // an earlier cluster of this call re-labelled something this one hits: remap_name over the current labels
        int np2 = 0, curid = -1;
        for (;;) {
            int mn = 0x7fffffff;
            for (int j = lane; j < v.nu; j += 64) {
                const int id = cur_label(E, v.uq[j]);
                if (id > curid) mn = min(mn, id);
            }
            mn = wmin_i(mn);
            if (mn == 0x7fffffff) break;
            int cnt = 0;
            for (int j = lane; j < v.nu; j += 64) cnt += cur_label(E, v.uq[j]) == mn ? 1 : 0;
            cnt = wsum_i(cnt);
            if (lane == 0) K.rp[np2] = make_int2(mn, cnt);
            if (np2 == 0) {
                v.L0 = mn;
                v.c0 = cnt;
            }
            ++np2;
            curid = mn;
        }
        wave_sync();
        v.np = np2;
        v.pr = K.rp;
        v.prid = nullptr;
        v.have0 = false;
    }
    int state = -1;
    if (v.np == 0) {
        state = 1;  // ssc.cpp:1323-1326
    } else if (v.np == 1) {
        const int L = v.L0, c = v.c0;
        const uint32_t ct = (v.have0 && !S.n_dirty) ? v.ct0 : cur_cnttype(E, L);
        const int nvx = (int)(ct & 0x0fffffffu), typ = (int)(ct >> 28);
        const float ratio = (float)c / (float)nvx;  // ssc.cpp:1336
        if (ratio < J.occupancy) {
            if (typ == 2) {
                state = 1;  // ssc.cpp:1337-1349
            } else {       // ssc.cpp:1351-1372: the hit voxels leave the label for a new cluster of the same type
                state = 0;
                const int Y = nv + S.n_newlab;
                const int Kd = S.kid;
                S.kid = -1;
                if (Kd >= 0) {
                    // the split-off cluster is called K and K is alive: the hit voxels take K's label (ssc.cpp:1366), the source
                    // loses them (reduceVec, :1364), `cluster_set.insert` (:1372) is a no-op -- cluster K keeps its own
                    // occupy_voxels / type and now answers for these voxels too
                    if (Kd != L)
                        for (int j = lane; j < v.nu; j += 64)
                            if (cur_label(E, v.uq[j]) == L) st64(&E.vlab[v.uq[j]], ((u64)E.epoch << 32) | (uint32_t)Kd);
                    if (lane == 0) st64(&E.lcnt[L], ((u64)E.epoch << 32) | (uint32_t)(nvx - c) | ((uint32_t)typ << 28));
                    ++S.n_dirty;
                } else if (S.n_newlab < C.ws.cap_ent) {
                    for (int j = lane; j < v.nu; j += 64)
                        if (cur_label(E, v.uq[j]) == L) st64(&E.vlab[v.uq[j]], ((u64)E.epoch << 32) | (uint32_t)Y);
                    if (lane == 0) {
                        st64(&E.lcnt[L], ((u64)E.epoch << 32) | (uint32_t)(nvx - c) | ((uint32_t)typ << 28));
                        st64(&E.lcnt[Y], ((u64)E.epoch << 32) | (uint32_t)c | ((uint32_t)typ << 28));
                        st32(&K.newent[S.n_newlab], -1);
                    }
                    ++S.n_newlab;
                    ++S.n_dirty;
                } else if (lane == 0) {
                    atomicOr(&C.stats[0], 4);
                }
                wave_sync();
            }
        } else if (typ == 2) {  // ssc.cpp:1377-1384: static, its transformed cloud joins the successor cluster's cloud
            state = 0;
            int dst;
            if (L < nv)
                dst = lower_bound_i(car_j, ncar_j, E.tab[L].y);
            else
                dst = ncar_j + ld32(&K.newent[L - nv]);
            if (S.nl < C.ws.cap_ent) {
                if (lane == 0) {
                    links[S.nl] = make_int4(k, dst, v.size, 0);
                    if (dsz_now) dsz_now[dst] += v.size;
                }
                ++S.nl;
            } else if (lane == 0) {
                atomicOr(&C.stats[0], 2);
            }
        }
    } else {  // ssc.cpp:1396-1419: the car clusters hit at or above the ratio fuse into one new car cluster
        state = 0;
        const int Kd = S.kid;
        S.kid = -1;
        bool lost = false;
        if (Kd >= 0) {
            // the fused cluster is called K.  If K itself is one of the clusters fused, `erase` (ssc.cpp:1411) frees the name
            // before the insert and nothing differs from a fresh number; otherwise the voxels of the fused clusters take K's
            // label (:1417), the clusters leave cluster_set (:1411) and the insert (:1419) is a no-op: the fused cluster is
            // lost, nothing walks its points in the next call
            bool k_fused = false;
            for (int p = lane; p < v.np; p += 64) {
                const int L = v.prid ? v.prid[p] : v.pr[p].x;
                if (L != Kd) continue;
                const uint32_t ct = cur_cnttype(E, L);
                k_fused |= (int)(ct >> 28) == 2 && (float)v.pr[p].y / (float)(int)(ct & 0x0fffffffu) >= J.occupancy;
            }
            lost = !__any(k_fused);
        }
        if (lost) {
            bool any = false;
            for (int p = 0; p < v.np; ++p) {
                const int L = v.prid ? v.prid[p] : v.pr[p].x;
                const int c = v.pr[p].y;
                const uint32_t ct = cur_cnttype(E, L);
                const int nvx = (int)(ct & 0x0fffffffu), typ = (int)(ct >> 28);
                if (typ != 2 || !((float)c / (float)nvx >= J.occupancy)) continue;
                if (lane == 0) st64(&E.lfwd[L], ((u64)E.epoch << 32) | (uint32_t)Kd);
                any = true;
            }
            if (any) ++S.n_dirty;
            wave_sync();
        } else if (S.n_newlab < C.ws.cap_ent && S.n_created < C.ws.cap_ent) {
            const int N = nv + S.n_newlab;
            const int q = S.n_created;
            const int cbeg = S.n_cparts;
            int cntN = 0;
            for (int p = 0; p < v.np; ++p) {
                const int L = v.prid ? v.prid[p] : v.pr[p].x;
                const int c = v.pr[p].y;
                const uint32_t ct = cur_cnttype(E, L);
                const int nvx = (int)(ct & 0x0fffffffu), typ = (int)(ct >> 28);
                if (typ != 2 || !((float)c / (float)nvx >= J.occupancy)) continue;
                if (L < nv) {
                    const int o = lower_bound_i(car_j, ncar_j, E.tab[L].y);
                    if (S.n_cparts < C.ws.cap_ent) {
                        if (lane == 0) st32(&K.cparts[S.n_cparts], o);
                        ++S.n_cparts;
                    }
                } else {
                    const int4 m = K.cmeta[ld32(&K.newent[L - nv])];
                    for (int z = 0; z < m.y; ++z) {
                        if (S.n_cparts < C.ws.cap_ent) {
                            if (lane == 0) st32(&K.cparts[S.n_cparts], ld32(&K.cparts[m.x + z]));
                            ++S.n_cparts;
                        }
                    }
                }
                if (lane == 0) st64(&E.lfwd[L], ((u64)E.epoch << 32) | (uint32_t)N);
                cntN += nvx;
                wave_sync();
            }
            if (lane == 0) {
                st64(&E.lcnt[N], ((u64)E.epoch << 32) | (uint32_t)(cntN & 0x0fffffff) | (2u << 28));
                st32(&K.newent[S.n_newlab], q);
                K.cmeta[q] = make_int4(cbeg, S.n_cparts - cbeg, N, 0);
            }
            ++S.n_newlab;
            ++S.n_created;
            ++S.n_dirty;
            wave_sync();
        } else if (lane == 0) {
            atomicOr(&C.stats[0], 4);
        }
    }
    return state;
}

// One step of the chain: SSC::tracking(frame si, frame sj) on the state in slot `cur`; leaves the successor's state (as the
// next `pre`) in slot cur ^ 1.  write_out: states / dynamic counters of frame si are the chain's result (not warm-up).
// GENERIC path: any number of clusters, everything through the workspace in HBM.
__device__ __forceinline__ void chain_step_big(const DevParams& P, const Arena& A, const TrackBatch& J, const ChainJob& C, const Wk& K, Shared& sh,
                               uint32_t* bits_all, int cur, int si, int sj, bool write_out, int from_apri) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base_i = A.scan_off[si], base_j = A.scan_off[sj];
    const int nv = A.counts[sj * 8 + 6];
    const int ncar_j = min(A.tk_scan[sj * 4 + 0], C.ws.cap_ent);
    if (write_out) {  // a car cluster the walk does not reach (its frame lost it: commit_entry, Frame::max_name) keeps Cluster::state -1
        const int ncar_i = A.tk_scan[si * 4 + 0];
        for (int o = tid; o < ncar_i; o += kChThreads) A.cl_state[(size_t)base_i + A.tk_clusters[(size_t)base_i + o]] = -1;
    }
    const int4* tab = A.vox_track + base_j;
    const int nent = K.hdr[H_NENT + cur];
    const int ncarried = K.hdr[H_NCARRIED + cur];
    const int4* ent = K.ent[cur];
    const int32_t* parts = K.parts[cur];
    float4* pool = K.pool[cur];
    const int nxt = cur ^ 1;
    StepEnv E;
    E.tab = tab;
    E.rep = A.vox_rep + base_j;
    E.nv = nv;
    E.vlab = K.vlab;
    E.lcnt = K.lcnt;
    E.lfwd = K.lfwd;
    if (tid == 0) K.hdr[H_EPOCH] = K.hdr[H_EPOCH] + 1;
    __syncthreads();
    E.epoch = (uint32_t)K.hdr[H_EPOCH];
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = J.T[12 * si + i];

    // ---- phase 1: the carried points move into the successor's frame and are looked up there ----
    if (ncarried > 0) {
        int shift = 0;
        while (((nv + (1 << shift) - 1) >> shift) > kChSamples) ++shift;
        const int ns = (nv + (1 << shift) - 1) >> shift;
        for (int j = tid; j < ns; j += kChThreads) sh.skeys[j] = tab[(size_t)j << shift].x;
        __syncthreads();
        carried_probe(P, K, pool, ncarried, T, tab, nv, sh.skeys, ns, shift);
    }
    __syncthreads();

    // ---- phase 2: entries the first-order pass could not decide (appended points, fused clusters) ----
    {
        const int words = C.words;
        const int nw = (nv + 31) >> 5;
        if (nw > words) {  // table larger than the bitsets were sized for: flagged, the entries count as hitting nothing
            if (tid == 0) atomicOr(&C.stats[0], 8);
            for (int k = tid; k < nent; k += kChThreads) K.evr[k] = make_int2(0, 0);
        } else if (wave < C.n_eval_waves) {
            uint32_t* bits = bits_all + (size_t)wave * words;
            int cx = 0;
            for (int k = 0; k < nent; ++k) {
                const int4 e0 = ent[2 * k];
                if (e0.y == 1 && e0.w == 0) continue;  // one original cluster, nothing appended: scvod_track.hip's result stands
                if ((cx++ % C.n_eval_waves) != wave) continue;
                const int4 e1 = ent[2 * k + 1];
                const int2 r = eval_entry(A, K, E, base_i, parts, e0, bits, K.suniq + e1.y + e0.z, K.spairs + e1.y + e0.z);
                if (lane == 0) K.evr[k] = r;
            }
        }
    }
    __syncthreads();

    // ---- phase 3: the clusters in walking order, one wave: decision against the successor AS IT IS NOW, re-labelling ----
    if (wave == 0) {
        StepCounters S = {0, 0, 0, 0, 0, max_name_label(A, C, E, sj)};
        int ndc = 0, ndp = 0;
        const int32_t* car_j = A.tk_clusters + base_j;
        for (int k = 0; k < nent; ++k) {
            const int4 e0 = ent[2 * k];
            const int4 e1 = ent[2 * k + 1];
            EntEval v;
            if (e0.y == 1 && e0.w == 0) {
                const int root = A.tk_clusters[(size_t)base_i + parts[e0.x]];
                const int mb = A.tk_mbegin[(size_t)base_i + root];
                v.np = A.tk_npairs[(size_t)base_i + root];
                v.nu = A.tk_nuniq[(size_t)base_i + root];
                v.uq = A.tk_uniq + (size_t)base_i + mb;
                v.pr = A.tk_pairs + (size_t)base_i + mb;
                v.prid = A.tk_prep + (size_t)base_i + mb;
            } else {
                const int2 r = K.evr[k];
                v.np = r.x;
                v.nu = r.y;
                v.uq = K.suniq + e1.y + e0.z;
                v.pr = K.spairs + e1.y + e0.z;
                v.prid = nullptr;
            }
            v.L0 = v.np >= 1 ? (v.prid ? v.prid[0] : v.pr[0].x) : -1;
            v.c0 = v.np >= 1 ? v.pr[0].y : 0;
            v.ct0 = 0;
            v.have0 = false;
            v.size = e1.x + e0.w;
            const int state = commit_entry(E, J, C, K, v, k, car_j, ncar_j, S, K.links, nullptr);
            if (write_out) {
                for (int p = lane; p < e0.y; p += 64) {
                    const int root = A.tk_clusters[(size_t)base_i + parts[e0.x + p]];
                    A.cl_state[(size_t)base_i + root] = (int8_t)state;
                }
                if (state == 1) {
                    ++ndc;
                    ndp += e1.x;
                }
            }
        }
        if (write_out && lane == 0) {
            A.tk_scan[si * 4 + 2] = ndc;
            A.tk_scan[si * 4 + 3] = ndp;
        }
        wave_sync();
        const int nl = S.nl, n_created = S.n_created;

        // ---- the successor as the next `pre`: its car clusters that are still there in ascending name, then the created ones in
        // creation order; every one with the clouds appended to it ----
        const int ntot = ncar_j + n_created;
        for (int e = lane; e < ntot; e += 64) K.dsz[e] = 0;
        wave_sync();
        if (lane == 0)
            for (int l = 0; l < nl; ++l) {
                const int4 L = K.links[l];
                K.dsz[L.y] += L.z;
            }
        wave_sync();
        int4* nent_rec = K.ent[nxt];
        int32_t* nparts = K.parts[nxt];
        int run_e = 0, run_c = 0, run_p = 0, run_o = 0;
        for (int eb = 0; eb < ntot; eb += 64) {
            const int e = eb + lane;
            bool alive = false;
            int pc = 0, own = 0, csz = 0, pbeg_src = 0;
            if (e < ncar_j) {
                const int id = A.tk_crep[(size_t)base_j + e];
                alive = id >= 0 && !stamped(ld64(&E.lfwd[id]), E.epoch);
                pc = 1;
                own = A.cl_count[(size_t)base_j + car_j[e]];
            } else if (e < ntot) {
                const int4 m = K.cmeta[e - ncar_j];
                alive = !stamped(ld64(&E.lfwd[m.z]), E.epoch);
                pc = m.y;
                pbeg_src = m.x;
                for (int z = 0; z < m.y; ++z) own += A.cl_count[(size_t)base_j + car_j[ld32(&K.cparts[m.x + z])]];
            }
            if (e < ntot) csz = K.dsz[e];
            if (!alive) pc = own = csz = 0;
            const int ie = wave_incl_scan(alive ? 1 : 0), ic = wave_incl_scan(csz), ip = wave_incl_scan(pc), io = wave_incl_scan(own);
            if (e < ntot) K.eidx[e] = alive ? run_e + ie - 1 : -1;
            if (alive) {
                const int idx = run_e + ie - 1;
                if (idx < C.ws.cap_ent && run_p + ip <= C.ws.cap_ent) {
                    nent_rec[2 * idx] = make_int4(run_p + ip - pc, pc, run_c + ic - csz, csz);
                    nent_rec[2 * idx + 1] = make_int4(own, run_o + io - own, 0, 0);
                    if (e < ncar_j) {
                        nparts[run_p + ip - pc] = e;
                    } else {
                        for (int z = 0; z < pc; ++z) nparts[run_p + ip - pc + z] = ld32(&K.cparts[pbeg_src + z]);
                    }
                }
            }
            run_e += __shfl(ie, 63);
            run_c += __shfl(ic, 63);
            run_p += __shfl(ip, 63);
            run_o += __shfl(io, 63);
        }
        bool overflow = run_c > C.ws.cap_pool || run_e > C.ws.cap_ent || run_p > C.ws.cap_ent;
        if (overflow) {  // the state does not fit: flagged, the chain continues without the appended clouds
            if (lane == 0) atomicOr(&C.stats[0], run_c > C.ws.cap_pool ? 1 : 2);
            run_c = 0;
            run_e = min(run_e, C.ws.cap_ent);
            run_p = min(run_p, C.ws.cap_ent);
        }
        wave_sync();
        if (lane == 0) {
            // offsets of the appended clouds inside their cluster's region: in walking order of the sources
            for (int e = 0; e < ntot; ++e) K.dsz[e] = 0;  // cursors
            for (int l = 0; l < nl; ++l) {
                int4 L = K.links[l];
                const int idx = K.eidx[L.y];
                if (idx < 0 || overflow) {
                    L.w = -1;  // fused later in the same call (cloud_use only, ssc.cpp:1412) or dropped
                } else {
                    L.w = nent_rec[2 * idx].z + K.dsz[L.y];
                    K.dsz[L.y] += L.z;
                }
                K.links[l] = L;
            }
            if (overflow)
                for (int i = 0; i < run_e; ++i) {
                    int4 r = nent_rec[2 * i];
                    r.z = r.w = 0;
                    nent_rec[2 * i] = r;
                }
            K.hdr[H_NENT + nxt] = run_e;
            K.hdr[H_NCARRIED + nxt] = run_c;
            K.hdr[H_NPARTS + nxt] = run_p;
            sh.bc[0] = nl;
        }
    }
    __syncthreads();
    const int nl = sh.bc[0];

    // ---- phase 4: the appended clouds: cloud_use of the walked cluster's parts, transformed, then what it carried ----
    float4* npool = K.pool[nxt];
    for (int l = 0; l < nl; ++l) {
        const int4 L = K.links[l];
        if (L.w < 0) continue;
        const int4 e0 = ent[2 * L.x];
        int off = L.w;
        for (int p = 0; p < e0.y; ++p) {
            const int root = A.tk_clusters[(size_t)base_i + parts[e0.x + p]];
            const int mb = A.tk_mbegin[(size_t)base_i + root];
            const int cnt = A.cl_count[(size_t)base_i + root];
            for (int m = tid; m < cnt; m += kChThreads) {
                const int i = A.tk_members[(size_t)base_i + mb + m];
                float4 q;
                if (!from_apri) {
                    q = A.pts[base_i + A.apri_src[(size_t)base_i + i]];
                } else {
                    const scvod_apri& a = A.apri[(size_t)base_i + i];
                    q = make_float4(a.x, a.y, a.z, a.intensity);
                }
                const float x = T[0] * q.x + T[1] * q.y + T[2] * q.z + T[3];
                const float y = T[4] * q.x + T[5] * q.y + T[6] * q.z + T[7];
                const float z = T[8] * q.x + T[9] * q.y + T[10] * q.z + T[11];
                npool[off + m] = make_float4(x, y, z, q.w);
            }
            off += cnt;
        }
        for (int c = tid; c < e0.w; c += kChThreads) npool[off + c] = pool[e0.z + c];
    }
    __syncthreads();
}

// The same step when the walked clusters and the successor's car clusters fit the LDS tables (nent <= kLdsEnt,
// ncar_j + nent <= 2 kLdsEnt: every street scan): all threads fetch what the walk needs of every cluster at once -- the
// first-order remap_name head, the fresh |occupy_voxels| and type of its first label -- so that the sequential walk of one
// wave runs on LDS and touches HBM only where a re-labelling happened; the appended clouds are copied in one flat pass.
__device__ __forceinline__ void chain_step_small(const DevParams& P, const Arena& A, const TrackBatch& J, const ChainJob& C, const Wk& K, Shared& sh,
                                 uint32_t* bits_all, int cur, int si, int sj, bool write_out, int from_apri) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ShSmall& M = sh.sm;
    const int base_i = A.scan_off[si], base_j = A.scan_off[sj];
    const int nv = A.counts[sj * 8 + 6];
    const int ncar_j = A.tk_scan[sj * 4 + 0];
    const int4* tab = A.vox_track + base_j;
    const int nent = K.hdr[H_NENT + cur];
    if (write_out) {  // (see chain_step_big)
        const int ncar_i = A.tk_scan[si * 4 + 0];
        for (int o = tid; o < ncar_i; o += kChThreads) A.cl_state[(size_t)base_i + A.tk_clusters[(size_t)base_i + o]] = -1;
    }
    const int ncarried = K.hdr[H_NCARRIED + cur];
    const int4* ent = K.ent[cur];
    const int32_t* parts = K.parts[cur];
    float4* pool = K.pool[cur];
    const int nxt = cur ^ 1;
    const int32_t* car_j = A.tk_clusters + base_j;
    StepEnv E;
    E.tab = tab;
    E.rep = A.vox_rep + base_j;
    E.nv = nv;
    E.vlab = K.vlab;
    E.lcnt = K.lcnt;
    E.lfwd = K.lfwd;
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = J.T[12 * si + i];

    // ---- A: everything the step reads of the two frames, fetched by all threads at once ----
    CH_T0;
    int shift = 0;
    while (((nv + (1 << shift) - 1) >> shift) > kSmSamples) ++shift;
    const int ns = (nv + (1 << shift) - 1) >> shift;
    if (ncarried > 0)
        for (int j = tid; j < ns; j += kChThreads) M.skeys[j] = tab[(size_t)j << shift].x;
    int any_complex = 0;
    for (int k = tid; k < nent; k += kChThreads) {
        const int4 e0 = ent[2 * k], e1 = ent[2 * k + 1];
        if (e0.y == 1 && e0.w == 0) {
            const int root = A.tk_clusters[(size_t)base_i + parts[e0.x]];
            const int mb = A.tk_mbegin[(size_t)base_i + root];
            const int np = A.tk_npairs[(size_t)base_i + root];
            int L0 = -1, c0 = 0;
            uint32_t ct0 = 0;
            if (np >= 1) {
                L0 = A.tk_prep[(size_t)base_i + mb];
                c0 = A.tk_pairs[(size_t)base_i + mb].y;
                const int4 r = tab[L0];
                ct0 = (uint32_t)r.z | ((uint32_t)r.w << 28);
            }
            M.fr[k] = make_int4(np, A.tk_nuniq[(size_t)base_i + root], L0, c0);
            M.fx[k] = make_int4((int)ct0, root, e1.x, e1.x);
            M.fuo[k] = mb;
        } else {
            M.fr[k] = make_int4(0, 0, -1, 0);
            M.fx[k] = make_int4(0, -1, e1.x + e0.w, e1.x);
            M.fuo[k] = e1.y + e0.z;
            any_complex = 1;
        }
    }
    for (int e = tid; e < ncar_j; e += kChThreads) {
        M.nown[e] = A.cl_count[(size_t)base_j + car_j[e]];
        M.ncrep[e] = A.tk_crep[(size_t)base_j + e];
    }
    for (int e = tid; e < ncar_j + nent; e += kChThreads) M.dsz[e] = 0;
    if (tid == 0) K.hdr[H_EPOCH] = K.hdr[H_EPOCH] + 1;
    any_complex = __syncthreads_or(any_complex);
    E.epoch = (uint32_t)K.hdr[H_EPOCH];
    CH_MARK(0);

    // ---- B: the carried points ----
    if (ncarried > 0) {
        carried_probe(P, K, pool, ncarried, T, tab, nv, M.skeys, ns, shift);
        __syncthreads();
    }
    CH_MARK(1);

    // ---- C: clusters that carry appended points / fused clusters: their remap_name against the fresh successor ----
    if (any_complex) {
        const int nw = max((nv + 31) >> 5, 1);
        const int total_words = C.words * C.n_eval_waves;
        const int n_eval = min(kChWaves, total_words / nw);
        if (n_eval < 1) {
            if (tid == 0) atomicOr(&C.stats[0], 8);
        } else if (wave < n_eval) {
            uint32_t* bits = bits_all + (size_t)wave * nw;
            int cx = 0;
            for (int k = 0; k < nent; ++k) {
                if (M.fx[k].y >= 0) continue;
                if ((cx++ % n_eval) != wave) continue;
                const int4 e0 = ent[2 * k];
                int32_t* uq = K.suniq + M.fuo[k];
                int2* pr = K.spairs + M.fuo[k];
                const int2 r = eval_entry(A, K, E, base_i, parts, e0, bits, uq, pr);
                if (lane == 0) {
                    int L0 = -1, c0 = 0;
                    uint32_t ct0 = 0;
                    if (r.x >= 1) {
                        const int2 p0 = pr[0];
                        L0 = p0.x;
                        c0 = p0.y;
                        const int4 t = tab[L0];
                        ct0 = (uint32_t)t.z | ((uint32_t)t.w << 28);
                    }
                    M.fr[k] = make_int4(r.x, r.y, L0, c0);
                    M.fx[k].x = (int)ct0;
                }
            }
        }
        __syncthreads();
    }
    CH_MARK(2);

    // ---- D: the walk (one wave, LDS) and the successor's next state ----
    if (wave == 0) {
        StepCounters S = {0, 0, 0, 0, 0, max_name_label(A, C, E, sj)};
        int ndc = 0, ndp = 0;
        for (int k = 0; k < nent; ++k) {
            const int4 fr = M.fr[k], fx = M.fx[k];
            const int uo = M.fuo[k];
            EntEval v;
            v.np = fr.x;
            v.nu = fr.y;
            if (fx.y >= 0) {
                v.uq = A.tk_uniq + (size_t)base_i + uo;
                v.pr = A.tk_pairs + (size_t)base_i + uo;
                v.prid = A.tk_prep + (size_t)base_i + uo;
            } else {
                v.uq = K.suniq + uo;
                v.pr = K.spairs + uo;
                v.prid = nullptr;
            }
            v.L0 = fr.z;
            v.c0 = fr.w;
            v.ct0 = (uint32_t)fx.x;
            v.have0 = true;
            v.size = fx.z;
            const int state = commit_entry(E, J, C, K, v, k, car_j, ncar_j, S, M.lk, M.dsz);
            if (write_out) {
                if (fx.y >= 0) {
                    if (lane == 0) A.cl_state[(size_t)base_i + fx.y] = (int8_t)state;
                } else {
                    const int4 e0 = ent[2 * k];
                    for (int p = lane; p < e0.y; p += 64) A.cl_state[(size_t)base_i + A.tk_clusters[(size_t)base_i + parts[e0.x + p]]] = (int8_t)state;
                }
                if (state == 1) {
                    ++ndc;
                    ndp += fx.w;
                }
            }
        }
        if (write_out && lane == 0) {
            A.tk_scan[si * 4 + 2] = ndc;
            A.tk_scan[si * 4 + 3] = ndp;
        }
        wave_sync();
        const int nl = S.nl, n_created = S.n_created;
        const int ntot = ncar_j + n_created;
        int4* nent_rec = K.ent[nxt];
        int32_t* nparts = K.parts[nxt];
        int run_e = 0, run_c = 0, run_p = 0, run_o = 0;
        for (int eb = 0; eb < ntot; eb += 64) {
            const int e = eb + lane;
            bool alive = false;
            int pc = 0, own = 0, csz = 0, pbeg_src = 0;
            if (e < ncar_j) {
                const int id = M.ncrep[e];
                alive = id >= 0 && (S.n_dirty == 0 || !stamped(ld64(&E.lfwd[id]), E.epoch));
                pc = 1;
                own = M.nown[e];
            } else if (e < ntot) {
                const int4 m = K.cmeta[e - ncar_j];
                alive = !stamped(ld64(&E.lfwd[m.z]), E.epoch);
                pc = m.y;
                pbeg_src = m.x;
                for (int z = 0; z < m.y; ++z) own += M.nown[ld32(&K.cparts[m.x + z])];
            }
            if (e < ntot) csz = M.dsz[e];
            if (!alive) pc = own = csz = 0;
            const int ie = wave_incl_scan(alive ? 1 : 0), ic = wave_incl_scan(csz), ip = wave_incl_scan(pc), io = wave_incl_scan(own);
            if (e < ntot) M.eidx[e] = alive ? run_e + ie - 1 : -1;
            if (alive) {
                const int idx = run_e + ie - 1;
                M.ncb[idx] = run_c + ic - csz;
                nent_rec[2 * idx] = make_int4(run_p + ip - pc, pc, run_c + ic - csz, csz);
                nent_rec[2 * idx + 1] = make_int4(own, run_o + io - own, 0, 0);
                if (e < ncar_j) {
                    nparts[run_p + ip - pc] = e;
                } else {
                    for (int z = 0; z < pc; ++z) nparts[run_p + ip - pc + z] = ld32(&K.cparts[pbeg_src + z]);
                }
            }
            run_e += __shfl(ie, 63);
            run_c += __shfl(ic, 63);
            run_p += __shfl(ip, 63);
            run_o += __shfl(io, 63);
        }
        const bool overflow = run_c > C.ws.cap_pool;  // (entries and parts: <= 2 kLdsEnt <= cap_ent)
        if (overflow) {  // the appended clouds do not fit: flagged, the chain continues without them
            if (lane == 0) atomicOr(&C.stats[0], 1);
            for (int i = lane; i < run_e; i += 64) {
                int4 r = nent_rec[2 * i];
                r.z = r.w = 0;
                nent_rec[2 * i] = r;
            }
            run_c = 0;
        }
        wave_sync();
        for (int e = lane; e < ntot; e += 64) M.dsz[e] = 0;  // cursors
        wave_sync();
        if (lane == 0) {
            int tot = 0;
            for (int l = 0; l < nl; ++l) {
                int4 L = M.lk[l];
                const int idx = M.eidx[L.y];
                M.lpre[l] = tot;
                if (idx < 0 || overflow) {
                    L.w = -1;  // fused later in the same call (cloud_use only, ssc.cpp:1412) or dropped
                } else {
                    L.w = M.ncb[idx] + M.dsz[L.y];
                    M.dsz[L.y] += L.z;
                    tot += L.z;
                }
                M.lk[l] = L;
            }
            M.lpre[nl] = tot;
            K.hdr[H_NENT + nxt] = run_e;
            K.hdr[H_NCARRIED + nxt] = run_c;
            K.hdr[H_NPARTS + nxt] = run_p;
            sh.bc[0] = nl;
            sh.bc[1] = tot;
        }
    }
    __syncthreads();
    CH_MARK(3);

    // ---- F: the appended clouds in one flat pass: cloud_use of the walked cluster (transformed), then what it carried ----
    const int nl = sh.bc[0], total = sh.bc[1];
    float4* npool = K.pool[nxt];
#ifndef CH_UF
#define CH_UF 4
#endif
    constexpr int UF = CH_UF;  // four points per thread and round: the three dependent gathers of each overlap
    for (int t0 = tid; t0 < total; t0 += kChThreads * UF) {
        int dsti[UF], idx[UF];   // destination in the next pool; member slot (own point) or pool index (carried point)
        bool own[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const int t = t0 + u * kChThreads;
            dsti[u] = -1;
            idx[u] = 0;
            own[u] = false;
            if (t >= total) continue;
            int lo = 0, hi = nl;  // the last link whose offset is <= t: the one that holds point t (links that copy nothing repeat an offset)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (M.lpre[mid] <= t)
                    lo = mid;
                else
                    hi = mid;
            }
            const int4 L = M.lk[lo];
            int r = t - M.lpre[lo];
            dsti[u] = L.w + r;
            const int4 fx = M.fx[L.x];
            if (fx.y >= 0) {  // one original cluster, nothing carried: all of the link is own points
                own[u] = true;
                idx[u] = M.fuo[L.x] + r;
            } else if (r < fx.w) {
                const int4 e0 = ent[2 * L.x];
                int root = 0;
                for (int p = 0; p < e0.y; ++p) {
                    root = A.tk_clusters[(size_t)base_i + parts[e0.x + p]];
                    const int cnt = A.cl_count[(size_t)base_i + root];
                    if (r < cnt) break;
                    r -= cnt;
                }
                own[u] = true;
                idx[u] = A.tk_mbegin[(size_t)base_i + root] + r;
            } else {
                idx[u] = ent[2 * L.x].z + (r - fx.w);
            }
        }
        int mi[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) mi[u] = own[u] ? A.tk_members[(size_t)base_i + idx[u]] : 0;
        float4 q[UF];
        if (!from_apri) {
#pragma unroll
            for (int u = 0; u < UF; ++u) mi[u] = own[u] ? A.apri_src[(size_t)base_i + mi[u]] : 0;
#pragma unroll
            for (int u = 0; u < UF; ++u) q[u] = own[u] ? A.pts[base_i + mi[u]] : pool[idx[u]];
        } else {
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                if (own[u]) {
                    const scvod_apri& a = A.apri[(size_t)base_i + mi[u]];
                    q[u] = make_float4(a.x, a.y, a.z, a.intensity);
                } else {
                    q[u] = pool[idx[u]];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            if (dsti[u] < 0) continue;
            if (own[u]) {
                const float x = T[0] * q[u].x + T[1] * q[u].y + T[2] * q[u].z + T[3];
                const float y = T[4] * q[u].x + T[5] * q[u].y + T[6] * q[u].z + T[7];
                const float z = T[8] * q[u].x + T[9] * q[u].y + T[10] * q[u].z + T[11];
                q[u] = make_float4(x, y, z, q[u].w);
            }
            npool[dsti[u]] = q[u];
        }
    }
    __syncthreads();
    CH_MARK(4);
}

__device__ __forceinline__ void chain_step(const DevParams& P, const Arena& A, const TrackBatch& J, const ChainJob& C, const Wk& K, Shared& sh,
                                           uint32_t* bits_all, int cur, int si, int sj, bool write_out, int from_apri) {
    const int nent = K.hdr[H_NENT + cur];
    const int ncar_j = A.tk_scan[sj * 4 + 0];
    if (nent <= kLdsEnt && ncar_j + nent <= 2 * kLdsEnt && !C.force_generic)
        chain_step_small(P, A, J, C, K, sh, bits_all, cur, si, sj, write_out, from_apri);
    else
        chain_step_big(P, A, J, C, K, sh, bits_all, cur, si, sj, write_out, from_apri);
}

// walks steps [t_begin, t_end) of a chain on workspace K starting from the state in slot `cur`; returns the slot of the final state
__device__ __forceinline__ int walk(const DevParams& P, const Arena& A, const TrackBatch& J, const ChainJob& C, const Wk& K, Shared& sh,
                    uint32_t* bits, const ChainWalker& W, int cur, int t_begin, int t_end, int out_from, int snap_at, int from_apri,
                    int32_t* ticks = nullptr) {
    const int32_t* frames = C.chain_scans + W.first;
    long long t_prev = ticks ? wall_clock64() : 0;
    for (int t = t_begin; t < t_end; ++t) {
        if (t == snap_at) {
            copy_state(K, cur, K, 2);
            if (threadIdx.x == 0) K.hdr[H_HAS_SNAP] = 1;
        }
        chain_step(P, A, J, C, K, sh, bits, cur, frames[t], frames[t + 1], t >= out_from, from_apri);
        cur ^= 1;
        if (ticks) {  // what this step cost (thread 0's clock between the ends of two steps): the next batch's segments are cut by it
            const long long now = wall_clock64();
            if (threadIdx.x == 0 && t >= out_from) ticks[t] = (int32_t)min(now - t_prev, 0x7fffffffll);
            t_prev = now;
        }
    }
    return cur;
}

}  // namespace

// speculative pass: one workgroup per segment, warm-up from the fresh state of an earlier frame
__global__ __launch_bounds__(kChThreads) void k_tk_chain(DevParams P, Arena A, TrackBatch J, ChainJob C, int from_apri) {
    __shared__ Shared sh;
    extern __shared__ uint32_t ch_bits[];
    const ChainWalker W = C.walkers[blockIdx.x];
    const Wk K = wk_of(C.ws, blockIdx.x);
    for (int i = threadIdx.x; i < C.n_eval_waves * C.words; i += kChThreads) ch_bits[i] = 0u;
    if (threadIdx.x == 0) K.hdr[H_HAS_SNAP] = 0;
    __syncthreads();
    fresh_state(A, K, 0, C.chain_scans[W.first + W.t0], C.ws.cap_ent, C.stats);
#ifdef SCVOD_PROFILE
    const long long wt0 = wall_clock64();
#endif
    const int end = walk(P, A, J, C, K, sh, ch_bits, W, 0, W.t0, W.b, W.a, W.t0 < W.a ? W.a : -1, from_apri, C.step_ticks ? C.step_ticks + W.first : nullptr);
    if (threadIdx.x == 0) K.hdr[H_END_SLOT] = end;
#ifdef SCVOD_PROFILE
    if (threadIdx.x == 0) K.hdr[15] = (int)(wall_clock64() - wt0);
#endif
}

// verification, part 1: one workgroup per segment compares the state its predecessor ended in with its warm-up's snapshot
__global__ __launch_bounds__(kChThreads) void k_tk_chain_cmp(ChainJob C) {
    const int w = blockIdx.x;
    const ChainWalker W = C.walkers[w];
    const Wk K = wk_of(C.ws, w);
    int ok = 1;
    if (W.a > 0 && !W.ext) {  // (the first segment of a chain starts from the true state -- or from the one another shard sends later)
        const Wk Kp = wk_of(C.ws, w - 1);
        ok = K.hdr[H_HAS_SNAP] != 0 && same_state(Kp, Kp.hdr[H_END_SLOT], K, 2);
    }
    if (threadIdx.x == 0) {
        K.hdr[H_SNAP_OK] = ok;
        K.hdr[H_SPEC] = 0;
    }
}

// verification, part 1b: failed segments are usually isolated, and a segment whose PREDECESSOR passed its check can be walked
// again at once -- all of them concurrently, one workgroup each -- from the state that predecessor ended in: that state is the
// true one unless a segment further up the chain fails and its new end state cascades down to it, which part 2 notices (it
// compares again whenever a predecessor's end state changed) and then walks the segment once more.
__global__ __launch_bounds__(kChThreads) void k_tk_chain_spec(DevParams P, Arena A, TrackBatch J, ChainJob C, int from_apri) {
    __shared__ Shared sh;
    extern __shared__ uint32_t ch_bits[];
    const int w = blockIdx.x;
    const ChainWalker W = C.walkers[w];
    if (W.a == 0 || W.ext) return;  // the first segment of a chain
    const Wk K = wk_of(C.ws, w), Kp = wk_of(C.ws, w - 1);
    if (K.hdr[H_SNAP_OK] != 0 || Kp.hdr[H_SNAP_OK] == 0) return;  // passed / the predecessor is being walked again itself
    for (int i = threadIdx.x; i < C.n_eval_waves * C.words; i += kChThreads) ch_bits[i] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&C.stats[1], 1);
    copy_state(Kp, Kp.hdr[H_END_SLOT], K, 0);
    const int end = walk(P, A, J, C, K, sh, ch_bits, W, 0, W.a, W.b, W.a, -1, from_apri);
    if (threadIdx.x == 0) {
        K.hdr[H_END_SLOT] = end;
        K.hdr[H_SPEC] = 1;
    }
}

// verification, part 2: one workgroup per chain; a segment whose warm-up did not reproduce the state its predecessor really
// ended in is walked again from that state (and its successor compared again with the new end state)
__global__ __launch_bounds__(kChThreads) void k_tk_chain_fix(DevParams P, Arena A, TrackBatch J, ChainJob C, int from_apri) {
    __shared__ Shared sh;
    extern __shared__ uint32_t ch_bits[];
    const int w0 = C.chain_first_walker[blockIdx.x], w1 = C.chain_first_walker[blockIdx.x + 1];
    if (w1 <= w0) return;  // (a chain without a step of its own)
    // resume: the state this chain's first walker had to match has arrived from the shard that walked the frames before
    bool first_rewalked = false;
    if (C.resume) {
        const ChainWalker W0 = C.walkers[w0];
        const unsigned char* ext = C.ext_state ? C.ext_state[blockIdx.x] : nullptr;
        if (!ext) return;
        const Wk K0 = wk_of(C.ws, w0);
        const ExtRec R = ext_of(ext);
        if (threadIdx.x == 0 && C.resume == 1) atomicAdd(&C.stats[2], 1);
        // a chain that continues another shard WITHOUT a halo (no warm-up step in front of its block: W0.ext == 0) started from the fresh
        // segmentation of its first frame and has no snapshot to compare: the received state is what it has to start from, so it
        // counts as "differs" and is walked again from that state (round-4 advice: it used to be ignored silently)
        const bool ok = R.hdr[3] == 1 && W0.ext != 0 && K0.hdr[H_HAS_SNAP] != 0 && same_state_ext(K0, 2, R);
        if (C.resume == 2) {  // compare only (scvod_batch_track_compare); a record that did not fit its exchange buffer (hdr[3] == 2) counts as a
            // difference, and so does a row nobody wrote (hdr[3] == 0: the sender had no state for this sub-sequence, or its row was never
            // exported) -- whoever hands a pointer over EXPECTS a state there (round-5 advice: such a chain used to pass unverified)
            if (threadIdx.x == 0 && !ok) atomicAdd(C.cmp_out, 1);
            return;
        }
        if (ok) return;  // the warm-up reproduced it: everything behind stands
        if (R.hdr[3] != 1) return;  // (no state to start from: nothing to do)
        for (int i = threadIdx.x; i < C.n_eval_waves * C.words; i += kChThreads) ch_bits[i] = 0u;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&C.stats[1], 1);
        copy_from_ext(R, K0, 0, C.ws.cap_ent, C.ws.cap_pool, C.stats);
        const int end = walk(P, A, J, C, K0, sh, ch_bits, W0, 0, W0.a, W0.b, W0.a, -1, from_apri);
        if (threadIdx.x == 0) K0.hdr[H_END_SLOT] = end;
        __syncthreads();
        first_rewalked = true;
    }
    // the verdicts of k_tk_chain_cmp, fetched by all threads at once: next_bad[i] = first walker >= w0 + 1 + i whose check failed
    constexpr int kFixLds = 1024;
    __shared__ int32_t next_bad[kFixLds + 1];
    const int nchk = min(w1 - w0 - 1, kFixLds);
    int all_ok = 1;
    for (int i = threadIdx.x; i < w1 - w0 - 1; i += kChThreads) {
        const int ok = wk_of(C.ws, w0 + 1 + i).hdr[H_SNAP_OK];
        all_ok &= ok;
        if (i < nchk) next_bad[i] = ok ? 0x7fffffff : w0 + 1 + i;
    }
    all_ok = __syncthreads_and(all_ok);
    if (threadIdx.x == 0 && !C.resume) atomicAdd(&C.stats[2], w1 - w0 - 1);
    if (all_ok && !first_rewalked) return;
    if (threadIdx.x == 0) {  // suffix minimum (a chain has a few hundred segments)
        next_bad[nchk] = (w0 + 1 + nchk < w1) ? w0 + 1 + nchk : 0x7fffffff;  // (beyond the table: taken one by one)
        for (int i = nchk - 1; i >= 0; --i) next_bad[i] = min(next_bad[i], next_bad[i + 1]);
    }
    for (int i = threadIdx.x; i < C.n_eval_waves * C.words; i += kChThreads) ch_bits[i] = 0u;
    __syncthreads();
    int w = w0 + 1;
    bool prev_rewalked = first_rewalked;
    while (w < w1) {
        if (!prev_rewalked) {  // jump to the next segment whose check failed
            const int i = w - (w0 + 1);
            const int nb = i <= nchk ? next_bad[i] : w;
            if (nb >= w1) break;
            w = nb;
        }
        const ChainWalker W = C.walkers[w];
        const Wk K = wk_of(C.ws, w), Kp = wk_of(C.ws, w - 1);
        const int pend = Kp.hdr[H_END_SLOT];
        const bool pred_changed = prev_rewalked;
        bool ok;
        if (pred_changed)  // the predecessor's end state changed: compare again
            ok = K.hdr[H_HAS_SNAP] != 0 && same_state(Kp, pend, K, 2);
        else
            ok = K.hdr[H_SNAP_OK] != 0;
        prev_rewalked = !ok;  // (either way this segment's end state is not the one its successor was compared with)
        if (ok) {
            ++w;
            continue;
        }
        // already walked again by k_tk_chain_spec from the state its predecessor still ends in: that walk stands
        if (!(K.hdr[H_SPEC] != 0 && !pred_changed)) {
            if (threadIdx.x == 0) atomicAdd(&C.stats[1], 1);
            copy_state(Kp, pend, K, 0);
            const int end = walk(P, A, J, C, K, sh, ch_bits, W, 0, W.a, W.b, W.a, -1, from_apri);
            if (threadIdx.x == 0) K.hdr[H_END_SLOT] = end;
        }
        __syncthreads();
        ++w;
    }
}

size_t chain_state_bytes(const ChainWs& ws) { return 16 + (size_t)32 * ws.cap_ent + (((size_t)4 * ws.cap_ent + 15) & ~(size_t)15) + (size_t)16 * ws.cap_pool; }

__global__ __launch_bounds__(kChThreads) void k_tk_chain_export(ChainJob C, int chain, int which, unsigned char* dst, long long cap_bytes) {
    const int w0 = C.chain_first_walker[chain], w1 = C.chain_first_walker[chain + 1];
    int32_t* hdr = (int32_t*)dst;
    bool valid = w1 > w0;
    Wk K = wk_of(C.ws, valid ? (which == 0 ? w0 : w1 - 1) : 0);
    int slot = 0;
    if (valid) {
        if (which == 0) {
            valid = C.walkers[w0].ext != 0 && K.hdr[H_HAS_SNAP] != 0;
            slot = 2;
        } else {
            slot = K.hdr[H_END_SLOT];
        }
    }
    const int ne = valid ? K.hdr[H_NENT + slot] : 0, nc = valid ? K.hdr[H_NCARRIED + slot] : 0, np = valid ? K.hdr[H_NPARTS + slot] : 0;
    const size_t need = 16 + (size_t)32 * ne + (((size_t)4 * np + 15) & ~(size_t)15) + (size_t)16 * nc;
    const bool too_large = valid && (long long)need > cap_bytes;  // (a fixed-size exchange buffer: the receiver treats it as "differs" and asks for the full record)
    if (too_large) valid = false;
    if (threadIdx.x == 0) {
        hdr[0] = valid ? ne : 0;
        hdr[1] = valid ? nc : 0;
        hdr[2] = valid ? np : 0;
        hdr[3] = valid ? 1 : (too_large ? 2 : 0);
    }
    if (!valid) return;
    int4* ent = (int4*)(dst + 16);
    int32_t* parts = (int32_t*)(dst + 16 + (size_t)32 * ne);
    float4* pool = (float4*)(dst + 16 + (size_t)32 * ne + (((size_t)4 * np + 15) & ~(size_t)15));
    for (int i = threadIdx.x; i < 2 * ne; i += kChThreads) ent[i] = K.ent[slot][i];
    for (int i = threadIdx.x; i < np; i += kChThreads) parts[i] = K.parts[slot][i];
    for (int i = threadIdx.x; i < nc; i += kChThreads) pool[i] = K.pool[slot][i];
}

void launch_chain_export_state(const ChainJob& C, int chain, int which, unsigned char* dst, long long cap_bytes, hipStream_t st) {
    hipLaunchKernelGGL(k_tk_chain_export, dim3(1), dim3(kChThreads), 0, st, C, chain, which, dst, cap_bytes);
}

void launch_track_chain_resume(const DevParams& P, const Arena& A, const TrackBatch& J, const ChainJob& C, int from_apri, hipStream_t st) {
    if (C.n_walkers <= 0) return;
    const size_t dyn = (size_t)C.n_eval_waves * C.words * 4;
    hipFuncSetAttribute((const void*)k_tk_chain_fix, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    hipLaunchKernelGGL(k_tk_chain_fix, dim3(C.n_chains), dim3(kChThreads), dyn, st, P, A, J, C, from_apri);
}

void launch_track_chain(const DevParams& P, const Arena& A, const TrackBatch& J, const ChainJob& C, int from_apri, hipStream_t st,
                        TimerHook th, void* tu) {
    if (C.n_walkers <= 0) return;
    const size_t dyn = (size_t)C.n_eval_waves * C.words * 4;
    hipFuncSetAttribute((const void*)k_tk_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    hipFuncSetAttribute((const void*)k_tk_chain_fix, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    hipFuncSetAttribute((const void*)k_tk_chain_spec, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    if (th) th(tu, "tk_chain", 1);
    hipLaunchKernelGGL(k_tk_chain, dim3(C.n_walkers), dim3(kChThreads), dyn, st, P, A, J, C, from_apri);
    if (th) th(tu, "tk_chain", 0);
    if (th) th(tu, "tk_chain_fix", 1);
    hipLaunchKernelGGL(k_tk_chain_cmp, dim3(C.n_walkers), dim3(kChThreads), 0, st, C);
    hipLaunchKernelGGL(k_tk_chain_spec, dim3(C.n_walkers), dim3(kChThreads), dyn, st, P, A, J, C, from_apri);
    hipLaunchKernelGGL(k_tk_chain_fix, dim3(C.n_chains), dim3(kChThreads), dyn, st, P, A, J, C, from_apri);
    if (th) th(tu, "tk_chain_fix", 0);
}

}  // namespace scvod
