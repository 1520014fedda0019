// scvod_map.hip -- world-frame static map of a sequence shard, and its merge across shards (gfx950).
//
// Reference analogue: the sequence-level accumulation `*instance_map += *rgb_ptr` of the clouds that are not dynamic
// (SSC::saveSegCloud mode 3, /root/reference/src/ssc.cpp:477-500, 532-554) together with the ground clouds and the
// range/FOV rejects the evaluation adds (ssc.cpp:1460-1480 `g_cloud_vec`, `cloud_eva_static` ssc.cpp:161-172), every
// scan moved to the world frame by pcl::getTransformation(pose) (ssc.cpp:1455-1458).  The reference appends points to one
// growing PCL cloud on one thread; a sequence sharded over GPUs needs a merge that does not depend on which shard saw a
// point first, so the map is kept as a SET OF OCCUPIED CELLS (edge `leaf`), one record per cell:
//     key = the cell's integer coordinates, val = the smallest packed {x, y, z offset inside the cell, intensity} of the
//     points that fell into it (16 bits each) -- an actual measured point, chosen by a rule that is independent of order.
// Insertion is one 64-bit atomicMin per point on an open-addressing table in HBM (linear probing, 16-byte records), so
// accumulating scans in any order, on any number of shards, and merging the shards' record lists (scvod_map_merge, the
// payload of the RCCL all_gather) gives bit-identical maps.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "scvod_dev.h"

using namespace scvod;

struct MapRec {
    unsigned long long key;  // ~0 = empty
    unsigned long long val;  // packed {qx, qy, qz, qi}, 16 bits each, smallest wins
};
static_assert(sizeof(MapRec) == 16, "map record");

struct scvod_map {
    int device = 0;
    float leaf = 0.2f;
    long long capacity = 0;  // power of two
    MapRec* table = nullptr;
    unsigned long long* counters = nullptr;  // [0] records exported, [1] insertions dropped (table full / out of range), [2..2+64) per-part counts / cursors
    float* d_pose = nullptr;                 // [pose_cap][12]
    long long pose_cap = 0;
    std::vector<float> up_pose;
    std::string err;
};

namespace scvod {
namespace {

constexpr unsigned long long kEmpty = ~0ull;
constexpr int kMapMaxProbes = 512;  // linear probing: longer runs only appear beyond ~95 % load
constexpr int kCellBits = 21, kCellBias = 1 << 20;
// k_map_accumulate: points per thread and round / points per workgroup.  Measured on the 2761-scan K64 job (round 3): one point per
// thread beats two / four / eight in flight (4.82 / 4.96 / 5.43 / 6.19 ms at 2048 points per workgroup: the extra registers cost
// occupancy and the later atomics act on staler peeks), and 8192 points per workgroup beat 512 .. 32768 (5.03 .. 4.45 .. 4.58 ms)
constexpr int kMapU = 1, kMapPts = 8192;

int mfail(scvod_map* m, int code, const char* fmt, ...) {
    if (m) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        m->err = buf;
    }
    return code;
}
#define MHIP(m, call)                                                                                    \
    do {                                                                                                 \
        hipError_t e__ = (call);                                                                         \
        if (e__ != hipSuccess) return mfail(m, SCVOD_ERR_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

__device__ __forceinline__ unsigned long long map_mix(unsigned long long k) {  // murmur3 finaliser
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

// home slot of a cell: a hash of the whole cell.  (Blocks of neighbouring cells laid side by side in the table were measured in rounds 3
// and 5, from 2 x 2 x 1 to 4 x 4 x 4 cells: every one slower -- profiles/r05_experiments.md.)
__device__ __forceinline__ unsigned long long map_home(unsigned long long key, unsigned long long mask) { return map_mix(key) & mask; }

// cell + packed in-cell offset of a world point.  floor(p * inv_leaf) like pcl::VoxelGrid's cell index; the offset is
// quantised to 16 bits of the cell edge, the intensity to 1/256 (clamped to [0, 255.996]).
__device__ __forceinline__ bool map_encode(float x, float y, float z, float intensity, float inv_leaf, unsigned long long& key,
                                           unsigned long long& val) {
    const float fx = x * inv_leaf, fy = y * inv_leaf, fz = z * inv_leaf;
    const float cx = floor_f(fx), cy = floor_f(fy), cz = floor_f(fz);
    if (!(cx >= -(float)kCellBias && cx < (float)kCellBias && cy >= -(float)kCellBias && cy < (float)kCellBias &&
          cz >= -(float)kCellBias && cz < (float)kCellBias))
        return false;  // also NaN
    const unsigned long long ux = (unsigned long long)((int)cx + kCellBias), uy = (unsigned long long)((int)cy + kCellBias),
                             uz = (unsigned long long)((int)cz + kCellBias);
    key = (ux << (2 * kCellBits)) | (uy << kCellBits) | uz;
    auto q16 = [](float f) -> unsigned long long {
        int q = (int)(f * 65536.0f);
        return (unsigned long long)(q < 0 ? 0 : (q > 65535 ? 65535 : q));
    };
    float in = intensity * 256.0f;
    in = in < 0.f ? 0.f : (in > 65535.f ? 65535.f : in);
    val = (q16(fx - cx) << 48) | (q16(fy - cy) << 32) | (q16(fz - cz) << 16) | (unsigned long long)(int)in;
    return true;
}

// One 16-byte load per probe (key and value together).  The load is an ordinary cached one: a key, once set, never changes
// and a value only decreases, so a stale record can only show (a) an empty slot that is taken by now -- the CAS then returns
// the owner -- or (b) a value above the current one -- the atomicMin then runs although it was not needed.  Both are harmless.
__device__ __forceinline__ MapRec map_peek(const MapRec* table, unsigned long long h) {
    const ulonglong2 r = *reinterpret_cast<const ulonglong2*>(&table[h]);
    MapRec m;
    m.key = r.x;
    m.val = r.y;
    return m;
}
// `first` = map_peek(table, h0) with h0 = map_mix(key) & mask (the caller issued it early, several probes in flight)
__device__ __forceinline__ bool map_insert_from(MapRec* table, unsigned long long mask, unsigned long long key, unsigned long long val,
                                                unsigned long long h, MapRec rec) {
    // a table that is (nearly) full is an error the caller hears about at export time, not a reason to walk 2^n slots
    for (int probes = 0; probes < kMapMaxProbes; ++probes) {
        unsigned long long k = rec.key;
        if (k == kEmpty) {
            k = atomicCAS(&table[h].key, kEmpty, key);
            if (k == kEmpty) {
                k = key;
                rec.val = kEmpty;  // a fresh record
            }
        }
        if (k == key) {
            // most points find a cell an earlier scan opened and a value that already beats theirs: test before the atomic
            if (val < rec.val) atomicMin(&table[h].val, val);
            return true;
        }
        h = (h + 1) & mask;
        rec = map_peek(table, h);
    }
    return false;
}
__device__ __forceinline__ bool map_insert(MapRec* table, unsigned long long mask, unsigned long long key, unsigned long long val) {
    const unsigned long long h = map_home(key, mask);
    return map_insert_from(table, mask, key, val, h, map_peek(table, h));
}

// static points of scan blockIdx.y: cloud_out (ground), cloud_eva_static (range/FOV rejects) and the apri points that are
// not dynamic, moved to the world frame with explicit fp32 dot products (the arithmetic of Utility::transformCloud,
// utility.h:400-405).  The scan is streamed IN INPUT ORDER (coalesced 16-byte loads; which cloud a point belongs to comes
// from the byte k_emit / k_tk_dyn left per input point): a sweep's neighbours in memory are neighbours in space, so runs of
// consecutive lanes that fall into the same cell are reduced in the wave first (segmented min over the run) and only the
// head of a run probes the table in HBM.
__global__ __launch_bounds__(256) void k_map_accumulate(int s_first, Arena A, const float* __restrict__ pose, MapRec* table, unsigned long long mask,
                                                        float inv_leaf, int with_ground, int with_rejected, int have_dyn, int have_pid,
                                                        int min_pts, int marks, int part, unsigned long long* counters) {
    const int s = s_first + blockIdx.y;
    const int base = A.scan_off[s];
    const int n = A.scan_off[s + 1] - base;
    const int lane = threadIdx.x & 63;
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = pose[12 * s + i];
    int dropped = 0;
    constexpr int U = kMapU;  // points per thread and round (see kMapU)
    for (int i0 = blockIdx.x * (256 * U); i0 < n; i0 += gridDim.x * (256 * U)) {
        int pid[U], pcnt[U];
        uint8_t cls[U];
        float4 q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(i0 + u * 256 + (int)threadIdx.x, n - 1);
            // the scan is read once: non-temporal loads keep it out of the way of the table's sectors in L2 (4.59 -> 4.39 / 4.61 -> 4.57 ms)
            pid[u] = have_pid ? (int)__builtin_nontemporal_load(&A.pid[(size_t)base + i]) : 0;
            cls[u] = marks ? __builtin_nontemporal_load(&A.pt_mapcls[(size_t)base + i]) : (uint8_t)0;
            const float* pp = reinterpret_cast<const float*>(&A.pts[base + i]);
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v v4 = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(pp));
            q[u] = make_float4(v4.x, v4.y, v4.z, v4.w);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) pcnt[u] = have_pid ? A.patch_count[s * kMaxPatches + max(pid[u], 0)] : 0;
        unsigned long long key[U], val[U], h[U];
        bool need[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256 + (int)threadIdx.x;
            key[u] = kEmpty;
            val[u] = kEmpty;
            if (i < n) {
                // in a cloud at all?  Patchwork drops the points outside its range gate / below the z cut (pid < 0) and whole
                // patches of at most num_min_pts points (patchwork.h:331); a batch binned without Patchwork keeps every point
                bool keep = !have_pid || (pid[u] >= 0 && pcnt[u] > min_pts);
                if (keep && marks) {
                    const uint8_t c = cls[u];
                    keep = !((c & kMapDynamic) && have_dyn) && !((c & kMapGround) && !with_ground) && !((c & kMapRejected) && !with_rejected);
                    // part 1: everything no tracking result can remove (not a car-cluster member); part 2: the car-cluster members
                    if (part == 1) keep = keep && !(c & kMapCar);
                    if (part == 2) keep = keep && (c & kMapCar);
                }
                if (keep) {
                    const float x = T[0] * q[u].x + T[1] * q[u].y + T[2] * q[u].z + T[3];
                    const float y = T[4] * q[u].x + T[5] * q[u].y + T[6] * q[u].z + T[7];
                    const float z = T[8] * q[u].x + T[9] * q[u].y + T[10] * q[u].z + T[11];
                    if (!map_encode(x, y, z, q[u].w, inv_leaf, key[u], val[u])) {
                        ++dropped;
                        key[u] = kEmpty;
                    }
                }
            }
            // runs of equal keys among consecutive lanes: the head of a run takes the smallest value of the run
            const unsigned long long prev = __shfl_up(key[u], 1);
            const bool head = (lane == 0) || (prev != key[u]);
            const unsigned long long heads = __ballot(head);
            // end of my run = position of the next head after my lane (exclusive), 64 if none
            const unsigned long long after = heads & ~((2ull << lane) - 1ull);
            const int run_end = after ? (__ffsll((long long)after) - 1) : 64;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned long long o = __shfl_down(val[u], d);
                if (lane + d < run_end && o < val[u]) val[u] = o;
            }
            need[u] = head && key[u] != kEmpty;
            h[u] = map_home(key[u], mask);
        }
        MapRec first[U];
#pragma unroll
        for (int u = 0; u < U; ++u) first[u] = map_peek(table, need[u] ? h[u] : 0ull);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (need[u] && !map_insert_from(table, mask, key[u], val[u], h[u], first[u])) ++dropped;
    }
    if (dropped) atomicAdd(&counters[1], (unsigned long long)dropped);
}

// only for maps without the ground or without the range/FOV rejects: mark the members of those two lists
__global__ __launch_bounds__(256) void k_map_mark_lists(Arena A) {
    const int s = blockIdx.y;
    const int base = A.scan_off[s];
    const int n_g = A.counts[s * 8 + 1], n_r = A.counts[s * 8 + 5];
    for (int t = blockIdx.x * 256 + threadIdx.x; t < n_g + n_r; t += gridDim.x * 256) {
        if (t < n_g)
            A.pt_mapcls[(size_t)base + A.ground_idx[(size_t)base + t]] = kMapGround;
        else
            A.pt_mapcls[(size_t)base + A.rejected_src[(size_t)base + (t - n_g)]] = kMapRejected;
    }
}

__global__ __launch_bounds__(256) void k_map_merge(const MapRec* __restrict__ recs, long long n, MapRec* table, unsigned long long mask,
                                                   unsigned long long* counters) {
    int dropped = 0;
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < n; t += (long long)gridDim.x * 256) {
        const MapRec r = recs[t];
        if (r.key == kEmpty) continue;  // padding of a gathered list
        if (!map_insert(table, mask, r.key, r.val)) ++dropped;
    }
    if (dropped) atomicAdd(&counters[1], (unsigned long long)dropped);
}

// occupied records -> dense list (order = whatever the atomics give: consumers that need a canonical order sort by key)
__global__ __launch_bounds__(256) void k_map_export(const MapRec* __restrict__ table, long long capacity, MapRec* out, long long cap_out,
                                                    float4* out_xyzi, float leaf, unsigned long long* counters) {
    __shared__ unsigned int cnt;
    __shared__ unsigned long long gstart;
    constexpr int IT = 8;  // slots per thread and round: one reservation on the shared counter per 2048 slots, not per wave
    const int lane = threadIdx.x & 63;
    for (long long t0 = blockIdx.x * (256ll * IT); t0 < capacity; t0 += (long long)gridDim.x * (256 * IT)) {
        if (threadIdx.x == 0) cnt = 0;
        __syncthreads();
        MapRec r[IT];
        int loc[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const long long t = t0 + it * 256 + threadIdx.x;
            r[it].key = kEmpty;
            r[it].val = kEmpty;
            if (t < capacity) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&table[t]);
                r[it].key = v.x;
                r[it].val = v.y;
            }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const bool occ = r[it].key != kEmpty;
            const unsigned long long bal = __ballot(occ);
            unsigned int wbase = 0;
            if (lane == 0 && bal) wbase = atomicAdd(&cnt, (unsigned int)__popcll(bal));
            wbase = __shfl(wbase, 0);
            loc[it] = occ ? (int)(wbase + __popcll(bal & ((1ull << lane) - 1ull))) : -1;
        }
        __syncthreads();
        if (threadIdx.x == 0 && cnt) gstart = atomicAdd(&counters[0], (unsigned long long)cnt);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            if (loc[it] < 0) continue;
            const long long o = (long long)gstart + loc[it];
            if (o >= cap_out) continue;
            const MapRec& q = r[it];
            if (out) *reinterpret_cast<ulonglong2*>(&out[o]) = make_ulonglong2(q.key, q.val);
            if (out_xyzi) {
                const int cx = (int)(q.key >> (2 * kCellBits)) - kCellBias, cy = (int)((q.key >> kCellBits) & ((1u << kCellBits) - 1u)) - kCellBias,
                          cz = (int)(q.key & ((1u << kCellBits) - 1u)) - kCellBias;
                const float qx = (float)((q.val >> 48) & 0xffffu), qy = (float)((q.val >> 32) & 0xffffu), qz = (float)((q.val >> 16) & 0xffffu);
                out_xyzi[o] = make_float4(((float)cx + (qx + 0.5f) / 65536.0f) * leaf, ((float)cy + (qy + 0.5f) / 65536.0f) * leaf,
                                          ((float)cz + (qz + 0.5f) / 65536.0f) * leaf, (float)(q.val & 0xffffu) / 256.0f);
            }
        }
        __syncthreads();
    }
}

// owner of a cell when the map is reduce-scattered over `n_parts` shards (a mix that is independent of the table hash)
__device__ __forceinline__ int map_part(unsigned long long key, int n_parts) {
    return (int)((map_mix(key ^ 0x9e3779b97f4a7c15ull) >> 33) % (unsigned long long)n_parts);
}
constexpr int kMapMaxParts = 64;

__global__ __launch_bounds__(256) void k_map_count_parts(const MapRec* __restrict__ table, long long capacity, int n_parts, unsigned long long* part_count) {
    __shared__ int hist[kMapMaxParts];
    if (threadIdx.x < kMapMaxParts) hist[threadIdx.x] = 0;
    __syncthreads();
    for (long long t = blockIdx.x * 256ll + threadIdx.x; t < capacity; t += (long long)gridDim.x * 256) {
        const unsigned long long k = table[t].key;
        if (k != kEmpty) atomicAdd(&hist[map_part(k, n_parts)], 1);
    }
    __syncthreads();
    if (threadIdx.x < n_parts && hist[threadIdx.x]) atomicAdd(&part_count[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

// records grouped by owner: a block ranks its occupied slots per part in LDS, reserves one range per part, writes
__global__ __launch_bounds__(256) void k_map_scatter_parts(const MapRec* __restrict__ table, long long capacity, int n_parts,
                                                           unsigned long long* part_cursor, MapRec* out, long long cap_out) {
    __shared__ int hist[kMapMaxParts];
    __shared__ unsigned long long start[kMapMaxParts];
    constexpr int IT = 8;
    for (long long t0 = blockIdx.x * (256ll * IT); t0 < capacity; t0 += (long long)gridDim.x * (256 * IT)) {
        if (threadIdx.x < kMapMaxParts) hist[threadIdx.x] = 0;
        __syncthreads();
        MapRec r[IT];
        int part[IT], rank[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const long long t = t0 + it * 256 + threadIdx.x;
            r[it].key = kEmpty;
            r[it].val = kEmpty;
            if (t < capacity) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&table[t]);
                r[it].key = v.x;
                r[it].val = v.y;
            }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            part[it] = -1;
            rank[it] = 0;
            if (r[it].key != kEmpty) {
                part[it] = map_part(r[it].key, n_parts);
                rank[it] = atomicAdd(&hist[part[it]], 1);
            }
        }
        __syncthreads();
        if (threadIdx.x < n_parts && hist[threadIdx.x]) start[threadIdx.x] = atomicAdd(&part_cursor[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            if (part[it] < 0) continue;
            const long long o = (long long)start[part[it]] + rank[it];
            if (o < cap_out) *reinterpret_cast<ulonglong2*>(&out[o]) = make_ulonglong2(r[it].key, r[it].val);
        }
        __syncthreads();
    }
}

// the same grouping into FIXED-SIZE slots: group p owns out[p * cap_part, (p + 1) * cap_part); the caller pre-fills the buffer
// with padding (key ~0).  Cursors start at 0; a group that outgrows its slot is counted in counters[1] (reported at the next
// synchronising call).  No size has to be known on the host: the all-to-all of a sharded map moves equal-sized slots.
__global__ __launch_bounds__(256) void k_map_scatter_parts_padded(const MapRec* __restrict__ table, long long capacity, int n_parts,
                                                                  unsigned long long* part_cursor, MapRec* out, long long cap_part,
                                                                  unsigned long long* counters) {
    __shared__ int hist[kMapMaxParts];
    __shared__ unsigned long long start[kMapMaxParts];
    constexpr int IT = 8;  // slots per thread and round: eight 16-byte loads in flight, one reservation per part and 2048 slots
    int dropped = 0;
    for (long long t0 = blockIdx.x * (256ll * IT); t0 < capacity; t0 += (long long)gridDim.x * (256 * IT)) {
        if (threadIdx.x < kMapMaxParts) hist[threadIdx.x] = 0;
        __syncthreads();
        MapRec r[IT];
        int part[IT], rank[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const long long t = t0 + it * 256 + threadIdx.x;
            r[it].key = kEmpty;
            r[it].val = kEmpty;
            if (t < capacity) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&table[t]);
                r[it].key = v.x;
                r[it].val = v.y;
            }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            part[it] = -1;
            rank[it] = 0;
            if (r[it].key != kEmpty) {
                part[it] = map_part(r[it].key, n_parts);
                rank[it] = atomicAdd(&hist[part[it]], 1);
            }
        }
        __syncthreads();
        if (threadIdx.x < n_parts && hist[threadIdx.x]) start[threadIdx.x] = atomicAdd(&part_cursor[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            if (part[it] < 0) continue;
            const long long o = (long long)start[part[it]] + rank[it];
            if (o < cap_part)
                *reinterpret_cast<ulonglong2*>(&out[(long long)part[it] * cap_part + o]) = make_ulonglong2(r[it].key, r[it].val);
            else
                ++dropped;
        }
        __syncthreads();
    }
    if (dropped) atomicAdd(&counters[1], (unsigned long long)dropped);
}

}  // namespace
}  // namespace scvod

// the one place the map code needs the batch context: its arena, parameters and validity flags
struct scvod_ctx;
extern "C" int scvod__ctx_types_valid(scvod_ctx* c);
extern "C" int scvod__ctx_view(scvod_ctx* ctx, Arena* arena, int* device, int* track_valid, int* batch_valid, int* n_scans,
                               int* max_scan_pts, int* batch_mode, int* num_min_pts);

extern "C" {

int scvod_map_create(int device, int64_t capacity_cells, float leaf, scvod_map** out) {
    if (!out || capacity_cells < 1024 || !(leaf > 0.f)) return SCVOD_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SCVOD_ERR_NO_DEVICE;
    scvod_map* m = new scvod_map();
    m->device = device;
    m->leaf = leaf;
    long long cap = 1024;
    while (cap < capacity_cells) cap <<= 1;
    m->capacity = cap;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&m->table, sizeof(MapRec) * (size_t)cap) != hipSuccess ||
        hipMalloc(&m->counters, 8 * (2 + 64)) != hipSuccess || hipMemset(m->table, 0xff, sizeof(MapRec) * (size_t)cap) != hipSuccess ||
        hipMemset(m->counters, 0, 8 * (2 + 64)) != hipSuccess) {
        if (m->table) hipFree(m->table);
        if (m->counters) hipFree(m->counters);
        delete m;
        return SCVOD_ERR_HIP;
    }
    *out = m;
    return SCVOD_OK;
}

void scvod_map_destroy(scvod_map* m) {
    if (!m) return;
    hipSetDevice(m->device);
    hipDeviceSynchronize();
    if (m->table) hipFree(m->table);
    if (m->counters) hipFree(m->counters);
    if (m->d_pose) hipFree(m->d_pose);
    delete m;
}

const char* scvod_map_last_error(const scvod_map* m) { return m ? m->err.c_str() : "null map"; }
int64_t scvod_map_capacity(const scvod_map* m) { return m ? (int64_t)m->capacity : 0; }

int scvod_map_clear(scvod_map* m, void* stream) {
    if (!m) return SCVOD_ERR_INVALID;
    MHIP(m, hipSetDevice(m->device));
    MHIP(m, hipMemsetAsync(m->table, 0xff, sizeof(MapRec) * (size_t)m->capacity, (hipStream_t)stream));
    MHIP(m, hipMemsetAsync(m->counters, 0, 16, (hipStream_t)stream));
    return SCVOD_OK;
}

// pcl::getTransformation(x, y, z, roll, pitch, yaw) as a row-major 3x4 matrix (the arithmetic scvod_pose_delta uses)
void scvod_pose_matrix(const float p[6], float t[12]) {
    const float x = p[0], y = p[1], z = p[2], roll = p[3], pitch = p[4], yaw = p[5];
    const float A = std::cos(yaw), B = std::sin(yaw), C = std::cos(pitch), D = std::sin(pitch), E = std::cos(roll), F = std::sin(roll),
                DE = D * E, DF = D * F;
    t[0] = A * C;
    t[1] = A * DF - B * E;
    t[2] = B * F + A * DE;
    t[3] = x;
    t[4] = B * C;
    t[5] = A * E + B * DF;
    t[6] = B * DE - A * F;
    t[7] = y;
    t[8] = -D;
    t[9] = C * F;
    t[10] = C * E;
    t[11] = z;
}

int scvod_batch_map_accumulate(scvod_ctx* ctx, scvod_map* m, const float* h_poses, int32_t flags, void* stream) {
    return scvod_batch_map_accumulate_range(ctx, m, h_poses, flags, 0, -1, stream);
}

int scvod_batch_map_accumulate_range(scvod_ctx* ctx, scvod_map* m, const float* h_poses, int32_t flags, int32_t first, int32_t count, void* stream) {
    if (!ctx || !m || !h_poses) return mfail(m, SCVOD_ERR_INVALID, "bad arguments");
    Arena A;
    int device = 0, track_valid = 0, batch_valid = 0, n_scans = 0, max_pts = 0, mode = 0, min_pts = 0;
    scvod__ctx_view(ctx, &A, &device, &track_valid, &batch_valid, &n_scans, &max_pts, &mode, &min_pts);
    if (!batch_valid || !A.pts) return mfail(m, SCVOD_ERR_STATE, "scvod_batch_map_accumulate needs a processed batch of input clouds");
    const int part = (flags & SCVOD_MAP_PART_UNTRACKED) ? 1 : ((flags & SCVOD_MAP_PART_TRACKED) ? 2 : 0);
    if ((flags & SCVOD_MAP_PART_UNTRACKED) && (flags & SCVOD_MAP_PART_TRACKED)) return mfail(m, SCVOD_ERR_INVALID, "the two part flags exclude each other");
    if (part && !scvod__ctx_types_valid(ctx)) return mfail(m, SCVOD_ERR_STATE, "a map part needs scvod_batch_cluster and scvod_batch_cluster_types of the batch");
    const int use_dyn = !(flags & SCVOD_MAP_IGNORE_DYNAMIC) && part != 1;  // (part 1 holds no point a tracking result concerns)
    if (use_dyn && !track_valid) return mfail(m, SCVOD_ERR_STATE, "no tracking result: run scvod_batch_track or pass SCVOD_MAP_IGNORE_DYNAMIC");
    if (device != m->device) return mfail(m, SCVOD_ERR_INVALID, "map and ctx live on different devices");
    MHIP(m, hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    std::vector<float> T((size_t)12 * n_scans);
    for (int s = 0; s < n_scans; ++s) scvod_pose_matrix(h_poses + 6 * s, T.data() + 12 * s);
    if (m->pose_cap < n_scans) {
        MHIP(m, hipStreamSynchronize(st));
        if (m->d_pose) hipFree(m->d_pose);
        m->d_pose = nullptr;
        m->pose_cap = 0;
        m->up_pose.clear();
        MHIP(m, hipMalloc(&m->d_pose, sizeof(float) * 12 * (size_t)n_scans));
        m->pose_cap = n_scans;
    }
    if (m->up_pose != T) {  // the staging vector may still feed an earlier asynchronous copy
        MHIP(m, hipStreamSynchronize(st));
        m->up_pose = T;
        MHIP(m, hipMemcpyAsync(m->d_pose, m->up_pose.data(), sizeof(float) * T.size(), hipMemcpyHostToDevice, st));
    }
    if (count < 0) count = n_scans - first;
    if (first < 0 || count < 0 || first + count > n_scans) return mfail(m, SCVOD_ERR_INVALID, "scan range [%d, %d) outside the batch of %d", first, first + count, n_scans);
    if (max_pts > 0 && count > 0) {
        const int need_lists = (flags & (SCVOD_MAP_NO_GROUND | SCVOD_MAP_NO_REJECTED)) ? 1 : 0;
        if (need_lists) {  // rare: a map without the ground / without the range-FOV rejects needs those two lists marked
            hipLaunchKernelGGL(k_map_mark_lists, dim3((max_pts + 2047) / 2048, n_scans), dim3(256), 0, st, A);
        }
        hipLaunchKernelGGL(k_map_accumulate, dim3((max_pts + kMapPts - 1) / kMapPts, count), dim3(256), 0, st, first, A, m->d_pose, m->table,
                           (unsigned long long)(m->capacity - 1), 1.0f / m->leaf, (flags & SCVOD_MAP_NO_GROUND) ? 0 : 1,
                           (flags & SCVOD_MAP_NO_REJECTED) ? 0 : 1, use_dyn, mode == 1 ? 1 : 0, min_pts, (use_dyn || need_lists || part) ? 1 : 0, part, m->counters);
        MHIP(m, hipGetLastError());
    }
    return SCVOD_OK;
}

static int map_export(scvod_map* m, void* d_records, void* d_xyzi, int64_t cap, int64_t* n_out, void* stream) {
    if (!m || !n_out || cap < 0) return mfail(m, SCVOD_ERR_INVALID, "bad arguments");
    MHIP(m, hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    MHIP(m, hipMemsetAsync(m->counters, 0, 8, st));
    hipLaunchKernelGGL(k_map_export, dim3(256 * 8), dim3(256), 0, st, m->table, m->capacity, (MapRec*)d_records, (long long)cap, (float4*)d_xyzi,
                       m->leaf, m->counters);
    MHIP(m, hipGetLastError());
    unsigned long long h[2] = {0, 0};
    MHIP(m, hipMemcpyAsync(h, m->counters, 16, hipMemcpyDeviceToHost, st));
    MHIP(m, hipStreamSynchronize(st));
    *n_out = (int64_t)h[0];
    if (h[1]) return mfail(m, SCVOD_ERR_CAPACITY, "%llu points did not fit the map (table of %lld cells full or coordinates out of range)", h[1], m->capacity);
    if ((d_records || d_xyzi) && (int64_t)h[0] > cap) return mfail(m, SCVOD_ERR_CAPACITY, "output buffer too small (%lld < %llu cells)", (long long)cap, h[0]);
    return SCVOD_OK;
}

int scvod_map_export(scvod_map* m, void* d_records, int64_t cap_records, int64_t* n_out, void* stream) {
    return map_export(m, d_records, nullptr, cap_records, n_out, stream);
}
int scvod_map_points(scvod_map* m, void* d_xyzi, void* d_keys_records, int64_t cap, int64_t* n_out, void* stream) {
    return map_export(m, d_keys_records, d_xyzi, cap, n_out, stream);
}

int scvod_map_export_parts(scvod_map* m, int32_t n_parts, void* d_records, int64_t cap_records, int64_t* h_counts, void* stream) {
    if (!m || n_parts < 1 || n_parts > kMapMaxParts || !d_records || !h_counts || cap_records < 0) return mfail(m, SCVOD_ERR_INVALID, "bad arguments");
    MHIP(m, hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* pc = m->counters + 2;
    MHIP(m, hipMemsetAsync(pc, 0, 8 * kMapMaxParts, st));
    hipLaunchKernelGGL(k_map_count_parts, dim3(256 * 8), dim3(256), 0, st, m->table, m->capacity, (int)n_parts, pc);
    unsigned long long h[2 + kMapMaxParts];
    MHIP(m, hipMemcpyAsync(h, m->counters, 8 * (2 + kMapMaxParts), hipMemcpyDeviceToHost, st));
    MHIP(m, hipStreamSynchronize(st));
    if (h[1]) return mfail(m, SCVOD_ERR_CAPACITY, "%llu points did not fit the map (table of %lld cells full or coordinates out of range)", h[1], m->capacity);
    unsigned long long cur[kMapMaxParts], run = 0;
    for (int p = 0; p < kMapMaxParts; ++p) {
        cur[p] = run;
        if (p < n_parts) {
            h_counts[p] = (int64_t)h[2 + p];
            run += h[2 + p];
        }
    }
    if ((int64_t)run > cap_records) return mfail(m, SCVOD_ERR_CAPACITY, "output buffer too small (%lld < %llu cells)", (long long)cap_records, run);
    MHIP(m, hipMemcpyAsync(pc, cur, 8 * kMapMaxParts, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_map_scatter_parts, dim3(256 * 8), dim3(256), 0, st, m->table, m->capacity, (int)n_parts, pc, (MapRec*)d_records,
                       (long long)cap_records);
    MHIP(m, hipGetLastError());
    MHIP(m, hipStreamSynchronize(st));  // `cur` lives on this stack frame
    return SCVOD_OK;
}

int scvod_map_export_parts_padded(scvod_map* m, int32_t n_parts, void* d_records, int64_t cap_per_part, void* d_counts, void* stream) {
    if (!m || n_parts < 1 || n_parts > kMapMaxParts || !d_records || cap_per_part < 1) return mfail(m, SCVOD_ERR_INVALID, "bad arguments");
    MHIP(m, hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* pc = m->counters + 2;
    MHIP(m, hipMemsetAsync(pc, 0, 8 * kMapMaxParts, st));
    MHIP(m, hipMemsetAsync(d_records, 0xff, (size_t)16 * (size_t)n_parts * (size_t)cap_per_part, st));  // padding: key ~0
    hipLaunchKernelGGL(k_map_scatter_parts_padded, dim3(256 * 8), dim3(256), 0, st, m->table, m->capacity, (int)n_parts, pc, (MapRec*)d_records,
                       (long long)cap_per_part, m->counters);
    MHIP(m, hipGetLastError());
    if (d_counts) MHIP(m, hipMemcpyAsync(d_counts, pc, 8 * (size_t)n_parts, hipMemcpyDeviceToDevice, st));
    return SCVOD_OK;
}

int scvod_map_merge(scvod_map* m, const void* d_records, int64_t n, void* stream) {
    if (!m || n < 0 || (n > 0 && !d_records)) return mfail(m, SCVOD_ERR_INVALID, "bad arguments");
    if (n == 0) return SCVOD_OK;
    MHIP(m, hipSetDevice(m->device));
    hipLaunchKernelGGL(k_map_merge, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const MapRec*)d_records, (long long)n, m->table,
                       (unsigned long long)(m->capacity - 1), m->counters);
    MHIP(m, hipGetLastError());
    return SCVOD_OK;
}

}  // extern "C"
