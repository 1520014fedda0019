// scvod_track.hip -- scan-vs-next-scan differencing of a whole sequence shard on the device (gfx950, wave64).
//
// Reference: SSC::tracking, /root/reference/src/ssc.cpp:1250-1426, as SSC::segDF drives it (ssc.cpp:1449-1451: every
// frame against its successor).  For every `car` cluster of scan s (type from the bounding-box rules, ssc.cpp:849-872):
//   transform its points into the successor's frame (utility.h:394-406), re-bin them WITHOUT range/FOV rejection or
//   clamping (ssc.cpp:1280-1286), look the voxel up in the successor's hash_cloud and keep it when its label != -1
//   (ssc.cpp:1304-1305); group the hits by that label, sampleVec each group (ssc.cpp:1319-1321) -> remap_name;
//   state = 1 (dynamic) when remap_name is empty (ssc.cpp:1323-1326) or when it has ONE label whose cluster is a car and
//   (float)|hits| / (float)|occupy_voxels| < occupancy (ssc.cpp:1336-1349); state = 0 (static) for one non-car label below
//   the ratio (ssc.cpp:1351-1352), one car label at or above it (ssc.cpp:1378-1380) and for several labels (ssc.cpp:1396-1397);
//   one non-car label at or above the ratio leaves the state untouched (-1).
// The labels / cluster sizes / types of the successor are those of its fresh segmentation (clusterAndCreateFrame +
// refineClusterByBoundingBox + the box rules of recognize), i.e. the state SSC::tracking finds before an earlier pair has
// re-labelled anything: the FIRST-ORDER decision, all pairs in parallel.  The sequential chain of SSC::segDF (ssc.cpp:1449-1451:
// a pair appends clouds to / splits / fuses the successor's clusters before the next pair walks them, ssc.cpp:1351-1419) is
// replayed on top of these results by scvod_chain.hip (launch_track_chain below, between the decision and the per-point bytes);
// what this file leaves for it per car cluster: the sorted unique hit list, remap_name with the label ids, the state.
//
// Everything stays in HBM: cluster names and types come from scvod_batch_cluster / scvod_batch_cluster_types, the member
// lists of the car clusters are built here (counting sort by cluster root), no host round trip inside a call.
#include "scvod_chain.h"

namespace scvod {

constexpr int kTkSamples = 8192;     // sampled successor keys staged in LDS (32 KB)

// lanes of a wave that hold the same key (neighbouring apri points / voxels mostly do): leader lane, rank inside the group
// and group size, so that ONE lane per distinct key issues the atomic.  key < 0 = idle lane.
__device__ __forceinline__ void wave_group(int key, int& leader, int& rank, int& count) {
    const int lane = threadIdx.x & 63;
    bool todo = key >= 0;
    leader = lane;
    rank = 0;
    count = 0;
    while (__any(todo)) {
        const int first = __ffsll((long long)__ballot(todo)) - 1;
        const int k0 = __shfl(key, first);
        const bool mine = todo && (key == k0);
        const unsigned long long mask = __ballot(mine);
        if (mine) {
            leader = first;
            rank = __popcll(mask & ((1ull << lane) - 1ull));
            count = __popcll(mask);
            todo = false;
        }
    }
}

struct NextTable {
    const int4* tab;  // records {key, label, cluster voxels, cluster type}, ascending key
    const int32_t* rep;  // per record: lowest slot carrying the same label (tables of the batch only; nullptr for a boundary table)
    int nv;
};
__device__ __forceinline__ NextTable next_table_of(const Arena& A, const TrackBatch& J, int s) {
    NextTable t;
    t.tab = nullptr;
    t.rep = nullptr;
    t.nv = -1;  // no successor
    const int nxt = J.next_scan[s];
    if (nxt >= 0) {
        t.tab = A.vox_track + A.scan_off[nxt];
        t.rep = A.vox_rep + A.scan_off[nxt];
        t.nv = A.counts[nxt * 8 + 6];
    } else if (nxt <= -2 && (-2 - nxt) < J.n_ext) {
        const int4* e = J.ext_tables[-2 - nxt];
        t.nv = e[0].x;
        t.tab = e + 1;
    }
    return t;
}

// per cluster root: reset the per-cluster words the decision pass accumulates into (member lists, pairs, states: a second
// scvod_batch_track on the same tables must start from clean words too)
__global__ __launch_bounds__(256) void k_tk_init(Arena A, int phase) {
    const int s = blockIdx.y;
    const int base = A.scan_off[s];
    const int n = A.counts[s * 8 + 4];
    if (blockIdx.x == 0 && threadIdx.x < 2) A.tk_scan[s * 4 + 2 + threadIdx.x] = 0;  // ([0], [1]: the car lists, written by k_cc_scan)
    // only the car clusters are walked, decided and read back (k_tk_decide, k_tk_dyn, scvod_batch_fetch_track): their roots
    const int ncar = A.tk_scan[s * 4 + 0];
    for (int o = blockIdx.x * 256 + threadIdx.x; o < ncar; o += gridDim.x * 256) {
        const int i = A.tk_clusters[(size_t)base + o];
        A.cl_state[(size_t)base + i] = -1;
        A.tk_npairs[(size_t)base + i] = 0;
        A.tk_nuniq[(size_t)base + i] = 0;
    }
}

// Transform + unfiltered re-bin + look-up of every car point of scan s in its successor's table.  The table's keys are
// staged in LDS sampled every 2^shift entries (<= 8192 samples); the last <= 2^shift candidates are one or two cache
// lines of the table itself.
__global__ __launch_bounds__(256) void k_tk_probe(DevParams P, Arena A, TrackBatch J, int from_apri) {
    __shared__ int32_t skeys[kTkSamples];
    const int s = blockIdx.y;
    const int n_car = A.tk_scan[s * 4 + 1];
    if ((int)blockIdx.x * 1024 >= n_car) return;
    const NextTable N = next_table_of(A, J, s);
    if (N.nv < 0) return;
    const int base = A.scan_off[s];
    int shift = 0;
    while (((N.nv + (1 << shift) - 1) >> shift) > kTkSamples) ++shift;
    const int ns = (N.nv + (1 << shift) - 1) >> shift;
    for (int j = threadIdx.x; j < ns; j += 256) skeys[j] = N.tab[(size_t)j << shift].x;
    __syncthreads();
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = J.T[12 * s + i];
    constexpr int U = 4;  // four points per thread and step: the dependent gathers and the searches of all four overlap
    for (int k0 = blockIdx.x * (256 * U); k0 < n_car; k0 += gridDim.x * (256 * U)) {
        int iv[U], key[U], lo[U], hi[U];
        float4 q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) iv[u] = A.tk_members[(size_t)base + min(k0 + u * 256 + (int)threadIdx.x, n_car - 1)];
        if (!from_apri) {  // the point itself is read through apri_src (cloud_use[i] = input point apri_src[i])
#pragma unroll
            for (int u = 0; u < U; ++u) iv[u] = A.apri_src[(size_t)base + iv[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) q[u] = A.pts[base + iv[u]];
        } else {  // apri_vec supplied by the caller (no input cloud on the device)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const scvod_apri& a = A.apri[(size_t)base + iv[u]];
                q[u] = make_float4(a.x, a.y, a.z, a.intensity);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // Utility::transformCloud (utility.h:401-404): explicit fp32 dot products, no FMA
            const float x = T[0] * q[u].x + T[1] * q[u].y + T[2] * q[u].z + T[3];
            const float y = T[4] * q[u].x + T[5] * q[u].y + T[6] * q[u].z + T[7];
            const float z = T[8] * q[u].x + T[9] * q[u].y + T[10] * q[u].z + T[11];
            int32_t vi;  // no range/FOV rejection, no clamping (ssc.cpp:1280-1286); the reference arithmetic next to a bin edge only
            if (!voxel_idx_fast(P.bin, P.binfast, x, y, z, &vi)) {
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
        for (int u = 0; u < U; ++u) first[u] = N.tab[min(max(lo[u] - 1, 0) << shift, max(N.nv - 1, 0))];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 256 + (int)threadIdx.x;
            if (k >= n_car) continue;
            int slot = -1;
            if (lo[u] > 0) {
                const int a0 = (lo[u] - 1) << shift;
                slot = tk_find_slot(N.tab, a0, min(a0 + (1 << shift), N.nv), first[u], key[u]);
            }
            A.tk_hit[(size_t)base + k] = slot;
        }
    }
}

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// One WAVE per car cluster (persistent over the clusters of its scan), no workgroup barriers: a private bitset over the
// successor's table in LDS turns the cluster's hits into the sorted unique list (sampleVec without a sort), the list is
// grouped by label -> remap_name, then the state rule.  `words` = LDS words per wave (covers the largest table).
template <int kTkWaves>
__global__ __launch_bounds__(64 * kTkWaves) void k_tk_decide(Arena A, TrackBatch J, int words, int min_words) {
    extern __shared__ uint32_t tk_bits[];
    const int s = blockIdx.y;
    const int ncl = A.tk_scan[s * 4 + 0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int first = blockIdx.x * kTkWaves + wave, stride = gridDim.x * kTkWaves;
    if (first >= ncl) return;
    const NextTable N = next_table_of(A, J, s);
    if (N.nv < 0) return;  // last scans of a sequence: no tracking call, states stay -1
    const int base = A.scan_off[s];
    uint32_t* bits = tk_bits + (size_t)wave * words;
    // two launches share the scans: tables of up to `words` bitset words here, larger ones in the launch with more LDS per wave
    if (((N.nv + 31) >> 5) > words || ((N.nv + 31) >> 5) <= min_words) return;
    const int nw = (N.nv + 31) >> 5;
    for (int w = lane; w < nw; w += 64) bits[w] = 0u;
    __builtin_amdgcn_wave_barrier();
    for (int ord = first; ord < ncl; ord += stride) {
        const int root = A.tk_clusters[(size_t)base + ord];
        const int k0 = A.tk_mbegin[(size_t)base + root];
        const int m = A.cl_count[(size_t)base + root];
        int wlo = 0x7fffffff, whi = -1;
        for (int j = lane; j < m; j += 64) {
            const int slot = A.tk_hit[(size_t)base + k0 + j];
            if (slot >= 0 && (slot >> 5) < nw) {
                atomicOr(&bits[slot >> 5], 1u << (slot & 31));
                wlo = min(wlo, slot >> 5);
                whi = max(whi, slot >> 5);
            }
        }
        wlo = wave_min_i(wlo);
        whi = wave_max_i(whi);
        // sorted unique slots of the cluster, only over the words it touched; the words are cleared on the way
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
            int o = k0 + U + inc - c;
            while (word) {
                const int b = __ffs(word) - 1;
                word &= word - 1;
                A.tk_uniq[(size_t)base + o++] = (w << 5) + b;
            }
            U += __shfl(inc, 63);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // remap_name: labels in ascending order with the number of unique voxels each
        int npairs = 0, one_count = 0, one_nvox = 1, one_type = 0;
        int cur = -1;
        for (;;) {
            int mn = 0x7fffffff;
            for (int j = lane; j < U; j += 64) {
                const int lab = N.tab[A.tk_uniq[(size_t)base + k0 + j]].y;
                if (lab > cur) mn = min(mn, lab);
            }
            mn = wave_min_i(mn);
            if (mn == 0x7fffffff) break;
            int cnt = 0, nvx = 0, typ = 0, rep = -1;
            for (int j = lane; j < U; j += 64) {
                const int slot = A.tk_uniq[(size_t)base + k0 + j];
                const int4 rec = N.tab[slot];
                if (rec.y == mn) {
                    ++cnt;
                    nvx = rec.z;
                    typ = rec.w;
                    if (N.rep) rep = N.rep[slot];
                }
            }
            nvx = wave_max_i(nvx);
            typ = wave_max_i(typ);
            rep = wave_max_i(rep);
            cnt = wave_sum_i(cnt);
            if (lane == 0) {
                A.tk_pairs[(size_t)base + k0 + npairs] = make_int2(mn, cnt);
                A.tk_prep[(size_t)base + k0 + npairs] = rep;
            }
            if (npairs == 0) {
                one_count = cnt;
                one_nvox = nvx;
                one_type = typ;
            }
            ++npairs;
            cur = mn;
        }
        if (lane == 0) {
            int state;
            if (npairs == 0) {
                state = 1;  // nothing of the successor under the transformed cluster (ssc.cpp:1323-1326)
            } else if (npairs == 1) {
                const float ratio = (float)one_count / (float)one_nvox;  // ssc.cpp:1336
                if (ratio < J.occupancy)
                    state = (one_type == 2) ? 1 : 0;  // ssc.cpp:1337-1352
                else
                    state = (one_type == 2) ? 0 : -1;  // ssc.cpp:1377-1380; a non-car successor leaves it untouched
            } else {
                state = 0;  // ssc.cpp:1396-1397
            }
            A.cl_state[(size_t)base + root] = (int8_t)state;
            A.tk_nuniq[(size_t)base + root] = U;
            A.tk_npairs[(size_t)base + root] = npairs;
            if (state == 1) {
                atomicAdd(&A.tk_scan[s * 4 + 2], 1);
                atomicAdd(&A.tk_scan[s * 4 + 3], m);
            }
        }
    }
}

// per apri point: dynamic when its cluster was decided dynamic; points of clusters the bounding-box refine erased belong
// to no cluster (the reference lists them as static, ssc.cpp:450-454)
__global__ __launch_bounds__(256) void k_tk_dyn(Arena A, int from_apri) {
    const int s = blockIdx.y;
    const int base = A.scan_off[s];
    const int n = A.counts[s * 8 + 4];
    // four points per thread in flight: the chain type -> cluster -> state is three dependent loads deep
    constexpr int U = 4;
    for (int i0 = blockIdx.x * (256 * U) + threadIdx.x; i0 < n; i0 += gridDim.x * (256 * U)) {
        int t[U], c[U];
        int8_t st[U];
#pragma unroll
        for (int u = 0; u < U; ++u) t[u] = A.pt_type[(size_t)base + min(i0 + u * 256, n - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = (t[u] == 2) ? A.pt_cluster[(size_t)base + min(i0 + u * 256, n - 1)] : 0;
#pragma unroll
        for (int u = 0; u < U; ++u) st[u] = (t[u] == 2) ? A.cl_state[(size_t)base + c[u]] : (int8_t)0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256;
            if (i >= n) break;
            uint8_t d = SCVOD_DYN_STATIC;
            if (t[u] == 0)
                d = SCVOD_DYN_UNCLUSTERED;
            else if (t[u] == 2 && st[u] == 1)
                d = SCVOD_DYN_DYNAMIC;
            A.pt_dyn[(size_t)base + i] = d;
            // the static map's mark of a car point: plain store, so that a second tracking run over the same clustering starts clean
            if (!from_apri && t[u] == 2) A.pt_mapcls[(size_t)base + A.apri_src[(size_t)base + i]] = (uint8_t)(kMapCar | (d == SCVOD_DYN_DYNAMIC ? kMapDynamic : 0));
        }
    }
}

// boundary message of a sequence shard: header {n_voxels, 0, 0, 0} + the vox_track records of scan s
__global__ __launch_bounds__(256) void k_tk_export(Arena A, int s, int4* out, long long cap_records) {
    const int base = A.scan_off[s];
    const int nv = A.counts[s * 8 + 6];
    const long long fit = cap_records - 1 < (long long)nv ? cap_records - 1 : (long long)nv;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = make_int4((int)fit, nv, 0, 0);
    for (long long v = blockIdx.x * 256 + threadIdx.x; v < fit; v += (long long)gridDim.x * 256) out[1 + v] = A.vox_track[(size_t)base + v];
}

#define TH_BEGIN(name) \
    if (th) th(tu, name, 1)
#define TH_END(name) \
    if (th) th(tu, name, 0)

// phases: 1 = successor tables only (labels, cluster sizes: what scvod_batch_export_table hands to another shard),
// 2 = member lists, probe, decision, per-point bytes (needs phase 1 of the same clustering), 3 = both
void launch_track_batch(const DevParams& P, const Arena& A, const TrackBatch& J, int from_apri, int phases, hipStream_t st,
                        TimerHook th, void* tu, const ChainJob* chain, hipEvent_t before_chain) {
    const int B = A.n_scans;
    if (B <= 0 || A.max_scan_pts <= 0) return;
    const dim3 g((A.max_scan_pts + 2047) / 2048, B);
    if (!(phases & 2)) return;  // (phase 1, the successor tables, is written by k_cc_scan with the clustering itself)
    TH_BEGIN("tk_init");  // (the member lists of the car clusters are written by k_cc_scan with the clustering itself)
    hipLaunchKernelGGL(k_tk_init, dim3(1, B), dim3(256), 0, st, A, 2);
    TH_END("tk_init");
    TH_BEGIN("tk_probe");
    hipLaunchKernelGGL(k_tk_probe, dim3(8, B), dim3(256), 0, st, P, A, J, from_apri);  // (a block stages up to 32 KB of keys: few, long-lived blocks)
    TH_END("tk_probe");
    // LDS words of a wave's bitset: a table holds at most max_scan_pts voxels
    // a wave's bitset covers the successor's voxel table: tables of up to 32 768 voxels (any street scan) take the launch
    // with 4 KB of LDS per wave (eight workgroups per CU), larger ones the launch sized for the largest scan
    const int small_words = 1024;
    const int words = (A.max_scan_pts + 31) / 32 + 1;  // SCVOD_MAX_SCAN_POINTS = 2^19 slots = 64 KB: one wave always fits
    const size_t per_wave = (size_t)words * 4;
    TH_BEGIN("tk_decide");
    hipLaunchKernelGGL(k_tk_decide<4>, dim3(8, B), dim3(256), 4 * small_words * 4, st, A, J, small_words, -1);
    // (a 128-beam scan on a fine grid: 70 k voxels in a batch sized for 260 k points) tables of up to 81 920 voxels: 10 KB per wave,
    // four workgroups per CU; up to 131 072 voxels: 16 KB per wave, two workgroups per CU
    int done_words = small_words;
    for (const int mid_words : {2560, 4096}) {
        if (words <= mid_words) break;
        hipFuncSetAttribute((const void*)k_tk_decide<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * mid_words * 4);
        hipLaunchKernelGGL(k_tk_decide<4>, dim3(8, B), dim3(256), 4 * mid_words * 4, st, A, J, mid_words, done_words);
        done_words = mid_words;
    }
    if (words > small_words) {
        const int small_words = done_words;  // (the launches below take the tables beyond what is done)
        if (4 * per_wave <= 64 * 1024) {  // two workgroups per CU
            hipFuncSetAttribute((const void*)k_tk_decide<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * per_wave));
            hipLaunchKernelGGL(k_tk_decide<4>, dim3(8, B), dim3(256), 4 * per_wave, st, A, J, words, small_words);
        } else if (2 * per_wave <= 150 * 1024) {
            hipFuncSetAttribute((const void*)k_tk_decide<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * per_wave));
            hipLaunchKernelGGL(k_tk_decide<2>, dim3(16, B), dim3(128), 2 * per_wave, st, A, J, words, small_words);
        } else {
            hipFuncSetAttribute((const void*)k_tk_decide<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)per_wave);
            hipLaunchKernelGGL(k_tk_decide<1>, dim3(32, B), dim3(64), per_wave, st, A, J, words, small_words);
        }
    }
    TH_END("tk_decide");
    // the reference's sequential chain on top of the first-order results (scvod_chain.hip): states and dynamic counters of
    // every frame that has a successor in the batch become those of SSC::segDF's loop
    if (chain) {
        if (before_chain) hipStreamWaitEvent(st, before_chain, 0);  // (Frame::max_name of every scan: scvod_lastname.hip, on its own stream)
        launch_track_chain(P, A, J, *chain, from_apri, st, th, tu);
    }
    TH_BEGIN("tk_dyn");
    hipLaunchKernelGGL(k_tk_dyn, g, dim3(256), 0, st, A, from_apri);
    TH_END("tk_dyn");
}

void launch_track_dyn(const Arena& A, int from_apri, hipStream_t st) {
    if (A.n_scans <= 0 || A.max_scan_pts <= 0) return;
    const dim3 g((A.max_scan_pts + 2047) / 2048, A.n_scans);
    hipLaunchKernelGGL(k_tk_dyn, g, dim3(256), 0, st, A, from_apri);
}

void launch_export_table(const Arena& A, int s, int4* out, long long cap_records, hipStream_t st) {
    hipLaunchKernelGGL(k_tk_export, dim3(64), dim3(256), 0, st, A, s, out, cap_records);
}

}  // namespace scvod
