// scvod_lastname.hip -- which cluster of a scan still carries Frame::max_name? (gfx950, wave64)
//
// Reference: SSC::clusterAndCreateFrame ends with `frame_ssc.max_name = cluster_name ++;` (/root/reference/src/ssc.cpp:354):
// max_name is the LAST USED running number K, not the next free one.  SSC::tracking hands it out again
// (`cluster_new.name = frame_next_.max_name ++`, ssc.cpp:1357 and :1401): the first cluster a call splits off or fuses in
// frame i + 1 is called K, and when a cluster K is still alive `cluster_set.insert` (ssc.cpp:1372, :1419) is a no-op -- the
// voxels were already re-labelled K, the new cluster is lost.  The chain (scvod_chain.hip) reproduces that, and for it the
// device has to know WHICH of its clusters (canonical names = smallest point index) the reference would call K.
//
// K is the number the last "opener" of the visiting loop created (ssc.cpp:303-350: a point that carries no label after it
// looked at its neighbours opens a new cluster), and it survives only if no later mergeClusters(oc, nc) call (ssc.cpp:329,
// :413-419: the VISITING point's cluster takes the NEIGHBOUR's name) renamed it.  Both depend on the visiting order, so the
// loop has to be followed -- but only where it matters:
//   * the loop never looks across a connected component of "lists" (every listed voxel ends up in the lister's component; index
//     triples outside the grid tie the clusters they touch into one such class), so each class can be followed on its own, at
//     voxel level: a voxel is unlabelled, labelled through its visited points, or FULLY labelled; only the first three visits
//     from a voxel change anything (the state machine of k_cc_scan's exact path), irregular points visit
//     one by one with their own lists;
//   * the last opener is at least as late as the birth (smallest point) L of the latest-born class.  Of all other classes only a
//     voxel whose first point comes after L can still hold a later opener, and such a voxel is none when a voxel it lists holds a
//     visited point, or CERTAINLY carries a label by then (a lister's third visit labels its whole list, the second everything
//     from its own cell on, any visit everything behind a voxel that holds a visited point): a thread-parallel pass over the
//     voxels and a wave-wide certificate over the 5 x 5 x 5 cells around the few candidates that are left.  The latest-born class
//     and the classes of the candidates no certificate settles are followed -- together, they never list each other.  When
//     refineClusterByBoundingBox erased every cluster among them, no live cluster carries K and nothing is followed;
//   * following a set of classes: (1) the times at which voxels become fully labelled, by Jacobi rounds over the visits (a
//     time depends on earlier times only: the fixed point is the sequential loop's, reached in 2-10 rounds); (2) the openers =
//     visits that start without a label and find none, the last one by a max; (3) the partition right after it = unions of
//     what the last visit of every voxel up to then joined; (4) the few visits after it walked in order by one wave (lane p
//     looks at the p-th listed voxel; the walk only steps through the DISTINCT classes a visit meets, the neighbour's name
//     winning each time) -- K's class either keeps the name or is renamed away;
//   * the tables of a set live in LDS: 320 nodes for nearly every scan (eight workgroups per CU), 1792 or 8192 for the scans
//     listed on the device for the larger passes; a set of up to 32 767 nodes (the facades of a 128-beam scan: 15-30 k voxels)
//     is followed by a fourth pass whose read-mostly tables (times, first points, flags) live in arena scratch and whose
//     atomics table (the next round's times, later the partition's parents) takes the LDS: 4 ms for a 28 k-node set against
//     0.8 ms for 5 k nodes in LDS.  Beyond that (node numbers are 16-bit in the lists) the scan reports "unknown" (counted,
//     scvod_batch_cluster_last_name) and the chain hands out a fresh number as if K had been merged away.
// Checked against a literal restatement of the loop in tests/test_gpu_lastname.py.
#include "scvod_dev.h"

namespace scvod {

namespace {

constexpr int kLnMaxWaves = 16;
constexpr int kLnIrr = 256;       // index triples outside the grid handled per scan (a 128-beam scan holds 36)
constexpr int kLnPairs = 1024;    // links between clusters that such points create
constexpr int kLnNames = 256;     // distinct cluster names in those links
constexpr int kLnSamples = 1024;  // sampled voxel keys (LDS)
constexpr int kLnMarked = 16;     // components to replay besides the latest-born one
constexpr int kLnCapTinyNodes = 320, kLnCapSmallNodes = 1792, kLnCapBigNodes = 8192;  // nodes the LDS tables of the three passes hold
constexpr int kLnCapHugeNodes = 32767;  // a fourth pass with its tables in arena scratch (round 5): the facade components of a 128-beam scan (~20 k voxels);
                                        // node numbers are 16-bit in the lists (Rp::rows)
constexpr int kLnChunk = 512;     // events staged per round
constexpr int kInf = 0x7fffffff;

enum : uint8_t { F_FULL = 1, F_REGVIS = 2 };

struct Ln {  // per-scan view
    const int32_t* vkey;
    const int32_t* vbeg;
    const int32_t* vpts;
    const int32_t* ptc;
    const int32_t* idx3;
    const int32_t* akey;
    int n, nv, R, S, Az;
    long long span;
    bool irregular;
    // scratch (arena arrays that are dead between the clustering and the tracking)
    int32_t* vcl;   // [nv] closure class of the voxel's first point
    int32_t* loc;   // [nv] voxel -> node of the current replay, -1
    int32_t* evn;   // [n] event point -> node
    uint32_t* bits; // [n / 32 + 1] event points
    int32_t* evl;   // [n] events in time order
    int32_t* cvl;   // [nv] node -> voxel
    int32_t* fp;    // [nv] first point of the voxel
    long long rows_bytes;  // room for the lists
    int n_raw;      // input points of the scan (size of the per-scan scratch regions)
    // LDS
    int32_t* skey;
    int sshift, ns;
    const int32_t* lkeys;  // all voxel keys in LDS while a class's lists are built (nullptr: sampled keys + the table in HBM)
    int32_t* irr_pt;    // [kLnIrr] sorted by point
    int32_t* irr_home;  // voxel slot
    int32_t* irr_cls;   // closure class
    int32_t* irr_reg0;  // first regular point of the home voxel
    int32_t* irr_xs;    // the voxels that hold such points: sorted, unique
    int n_irr, n_xs;
    int32_t* cname;  // class table: sorted names
    int32_t* crep;
    int n_names;
};

__device__ __forceinline__ void decode3(int32_t t, int& r, int& s, int& a) {
    r = (t & 2047) - 2;
    s = ((t >> 11) & 2047) - 2;
    a = ((t >> 22) & 1023) - 2;
}
__device__ __forceinline__ bool in_grid(const Ln& L, int r, int s, int a) { return r >= 0 && r < L.R && s >= 0 && s < L.S && a >= 0 && a < L.Az; }
__device__ __forceinline__ bool regular_point(const Ln& L, int i) {
    int r, s, a;
    decode3(L.idx3[i], r, s, a);
    return in_grid(L, r, s, a) && L.akey[i] == a * L.R * L.S + r * L.S + s;
}
// voxel slot of a key, -1 when no voxel carries it
__device__ __forceinline__ int slot_of_key(const Ln& L, int key) {
    int lo = 0, hi = L.ns;  // first sample > key
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (L.skey[mid] <= key)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo == 0) return -1;
    int a = (lo - 1) << L.sshift, b = min(a + (1 << L.sshift), L.nv);
    while (a < b) {
        const int mid = (a + b) >> 1;
        if (L.vkey[mid] < key)
            a = mid + 1;
        else
            b = mid;
    }
    return (a < L.nv && L.vkey[a] == key) ? a : -1;
}
// first voxel slot whose key is >= key
__device__ __forceinline__ int lower_slot(const Ln& L, int key) {
    if (L.lkeys) {
        int a = 0, b = L.nv;
        while (a < b) {
            const int mid = (a + b) >> 1;
            if (L.lkeys[mid] < key)
                a = mid + 1;
            else
                b = mid;
        }
        return a;
    }
    int lo = 0, hi = L.ns;  // first sample >= key
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (L.skey[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    int a = lo == 0 ? 0 : ((lo - 1) << L.sshift) + 1, b = lo >= L.ns ? L.nv : min(lo << L.sshift, L.nv);
    while (a < b) {
        const int mid = (a + b) >> 1;
        if (L.vkey[mid] < key)
            a = mid + 1;
        else
            b = mid;
    }
    return a;
}
__device__ __forceinline__ int slot_of_cell(const Ln& L, int r, int s, int a) {
    if (!in_grid(L, r, s, a)) return -1;
    return slot_of_key(L, a * L.R * L.S + r * L.S + s);
}
__device__ __forceinline__ bool cell_of_voxel(const Ln& L, int v, int& r, int& s, int& a) {
    const int key = L.vkey[v];
    if (key < 0 || (long long)key >= L.span) return false;
    const int RS = L.R * L.S;
    a = key / RS;
    const int rem = key - a * RS;
    r = rem / L.S;
    s = rem - r * L.S;
    return true;
}
__device__ __forceinline__ bool holds_irregular(const Ln& L, int v) {
    int lo = 0, hi = L.n_xs;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (L.irr_xs[mid] < v)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo < L.n_xs && L.irr_xs[lo] == v;
}
// e-th REGULAR point of voxel v (e = 0, 1, 2), kInf when it has fewer
__device__ __forceinline__ int reg_of(const Ln& L, int v, int e) {
    const int b = L.vbeg[v], c = L.vbeg[v + 1] - b;
    if (!L.irregular || !holds_irregular(L, v)) return e < c ? L.vpts[b + e] : kInf;
    int seen = 0;
    for (int j = 0; j < c; ++j) {
        const int i = L.vpts[b + j];
        if (regular_point(L, i) && seen++ == e) return i;
    }
    return kInf;
}
__device__ __forceinline__ int cls_of(const Ln& L, int name) {
    int lo = 0, hi = L.n_names;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (L.cname[mid] < name)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (lo < L.n_names && L.cname[lo] == name) ? L.crep[lo] : name;
}

template <int CAP>
struct Rp {  // tables of one class (LDS unless said otherwise; the CAP == kLnCapHugeNodes instantiation keeps T / fa / lab / fl in arena scratch --
             // a compile-time distinction: a run-time flag in these inner loops cost the LDS passes 1.5-1.8 x)
    static constexpr bool kHbm = (CAP == kLnCapHugeNodes);
    int32_t* T;      // [CAP] voxel node: time of the visit that labelled it FULLY, kInf never
    int32_t* fa;     // [CAP] voxel node: its first point (of any kind), kInf for the nodes of irregular points
    int32_t* Tn;     // [CAP] next round's times
    int32_t* par;    // [CAP] classes at / after the last opener
    uint8_t* fl;     // [CAP]
    int16_t* inext;  // [kLnIrr] next irregular point of the same voxel
    int16_t* inode;  // [kLnIrr] irr list entry -> node of this class, -1
    uint8_t* ivis;   // [kLnIrr]
    int32_t* ev_t;   // [kLnChunk]
    int32_t* ev_x;   // [kLnChunk]
    // arena scratch
    int16_t* rows;   // [nodes][32] listed nodes in findVoxelNeighbors order (ssc.cpp:400-410: range outermost, azimuth innermost), -1 none
    int32_t* ev3;    // [nodes][3] times of a node's visits
    int32_t* ifirst; // [nodes] voxel node -> its first irregular point of this class (index into the scan's irr list), -1
    int32_t* lab;    // [cap] or nullptr: min(T, fa) per node, kept beside the two when the tables live in HBM (one look-up per listed voxel instead of two)
    int cap;         // nodes the tables hold (CAP, or what the scratch of this scan holds when the tables live there)
    bool in_hbm;     // T / fa / Tn / par / fl live in arena scratch, not in LDS
};

struct RpOut {
    int last_open;   // time of the class's last opener, -1 none
    int fin;         // the cluster that point ends in still carries the number it created
    int canon;       // smallest point of that cluster
    int slot;        // lowest voxel slot whose first point belongs to it
    int n_events;    // events walked one by one (after the last opener)
    int too_big;
    int nodes, rounds, c_build, c_jacobi, c_open, c_cc, c_walk;  // development: size, Jacobi rounds, phase clocks (10 ns)
};

template <int CAP>
__device__ __forceinline__ int rp_find(const Rp<CAP>& T, int x) {
    while (T.par[x] != x) {
        const int p = T.par[x];
        T.par[x] = T.par[p];
        x = T.par[x];
    }
    return x;
}
template <int CAP>
__device__ __forceinline__ int rp_find_ro(const Rp<CAP>& T, int x) {
    int p;
    while ((p = __hip_atomic_load(&T.par[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != x) x = p;
    return x;
}
// concurrent union-find for the partition at the time of the last opener (names do not matter there): a root is hooked under
// a SMALLER root, so parents only decrease and path halving by plain stores is safe
template <int CAP>
__device__ __forceinline__ int rp_find_c(const Rp<CAP>& T, int x) {
    for (;;) {
        const int p = __hip_atomic_load(&T.par[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (p == x) return x;
        const int g = __hip_atomic_load(&T.par[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (g != p) __hip_atomic_store(&T.par[x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        x = p;
    }
}
template <int CAP>
__device__ __forceinline__ int rp_union(const Rp<CAP>& T, int a, int b) {  // returns the common root
    for (;;) {
        a = rp_find_c(T, a);
        b = rp_find_c(T, b);
        if (a == b) return a;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        if (atomicCAS(&T.par[a], a, b) == a) return b;
    }
}
// the visiting point (class oc, -1 = no label yet) meets the points of voxel k in index order (ssc.cpp:322-340); kroot: the
// class that carries the last number handed out, -1 once a mergeClusters call renamed it
template <int CAP>
__device__ __forceinline__ void rp_meet(const Ln& L, const Rp<CAP>& T, int& oc, int k, int& kroot) {
    auto take = [&](int c) {
        if (oc < 0) {
            oc = c;
        } else if (c != oc) {  // mergeClusters(oc, nc), ssc.cpp:329: the VISITOR's cluster takes the neighbour's name
            if (oc == kroot) kroot = -1;
            T.par[oc] = c;
            oc = c;
        }
    };
    if (T.fl[k] & F_FULL) {
        take(rp_find(T, k));
        return;
    }
    // labelled so far: the visited regular points (one class: the voxel node's) and the visited irregular points, by index
    int it = L.irregular ? T.ifirst[k] : -1;
    bool reg_todo = (T.fl[k] & F_REGVIS) != 0;
    const int r0 = (it >= 0) ? L.irr_reg0[it] : 0;
    for (;;) {
        while (it >= 0 && !T.ivis[it]) it = T.inext[it];
        int node;
        if (reg_todo && (it < 0 || r0 < L.irr_pt[it])) {
            node = k;
            reg_todo = false;
        } else if (it >= 0) {
            node = T.inode[it];
            it = T.inext[it];
        } else {
            break;
        }
        take(rp_find(T, node));
    }
    if (oc < 0) return;  // nothing labelled on either side: left alone (ssc.cpp:332-341, no branch assigns)
    const int rk = rp_find(T, k);
    if (rk != oc) T.par[rk] = oc;  // (the voxel's unlabelled rest: a root of its own until now)
    if (L.irregular)
        for (int j = T.ifirst[k]; j >= 0; j = T.inext[j]) {
            const int rj = rp_find(T, T.inode[j]);
            if (rj != oc) T.par[rj] = oc;
        }
    T.fl[k] |= F_FULL;
}

// what the visits of one node do, given the labelling times: is the visitor labelled when visit e starts (cs); q[e] = first
// listed voxel that holds a label by then (-1 none: the visit opens a cluster when the visitor carries none either).  The listed
// voxels from q on are joined.  One walk over the list serves the node's (up to three) visits.
template <int CAP>
__device__ __forceinline__ void rp_visit3(const Rp<CAP>& T, const int16_t* row16, int home, const int (&ie)[3], bool optimistic, bool (&cs)[3],
                                          int (&q)[3], int (&nb)[27]) {
    const uint4* row = reinterpret_cast<const uint4*>(row16);
    const uint4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
    const unsigned w[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
    const int th = home >= 0 ? T.T[home] : kInf;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        cs[e] = ie[e] != kInf && (optimistic || th < ie[e]);
        q[e] = -1;
    }
#pragma unroll
    for (int p = 0; p < 27; ++p) {
        const int u = (int)(int16_t)((w[p >> 1] >> ((p & 1) * 16)) & 0xffffu);
        nb[p] = u;
        if (u >= 0) {
            int lab;  // labelled before time t <=> lab < t
            if constexpr (Rp<CAP>::kHbm)
                lab = T.lab[u];
            else
                lab = min(T.T[u], T.fa[u]);
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if (q[e] < 0 && lab < ie[e] && ie[e] != kInf) q[e] = p;
        }
    }
}

// The classes of `set` together (classes never list each other, so walking several at once is walking each): the time of
// their last opener; and, when that is later than `t_beat`, whether the cluster this point ends in still
// carries the number it created.  All threads of the workgroup.
//   1. labelling times by Jacobi rounds over the class's visits (the fixed point is the sequential loop's: a time depends on
//      earlier times only), 2. openers = visits that start without a label and find none, 3. the partition right after the
//      last opener = unions of what every visit up to it joined, 4. the visits after it walked one by one (ssc.cpp:322-350).
template <int CAP, int TH>
__device__ RpOut replay_class(const Ln& L, const Rp<CAP>& T, const int* set, int n_set, int t_beat, int* wsum, int* bc) {
    auto in_set = [&](int c) {
        for (int t = 0; t < n_set; ++t)
            if (set[t] == c) return true;
        return false;
    };
    const int tid = threadIdx.x;
    RpOut out = {-1, 0, -1, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long ck = wall_clock64();
    auto lap = [&](int& dst) {
        const long long now = wall_clock64();
        dst = (int)(now - ck);
        ck = now;
    };
    // ---- nodes: the class's voxels (any order), then its irregular points ----
    if (tid == 0) bc[0] = 0;
    __syncthreads();
    for (int v0 = 0; v0 < L.nv; v0 += TH) {
        const int v = v0 + tid;
        const bool in = v < L.nv && in_set(L.vcl[v]);
        const unsigned long long mask = __ballot(in);
        if (mask) {
            const int lane = tid & 63, lead = __ffsll((long long)mask) - 1;
            int b0 = 0;
            if (lane == lead) b0 = atomicAdd(&bc[0], __popcll(mask));
            b0 = __shfl(b0, lead);
            const int at = b0 + __popcll(mask & ((1ull << lane) - 1ull));
            if (in && at < (Rp<CAP>::kHbm ? T.cap : CAP)) {
                L.cvl[at] = v;
                L.loc[v] = at;
            }
        }
    }
    __syncthreads();
    const int m = bc[0];
    __syncthreads();
    if (tid == 0) {
        int nn = m;
        for (int j = 0; j < L.n_irr; ++j) {
            T.inode[j] = -1;
            T.inext[j] = -1;
            T.ivis[j] = 0;
            if (in_set(L.irr_cls[j])) T.inode[j] = (int16_t)min(nn++, (Rp<CAP>::kHbm ? T.cap : CAP) - 1);
        }
        bc[0] = nn;
    }
    __syncthreads();
    const int nn = bc[0];
    auto leave = [&]() {
        for (int l = tid; l < min(m, Rp<CAP>::kHbm ? T.cap : CAP); l += TH) L.loc[L.cvl[l]] = -1;
        __syncthreads();
    };
    out.nodes = nn;
    if (nn > (Rp<CAP>::kHbm ? T.cap : CAP) || m > (Rp<CAP>::kHbm ? T.cap : CAP) || nn > 32767 || (long long)nn * 64 > (long long)L.rows_bytes || 3 * nn > L.n_raw) {  // does not fit
        leave();
        out.too_big = 1;
        return out;
    }
    if (L.irregular)
        for (int x = tid; x < nn; x += TH) T.ifirst[x] = -1;
    __syncthreads();
    if (tid == 0 && L.n_irr > 0) {  // per voxel: its irregular points of this class in index order (irr_pt is sorted)
        for (int j = L.n_irr - 1; j >= 0; --j) {
            if (T.inode[j] < 0) continue;
            const int hv = L.irr_home[j];
            const int l = hv >= 0 ? L.loc[hv] : -1;
            if (l >= 0) {
                T.inext[j] = (int16_t)T.ifirst[l];
                T.ifirst[l] = j;
            }
        }
    }
    auto irr_of_node = [&](int x) {
        int j = 0;
        while (T.inode[j] != x) ++j;
        return j;
    };
    // ---- lists: 27 cells around a voxel's own cell / around an irregular point's triple, found row by row (the three
    // sectors of a (range, azimuth) pair are consecutive keys: one search, then the records behind it) ----
    {
        const bool keys_fit = !Rp<CAP>::kHbm && nn >= 96 && (size_t)L.nv * 4 <= (size_t)CAP * 16;  // (T / fa / Tn / par are not in use yet)
        int32_t* lk = T.T;
        if (keys_fit)
            for (int v = tid; v < L.nv; v += TH) lk[v] = L.vkey[v];
        __syncthreads();
        Ln Lk = L;
        Lk.lkeys = keys_fit ? lk : nullptr;
        if constexpr (Rp<CAP>::kHbm) {  // the LDS table of the rounds is idle while the lists are built: a dense sample of the keys (every 4th of a 128-beam scan's 72 k)
            int sh = 0;  //   leaves two look-ups in HBM per row instead of seven behind the 1024 samples of the LDS passes
            while (((L.nv + (1 << sh) - 1) >> sh) > T.cap) ++sh;
            const int ns2 = (L.nv + (1 << sh) - 1) >> sh;
            for (int j = tid; j < ns2; j += TH) T.Tn[j] = L.vkey[(size_t)j << sh];
            Lk.skey = T.Tn;
            Lk.sshift = sh;
            Lk.ns = ns2;
            __syncthreads();
        }
        const int32_t* kk = keys_fit ? lk : L.vkey;
        for (int w = tid; w < nn * 9; w += TH) {
            const int x = w / 9, rw = w - x * 9;
            const int dx = rw / 3 - 1, dz = rw % 3 - 1;
            int r, s3, a;
            bool ok;
            if (x < m) {
                ok = cell_of_voxel(L, L.cvl[x], r, s3, a);
            } else {
                decode3(L.idx3[L.irr_pt[irr_of_node(x)]], r, s3, a);
                ok = true;
            }
            int16_t* row = T.rows + (size_t)x * 32;
            int got[3] = {-1, -1, -1};
            const int rr = r + dx, aa = a + dz;
            if (ok && rr >= 0 && rr < L.R && aa >= 0 && aa < L.Az) {
                const int s_lo = max(s3 - 1, 0), s_hi = min(s3 + 1, L.S - 1);
                if (s_lo <= s_hi) {
                    const int k_lo = aa * L.R * L.S + rr * L.S + s_lo, k_hi = k_lo + (s_hi - s_lo);
                    for (int j = lower_slot(Lk, k_lo); j < L.nv; ++j) {
                        const int key = kk[j];
                        if (key > k_hi) break;
                        got[key - k_lo + (s_lo - (s3 - 1))] = L.loc[j];  // (closure: a listed voxel belongs to the lister's class)
                    }
                }
            }
            for (int dy = 0; dy < 3; ++dy) row[(dx + 1) * 9 + dy * 3 + (dz + 1)] = (int16_t)got[dy];
            if (rw == 0)
                for (int pad = 27; pad < 32; ++pad) row[pad] = -1;
        }
        __syncthreads();
    }
    for (int x = tid; x < nn; x += TH) {
        T.T[x] = kInf;
        T.fa[x] = x < m ? L.fp[L.cvl[x]] : kInf;
        if constexpr (Rp<CAP>::kHbm) T.lab[x] = T.fa[x];
    }
    // the visits of node x: a voxel's first three regular points, an irregular point itself (cached); `home` = the voxel whose
    // full labelling labels the visitor before it starts
    for (int x = tid; x < nn; x += TH) {
        int ie[3] = {kInf, kInf, kInf};
        if (x < m) {
            const int v = L.cvl[x];
            for (int e = 0; e < 3; ++e) {
                ie[e] = reg_of(L, v, e);
                if (ie[e] == kInf) break;
            }
        } else {
            ie[0] = L.irr_pt[irr_of_node(x)];
        }
        for (int e = 0; e < 3; ++e) T.ev3[3 * x + e] = ie[e];
    }
    __syncthreads();
    auto visits_of = [&](int x, int (&ie)[3], int& home) {
        for (int e = 0; e < 3; ++e) ie[e] = T.ev3[3 * x + e];
        if (x < m) {
            home = x;
        } else {
            const int hv = L.irr_home[irr_of_node(x)];
            home = hv >= 0 ? L.loc[hv] : -1;
        }
    };
    out.nodes = nn;
    lap(out.c_build);
    // ---- 1. labelling times ----
    bool converged = false;
    for (int round = 0; round < 96 && !converged; ++round) {
        for (int x = tid; x < nn; x += TH) T.Tn[x] = kInf;
        __syncthreads();
        for (int x = tid; x < nn; x += TH) {
            int ie[3], home, q[3], nb[27];
            bool cs[3];
            visits_of(x, ie, home);
            if (ie[0] == kInf) continue;
            // round 0 starts from the optimistic end (every visit labels its whole list): any start reaches the same fixed
            // point, this one in fewer rounds than "nothing is ever labelled"
            rp_visit3(T, T.rows + (size_t)x * 32, home, ie, round == 0, cs, q, nb);
            int from[3];
#pragma unroll
            for (int e = 0; e < 3; ++e) from[e] = ie[e] == kInf ? 99 : ((cs[e] || q[e] < 0) ? 0 : q[e]);
#pragma unroll
            for (int p = 0; p < 27; ++p) {
                const int u = nb[p];
                if (u < 0 || u >= m) continue;
                const int te = p >= from[0] ? ie[0] : (p >= from[1] ? ie[1] : (p >= from[2] ? ie[2] : kInf));  // the earliest visit that joins it
                if (te < T.Tn[u]) atomicMin(&T.Tn[u], te);
            }
        }
        __syncthreads();
        int changed = 0;
        for (int x = tid; x < m; x += TH) {
            const int tv = T.Tn[x];
            changed |= tv != T.T[x];
            T.T[x] = tv;
            if constexpr (Rp<CAP>::kHbm) T.lab[x] = min(tv, T.fa[x]);
        }
        converged = !__syncthreads_or(changed);
        out.rounds = round + 1;
    }
    lap(out.c_jacobi);
    if (!converged) {  // (never seen; reported like a class that does not fit)
        leave();
        out.too_big = 1;
        return out;
    }
    // ---- 2. the last opener ----
    if (tid == 0) bc[1] = -1;
    __syncthreads();
    for (int x = tid; x < nn; x += TH) {
        int ie[3], home, q[3], nb[27];
        bool cs[3];
        visits_of(x, ie, home);
        if (ie[0] == kInf) continue;
        ie[1] = ie[2] = kInf;  // (only a voxel's first regular point or an irregular point can open a cluster)
        rp_visit3(T, T.rows + (size_t)x * 32, home, ie, false, cs, q, nb);
        if (!cs[0] && q[0] < 0) atomicMax(&bc[1], ie[0]);
    }
    __syncthreads();
    const int t_open = bc[1];
    out.last_open = t_open;
    lap(out.c_open);
    if (t_open <= t_beat) {
        leave();
        return out;
    }
    // ---- 3. the partition right after the last opener's visit ----
    for (int x = tid; x < nn; x += TH) T.par[x] = x;
    if (tid == 0) bc[2] = -1;
    __syncthreads();
    for (int x = tid; x < nn; x += TH) {
        int ie[3], home, q[3], nb[27];
        bool cs[3];
        visits_of(x, ie, home);
        if (ie[0] > t_open) continue;
        // what a later visit joins contains what the earlier ones joined (labels only spread): the last visit up to t_open decides
        const int el = ie[2] <= t_open ? 2 : (ie[1] <= t_open ? 1 : 0);
        if (ie[el] == t_open) bc[2] = x;
        ie[0] = ie[el];
        ie[1] = ie[2] = kInf;
        rp_visit3(T, T.rows + (size_t)x * 32, home, ie, false, cs, q, nb);
        const int from = (cs[0] || q[0] < 0) ? 0 : q[0];
        int r = x;
        for (int p = from; p < 27; ++p) {
            const int u = nb[p];
            if (u < 0) continue;
            r = rp_union(T, r, u);
            if (L.irregular)
                for (int j = T.ifirst[u]; j >= 0; j = T.inext[j]) r = rp_union(T, r, T.inode[j]);
        }
    }
    __syncthreads();
    for (int x = tid; x < nn; x += TH) {
        uint8_t f = 0;
        if (x < m) {
            if (T.T[x] <= t_open) f |= F_FULL;
            if (T.ev3[3 * x] <= t_open) f |= F_REGVIS;
        } else {
            T.ivis[irr_of_node(x)] = L.irr_pt[irr_of_node(x)] <= t_open;
        }
        T.fl[x] = f;
    }
    lap(out.c_cc);
    // ---- 4. the visits after it, in order ----
    int tmax = -1;
    for (int x = tid; x < nn; x += TH) {
        int ie[3], home;
        visits_of(x, ie, home);
        for (int e = 0; e < 3 && ie[e] != kInf; ++e)
            if (ie[e] > t_open) tmax = max(tmax, ie[e]);
    }
    for (int d = 32; d > 0; d >>= 1) tmax = max(tmax, __shfl_xor(tmax, d));
    __syncthreads();
    if ((tid & 63) == 0) wsum[tid >> 6] = tmax;
    __syncthreads();
    for (int w = 0; w < (TH / 64); ++w) tmax = max(tmax, wsum[w]);
    __syncthreads();
    int n_ev = 0;
    if (tmax > t_open) {
        const int w_lo = (t_open + 1) >> 5, w_hi = tmax >> 5;
        for (int w = w_lo + tid; w <= w_hi; w += TH) L.bits[w] = 0u;
        __syncthreads();
        for (int x = tid; x < nn; x += TH) {
            int ie[3], home;
            visits_of(x, ie, home);
            for (int e = 0; e < 3 && ie[e] != kInf; ++e)
                if (ie[e] > t_open) {
                    L.evn[ie[e]] = x;
                    atomicOr(&L.bits[ie[e] >> 5], 1u << (ie[e] & 31));
                }
        }
        __syncthreads();
        for (int w0 = w_lo; w0 <= w_hi; w0 += TH) {
            const int w = w0 + tid;
            uint32_t word = (w <= w_hi) ? __hip_atomic_load(&L.bits[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            int total;
            int o = n_ev + block_excl_scan<TH>(__popc(word), total, wsum);
            while (word) {
                const int b = __ffs((int)word) - 1;
                word &= word - 1;
                L.evl[o++] = (w << 5) + b;
            }
            n_ev += total;
        }
        __syncthreads();
    }
    out.n_events = n_ev;
    // One wave walks; lane p looks at the p-th listed voxel.  A visit whose list touches irregular points (or that IS one)
    // takes the scalar routine; every other one reads flags and roots of its 27 voxels at once and only steps through the
    // DISTINCT classes it meets, in list order (the neighbour's name wins each time, ssc.cpp:329).
    int kroot = -1;
    if (tid < 64) kroot = rp_find_ro(T, bc[2]);
    if (tid == 0) bc[3] = 0;  // a visit after the "last" opener opened a cluster: the model is broken (never seen; reported as unknown)
    for (int e0 = 0; e0 < n_ev; e0 += kLnChunk) {
        const int ce = min(kLnChunk, n_ev - e0);
        __syncthreads();
        for (int e = tid; e < ce; e += TH) {
            const int i = __hip_atomic_load(&L.evl[e0 + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            T.ev_t[e] = i;
            T.ev_x[e] = __hip_atomic_load(&L.evn[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (tid < 64) {
            const int lane = tid;
            auto wsync = [&]() {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            };
            for (int e = 0; e < ce && kroot >= 0; ++e) {  // (once the number is gone nothing brings it back)
                const int x = T.ev_x[e];
                const bool reg = x < m;
                const int16_t* row = T.rows + (size_t)x * 32;
                const int k = lane < 27 ? (int)row[lane] : -1;
                bool slow = !reg;
                if (L.irregular && k >= 0) slow |= T.ifirst[k] >= 0;
                if (__any(slow)) {  // every lane runs the same scalar walk (same values stored by all)
                    int lv, jirr = -1;
                    if (reg) {
                        lv = x;
                    } else {
                        jirr = irr_of_node(x);
                        const int hv = L.irr_home[jirr];
                        lv = hv >= 0 ? L.loc[hv] : -1;
                    }
                    int oc = (lv >= 0 && (T.fl[lv] & F_FULL)) ? rp_find_ro(T, lv) : -1;  // the visiting point's own label
                    for (int pos = 0; pos < 27; ++pos) {
                        const int u = __shfl(k, pos);
                        if (u >= 0) rp_meet(L, T, oc, u, kroot);
                    }
                    if (oc < 0) {
                        bc[3] = 1;
                        break;
                    }
                    if (reg) {
                        if (!(T.fl[x] & (F_FULL | F_REGVIS))) {  // only the visited point carries the label: the voxel node holds its class
                            const int r = rp_find_ro(T, x);
                            if (r != oc) T.par[r] = oc;
                        }
                        T.fl[x] |= F_REGVIS;
                    } else {
                        const int r = rp_find_ro(T, x);
                        if (r != oc) T.par[r] = oc;
                        T.ivis[jirr] = 1;
                    }
                    wsync();
                    continue;
                }
                const bool valid = k >= 0;
                const uint8_t f = valid ? T.fl[k] : (uint8_t)0;
                const bool lab = valid && (f & (F_FULL | F_REGVIS));
                int root = lab ? rp_find_ro(T, k) : -1;
                int oc = (T.fl[x] & F_FULL) ? rp_find_ro(T, x) : -1;  // the visiting point's own label
                const unsigned long long labm = __ballot(lab), valm = __ballot(valid);
                unsigned long long joined = valm;
                if (oc < 0) {
                    if (!labm) {
                        if (lane == 0) bc[3] = 1;
                        break;
                    }
                    joined = valm & ~((1ull << (__ffsll((long long)labm) - 1)) - 1ull);  // the voxels before the first labelled one are left alone
                }
                unsigned long long todo = labm & joined;
                while (todo) {
                    const int b = __ffsll((long long)todo) - 1;
                    const int c = __shfl(root, b);
                    if (oc < 0) {
                        oc = c;
                    } else if (c != oc) {  // mergeClusters(oc, nc), ssc.cpp:329
                        if (oc == kroot) kroot = -1;
                        if (lane == 0) T.par[oc] = c;
                        if (root == oc) root = c;
                        oc = c;
                    }
                    todo &= ~__ballot(root == oc);
                }
                wsync();
                const bool mine = (joined >> lane) & 1ull;
                if (mine) {
                    if (!lab) T.par[k] = oc;  // (an unlabelled voxel is a root of its own until now)
                    T.fl[k] = f | F_FULL;
                }
                wsync();
                if (lane == 0) {
                    if (!(T.fl[x] & (F_FULL | F_REGVIS))) {  // only the visited point carries the label: the voxel node holds its class
                        const int r = rp_find_ro(T, x);
                        if (r != oc) T.par[r] = oc;
                    }
                    T.fl[x] |= F_REGVIS;
                }
                wsync();
            }
        }
    }
    if (tid == 0) bc[6] = kroot;
    __syncthreads();
    kroot = bc[6];
    lap(out.c_walk);
    // ---- result ----
    if (tid == 0) {
        bc[0] = kroot >= 0 ? rp_find_ro(T, kroot) : -1;
        bc[4] = kInf;  // canon
        bc[5] = kInf;  // slot
    }
    __syncthreads();
    if (bc[3]) {
        leave();
        out.too_big = 1;
        return out;
    }
    const int root = bc[0];
    out.fin = root >= 0;
    if (root >= 0) {
        int canon = kInf, slot = kInf;
        for (int x = tid; x < nn; x += TH) {
            if (rp_find_ro(T, x) != root) continue;
            if (x < m) {
                const int v = L.cvl[x];
                canon = min(canon, reg_of(L, v, 0));
                if (!L.irregular || regular_point(L, L.fp[v])) slot = min(slot, v);
            } else {
                const int j = irr_of_node(x);
                canon = min(canon, L.irr_pt[j]);
                const int hv = L.irr_home[j];
                if (hv >= 0 && L.fp[hv] == L.irr_pt[j]) slot = min(slot, hv);
            }
        }
        atomicMin(&bc[4], canon);
        atomicMin(&bc[5], slot);
        __syncthreads();
        out.canon = bc[4] == kInf ? -1 : bc[4];
        out.slot = bc[5] == kInf ? -1 : bc[5];
    }
    leave();
    return out;
}

template <int CAP>
constexpr size_t ln_lds_bytes() {
    return (size_t)CAP * (4 + 4 + 4 + 4 + 1) + kLnIrr * (2 + 2 + 1) + kLnChunk * 8  // tables of a class
           + kLnSamples * 4 + kLnIrr * 5 * 4 + kLnNames * 8 + 512;
}

// per wave: the 5 x 5 x 5 block of cells around a candidate
struct Blk {
    int32_t fa[125];      // first point of the voxel in the cell, kInf: no voxel
    int32_t reg[125][3];  // its first three regular points
};

// One scan: everything above, by the TH threads of one workgroup on `ln_smem` (ln_lds_bytes<CAP>() bytes of LDS).  A scan whose
// classes do not fit CAP nodes goes to `redo` / `redo_big` (lists on the device, [n_scans] = how many) or is reported unknown.
// MODE 0: the whole pass.  MODE 1 (triage): everything up to the set of classes to follow; the set, the irregular points and the
// class table are left in A.ln_state and the scan is listed for the pass whose tables hold the set (lists[0 / 1 / 2]: up to
// kLnCapTiny / kLnCapSmall / kLnCapBig nodes).  MODE 2: follows the set a triage left.
constexpr int kLnStateWords = 8 + (kLnMarked + 1) + kLnIrr * 5 + kLnNames * 2;
template <int CAP, int TH, int MODE, bool GT = false>
__device__ void ln_scan(const DevParams& P, const Arena& A, int s, unsigned char* ln_smem, int32_t* redo, int32_t* redo_big, int big_from) {
    __shared__ int wsum[2 * kLnMaxWaves + 2];
    __shared__ int bc[8];
    __shared__ int mk_cls[kLnMarked + 1], mk_t[kLnMarked + 1], n_mk, n_pairs_s, n_irr_s, fail_s;
    __shared__ int red[kLnMaxWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long pt0 = wall_clock64();
    Ln L;
    const int base = A.scan_off[s];
    L.n = A.counts[s * 8 + 4];
    L.nv = A.counts[s * 8 + 6];
    int32_t* outp = A.cc_last + (size_t)s * 4;
    if (L.n <= 0 || L.nv <= 0) {
        if (tid == 0) outp[0] = outp[1] = -1, outp[2] = 0, outp[3] = 0;
        return;
    }
    L.vkey = A.vox_key + base;
    L.vbeg = A.vox_pt_begin + base + s;
    L.vpts = A.vox_pts + base;
    L.ptc = A.pt_cluster + base;
    L.idx3 = A.apri_idx3 + base;
    L.akey = A.apri_key + base;
    L.R = P.bin.range_num, L.S = P.bin.sector_num, L.Az = P.bin.azimuth_num;
    L.span = (long long)L.R * L.S * L.Az;
    L.irregular = A.scan_irr[s] != 0;
    L.vcl = A.tmp_vox_key + base;
    L.loc = A.tmp_vox_begin + base;
    L.evn = (int32_t*)A.tmp_vox_av + base;
    L.bits = (uint32_t*)A.tmp_vox_cov + base;
    L.evl = (int32_t*)(A.vkeys + base);
    L.cvl = (int32_t*)A.sorted_xyz + 3 * (size_t)base;  // (nothing the tracking kernels touch: this pass runs beside them)
    L.fp = (int32_t*)A.sorted_idx + base;
    L.n_raw = A.scan_off[s + 1] - base;
    L.rows_bytes = 8ll * L.n_raw;
    // LDS carve
    unsigned char* q = ln_smem;
    auto take = [&](size_t bytes) {
        unsigned char* p = q;
        q += (bytes + 15) & ~(size_t)15;
        return p;
    };
    L.skey = (int32_t*)take(kLnSamples * 4);
    L.irr_pt = (int32_t*)take(kLnIrr * 4);
    L.irr_home = (int32_t*)take(kLnIrr * 4);
    L.irr_cls = (int32_t*)take(kLnIrr * 4);
    L.irr_reg0 = (int32_t*)take(kLnIrr * 4);
    L.irr_xs = (int32_t*)take(kLnIrr * 4);
    L.cname = (int32_t*)take(kLnNames * 4);
    L.crep = (int32_t*)take(kLnNames * 4);
    unsigned char* ovl = q;  // from here on: the replay tables, overlaid by the link pairs / the candidates' blocks
    Rp<CAP> T;
    static_assert(GT == Rp<CAP>::kHbm, "the pass with its tables in arena scratch is the kLnCapHugeNodes one");
    if constexpr (GT) {
        // the read-mostly class tables (times, first points, their minimum, flags: 13 bytes per node) in arena scratch -- `seg`, 4 bytes per input
        // point, dead since k_emit -- and the one the rounds hammer with atomics in LDS: the next round's times, whose room the
        // partition's parent array takes over when the rounds are done (the two are never live together)
        const int fit = (int)min((long long)CAP, (4ll * L.n_raw - 64) / 13);
        T.cap = fit > 0 ? (fit & ~3) : 0;
        T.in_hbm = true;
        int32_t* g = (int32_t*)(A.seg + base);
        T.T = g;
        T.fa = g + T.cap;
        T.lab = g + 2 * (size_t)T.cap;
        T.fl = (uint8_t*)(g + 3 * (size_t)T.cap);
        T.Tn = (int32_t*)take((size_t)CAP * 4);
        T.par = T.Tn;
    } else {
        T.cap = CAP;
        T.in_hbm = false;
        T.lab = nullptr;
        T.T = (int32_t*)take((size_t)CAP * 4);
        T.fa = (int32_t*)take((size_t)CAP * 4);
        T.Tn = (int32_t*)take((size_t)CAP * 4);
        T.par = (int32_t*)take((size_t)CAP * 4);
        T.fl = (uint8_t*)take(CAP);
    }
    T.inext = (int16_t*)take(kLnIrr * 2);
    T.inode = (int16_t*)take(kLnIrr * 2);
    T.ivis = (uint8_t*)take(kLnIrr);
    T.ev_t = (int32_t*)take(kLnChunk * 4);
    T.ev_x = (int32_t*)take(kLnChunk * 4);
    T.rows = (int16_t*)(A.keys + base);
    T.ev3 = (int32_t*)A.zkey + base;
    T.ifirst = (int32_t*)A.sorted_xyz + 3 * (size_t)base + L.n_raw;
    int2* pairs = (int2*)ovl;        // [kLnPairs]
    Blk* blk = (Blk*)ovl + wave;     // [(TH / 64)]
    constexpr size_t kOvl = (size_t)(GT ? kLnCapBigNodes : CAP) * 17 + kLnIrr * 5 + kLnChunk * 8;
    static_assert(sizeof(Blk) * (TH / 64) <= kOvl && kLnPairs * 8 <= kOvl,
                  "overlays fit the class tables (and the walk's staging behind them: neither is live at the time)");

    L.n_irr = 0;
    L.n_xs = 0;
    L.n_names = 0;
    L.lkeys = nullptr;
    // sampled keys
    L.sshift = 0;
    while (((L.nv + (1 << L.sshift) - 1) >> L.sshift) > kLnSamples) ++L.sshift;
    L.ns = (L.nv + (1 << L.sshift) - 1) >> L.sshift;
    for (int j = tid; j < L.ns; j += TH) L.skey[j] = L.vkey[(size_t)j << L.sshift];
    if (tid == 0) n_mk = 0, n_pairs_s = 0, n_irr_s = 0, fail_s = 0;
    __syncthreads();

    int32_t* state = A.ln_state + (size_t)s * kLnStateWords;  // {n_set, n_irr, n_xs, n_names, 0..}, set, irr_pt / home / cls / reg0 / xs, cname, crep
    bool unknown = false;
    long long pt1 = pt0, pt2 = pt0, pt3 = pt0, pt4 = pt0;
    int prof_cand = 0, prof_mk = 0;
    if (MODE != 2) {
    // ---- index triples outside the grid: the points, then which clusters their lists tie into one closure class ----
    if (L.irregular) {
        const int32_t* il = A.irr_list + (size_t)s * (kIrrListCap + 1);
        const int listed = il[0];
        if (listed >= 0) {  // k_emit listed them while it binned the scan
            for (int x = tid; x < min(listed, kLnIrr); x += TH) L.irr_pt[x] = il[1 + x];
            if (tid == 0) n_irr_s = listed;
        } else {
            for (int i0 = 0; i0 < L.n; i0 += TH) {
                const int i = i0 + tid;
                if (i < L.n && !regular_point(L, i)) {
                    const int x = atomicAdd(&n_irr_s, 1);
                    if (x < kLnIrr) L.irr_pt[x] = i;
                }
            }
        }
        __syncthreads();
        if (n_irr_s > kLnIrr) {
            if (tid == 0) {
                outp[0] = outp[1] = -1, outp[2] = 2, outp[3] = 0;
                atomicAdd(&A.ln_stats[1], 1);
            }
            return;
        }
        L.n_irr = n_irr_s;
        if (tid == 0) {  // by point index (a handful)
            for (int a = 1; a < L.n_irr; ++a) {
                const int v = L.irr_pt[a];
                int b = a - 1;
                while (b >= 0 && L.irr_pt[b] > v) {
                    L.irr_pt[b + 1] = L.irr_pt[b];
                    --b;
                }
                L.irr_pt[b + 1] = v;
            }
        }
        __syncthreads();
        for (int j = tid; j < L.n_irr; j += TH) {
            L.irr_home[j] = slot_of_key(L, L.akey[L.irr_pt[j]]);
            L.irr_cls[j] = L.ptc[L.irr_pt[j]];
        }
        __syncthreads();
        if (tid == 0) {  // the voxels concerned: sorted, unique
            int nx = 0;
            for (int j = 0; j < L.n_irr; ++j) {
                const int h = L.irr_home[j];
                if (h < 0) continue;
                int b = nx - 1;
                bool dup = false;
                for (int t = 0; t < nx; ++t) dup |= L.irr_xs[t] == h;
                if (dup) continue;
                while (b >= 0 && L.irr_xs[b] > h) {
                    L.irr_xs[b + 1] = L.irr_xs[b];
                    --b;
                }
                L.irr_xs[b + 1] = h;
                ++nx;
            }
            bc[7] = nx;
        }
        __syncthreads();
        L.n_xs = bc[7];
        for (int j = tid; j < L.n_irr; j += TH) L.irr_reg0[j] = L.irr_home[j] >= 0 ? reg_of(L, L.irr_home[j], 0) : kInf;
        __syncthreads();
        auto link = [&](int a, int b) {
            if (a == b) return;
            const int x = atomicAdd(&n_pairs_s, 1);
            if (x < kLnPairs) pairs[x] = make_int2(a, b);
        };
        // one wave per irregular point: what it lists, and who lists its voxel
        for (int j = wave; j < L.n_irr; j += (TH / 64)) {
            const int p = L.irr_pt[j], cp = L.ptc[p];
            int r, s3, a;
            decode3(L.idx3[p], r, s3, a);
            if (lane < 27) {
                const int dx = lane / 9 - 1, dy = (lane / 3) % 3 - 1, dz = lane % 3 - 1;
                const int k = slot_of_cell(L, r + dx, s3 + dy, a + dz);
                if (k >= 0) {
                    const int r0 = reg_of(L, k, 0);
                    if (r0 != kInf) link(cp, L.ptc[r0]);
                    for (int j2 = 0; j2 < L.n_irr; ++j2)
                        if (L.irr_home[j2] == k) link(cp, L.ptc[L.irr_pt[j2]]);
                }
            }
            const int hv = L.irr_home[j];
            int hr, hs, ha;
            if (hv >= 0 && cell_of_voxel(L, hv, hr, hs, ha) && lane < 27) {  // the regular points around the voxel's own cell list it
                const int dx = lane / 9 - 1, dy = (lane / 3) % 3 - 1, dz = lane % 3 - 1;
                const int w = slot_of_cell(L, hr + dx, hs + dy, ha + dz);
                if (w >= 0) {
                    const int r0 = reg_of(L, w, 0);
                    if (r0 != kInf) link(cp, L.ptc[r0]);
                }
            }
        }
        __syncthreads();
        if (n_pairs_s > kLnPairs) {
            if (tid == 0) {
                outp[0] = outp[1] = -1, outp[2] = 2, outp[3] = 0;
                atomicAdd(&A.ln_stats[1], 1);
            }
            return;
        }
        if (tid == 0 && n_pairs_s > 0) {  // union-find over the names that occur (a few dozen)
            int nn = 0;
            bool over = false;
            auto idx_of = [&](int name) -> int {
                for (int t = 0; t < nn; ++t)
                    if (L.cname[t] == name) return t;
                if (nn >= kLnNames) {
                    over = true;
                    return 0;
                }
                L.cname[nn] = name;
                L.crep[nn] = nn;
                return nn++;
            };
            auto fnd = [&](int x) {
                while (L.crep[x] != x) x = L.crep[x] = L.crep[L.crep[x]];
                return x;
            };
            for (int t = 0; t < n_pairs_s; ++t) {
                const int a = fnd(idx_of(pairs[t].x)), b = fnd(idx_of(pairs[t].y));
                if (a != b) {  // the class is called after its smallest name (= its birth)
                    if (L.cname[a] < L.cname[b])
                        L.crep[b] = a;
                    else
                        L.crep[a] = b;
                }
            }
            for (int t = 0; t < nn; ++t) pairs[t] = make_int2(L.cname[t], L.cname[fnd(t)]);
            // sorted by name for cls_of
            for (int a = 1; a < nn; ++a) {
                const int2 v = pairs[a];
                int b = a - 1;
                while (b >= 0 && pairs[b].x > v.x) {
                    pairs[b + 1] = pairs[b];
                    --b;
                }
                pairs[b + 1] = v;
            }
            for (int t = 0; t < nn; ++t) {
                L.cname[t] = pairs[t].x;
                L.crep[t] = pairs[t].y;
            }
            bc[0] = over ? -1 : nn;
        } else if (tid == 0) {
            bc[0] = 0;
        }
        __syncthreads();
        if (bc[0] < 0) {
            if (tid == 0) {
                outp[0] = outp[1] = -1, outp[2] = 2, outp[3] = 0;
                atomicAdd(&A.ln_stats[1], 1);
            }
            return;
        }
        L.n_names = bc[0];
        __syncthreads();
        for (int j = tid; j < L.n_irr; j += TH) L.irr_cls[j] = cls_of(L, L.irr_cls[j]);
        __syncthreads();
    }

    // ---- A: closure class of every voxel (through its first point), the latest-born class.  k_cc_scan left the cluster of every
    // voxel's first point in vcl and the point in fp (-1: not settled there, looked up here) ----
    int lmax = -1;
    for (int v = tid; v < L.nv; v += TH) {
        int c = L.vcl[v];
        if (c < 0) c = L.ptc[L.fp[v]];
        if (L.n_names) c = cls_of(L, c);
        L.vcl[v] = c;
        L.loc[v] = -1;
        lmax = max(lmax, c);
    }
    for (int j = tid; j < L.n_irr; j += TH) lmax = max(lmax, L.irr_cls[j]);
    for (int d = 32; d > 0; d >>= 1) lmax = max(lmax, __shfl_xor(lmax, d));
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    for (int w = 0; w < (TH / 64); ++w) lmax = max(lmax, red[w]);
    __syncthreads();
    const int CL = lmax;

    // ---- B: later points of the other classes that could still open a cluster (the latest-born class opens one with its
    // first point, at time CL: only what comes after that matters) ----
    pt1 = pt2 = pt3 = wall_clock64();
    {
        const int t_best = CL;
        // candidates: voxels whose first regular point comes after t_best (a few), irregular points after it -- listed first
        if (tid == 0) bc[5] = 0;
        __syncthreads();
        // (a voxel that holds irregular points goes by its first REGULAR point: those few voxels take the second loop)
        auto list_voxel = [&](int v, int ti) {
            int cr, cs, ca;
            bool c = true;
            // the cells it lists: a visited point there and it is no opener (nine in ten stop here, in the thread that found
            // them: the wave-wide certificate below is for the rest)
            if (cell_of_voxel(L, v, cr, cs, ca))
                for (int p = 0; p < 27 && c; ++p) {
                    const int k = slot_of_cell(L, cr + p / 9 - 1, cs + (p / 3) % 3 - 1, ca + p % 3 - 1);
                    if (k >= 0 && L.fp[k] < ti) c = false;
                }
            if (c) L.evl[atomicAdd(&bc[5], 1)] = v;
        };
        for (int v = tid; v < L.nv; v += TH) {
            const int f = L.fp[v];
            if (f > t_best && L.vcl[v] != CL && !(L.n_xs && holds_irregular(L, v))) list_voxel(v, f);
        }
        for (int t = tid; t < L.n_xs; t += TH) {
            const int v = L.irr_xs[t];
            const int r0 = reg_of(L, v, 0);
            if (r0 != kInf && r0 > t_best && (L.n_names ? cls_of(L, L.ptc[r0]) : L.ptc[r0]) != CL) list_voxel(v, r0);
        }
        for (int j = tid; j < L.n_irr; j += TH)
            if (L.irr_pt[j] > t_best && L.irr_cls[j] != CL) L.evl[atomicAdd(&bc[5], 1)] = L.nv + j;
        __syncthreads();
        const int n_cand = bc[5];
        prof_cand = n_cand;
        for (int c0 = wave; c0 < n_cand; c0 += (TH / 64)) {
            int ti = kInf, cr = 0, cs = 0, ca = 0, ccls = -1;
            bool cand = false;
            const int item = __hip_atomic_load(&L.evl[c0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (item < L.nv) {
                const int v = item;
                const int r0 = reg_of(L, v, 0);
                cand = cell_of_voxel(L, v, cr, cs, ca);
                ti = r0;
                ccls = L.n_names ? cls_of(L, L.ptc[r0]) : L.ptc[r0];
            } else {
                const int j = item - L.nv;
                cand = true;
                ti = L.irr_pt[j];
                decode3(L.idx3[ti], cr, cs, ca);
                ccls = L.irr_cls[j];
            }
            if (!cand) continue;
            // the cells it lists first: a visited point there and it is no opener (nine in ten stop here)
            int fa_in = kInf;
            if (lane < 27) {
                const int k = slot_of_cell(L, cr + lane / 9 - 1, cs + (lane / 3) % 3 - 1, ca + lane % 3 - 1);
                if (k >= 0) fa_in = L.fp[k];
            }
            if (__any(fa_in < ti)) continue;
            // the block of cells around it
            for (int c = lane; c < 125; c += 64) {
                const int dx = c / 25 - 2, dy = (c / 5) % 5 - 2, dz = c % 5 - 2;
                const int k = slot_of_cell(L, cr + dx, cs + dy, ca + dz);
                blk->fa[c] = k >= 0 ? L.fp[k] : kInf;
                for (int e = 0; e < 3; ++e) blk->reg[c][e] = k >= 0 ? reg_of(L, k, e) : kInf;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            bool possible = true, cert = false;
            if (lane < 27) {
                const int wx = lane / 9 - 1, wy = (lane / 3) % 3 - 1, wz = lane % 3 - 1;
                const int cw = (wx + 2) * 25 + (wy + 2) * 5 + (wz + 2);
                const int faw = blk->fa[cw];
                if (faw != kInf) {
                    if (faw < ti) possible = false;  // a visited point in the list: the candidate takes its label
                    // who lists w: the regular points of the cells around it
                    for (int u = 0; u < 27 && !cert; ++u) {
                        const int ux = wx + u / 9 - 1, uy = wy + (u / 3) % 3 - 1, uz = wz + u % 3 - 1;  // cell of the lister
                        if (ux < -2 || ux > 2 || uy < -2 || uy > 2 || uz < -2 || uz > 2) continue;
                        const int cu = (ux + 2) * 25 + (uy + 2) * 5 + (uz + 2);
                        const int* rg = blk->reg[cu];
                        if (rg[0] >= ti) continue;
                        if (rg[2] < ti) {  // the third visit labels the whole list
                            cert = true;
                            break;
                        }
                        const int posw = (wx - ux + 1) * 9 + (wy - uy + 1) * 3 + (wz - uz + 1);  // w's place in u's list
                        if (rg[1] < ti && posw >= 13) {  // the second visit finds its own voxel labelled: everything from there on joins
                            cert = true;
                            break;
                        }
                        for (int e = 0; e < 2 && !cert; ++e) {
                            const int te = rg[e];
                            if (te >= ti) break;
                            for (int z = 0; z <= posw; ++z) {  // a voxel at or before w in u's list that held a visited point at the time
                                const int kx = ux + z / 9 - 1, ky = uy + (z / 3) % 3 - 1, kz = uz + z % 3 - 1;
                                if (kx < -2 || kx > 2 || ky < -2 || ky > 2 || kz < -2 || kz > 2) continue;
                                if (blk->fa[(kx + 2) * 25 + (ky + 2) * 5 + (kz + 2)] < te) {
                                    cert = true;
                                    break;
                                }
                            }
                        }
                    }
                }
            }
            possible = !__any(!possible);
            cert = __any(cert);
            if (possible && !cert && lane == 0) {
                // its class has to be replayed
                int at = -1;
                const int cur = atomicAdd(&n_mk, 0);
                for (int t = 0; t < min(cur, kLnMarked); ++t)
                    if (mk_cls[t] == ccls) at = t;
                if (at >= 0) {
                    atomicMax(&mk_t[at], ti);
                } else {
                    const int x = atomicAdd(&n_mk, 1);
                    if (x < kLnMarked) {
                        mk_cls[x] = ccls;
                        mk_t[x] = ti;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        if (n_mk > kLnMarked) unknown = true;
        pt3 = wall_clock64();
        prof_mk = n_mk;
    }
    // ---- C: the latest-born class and the classes of the candidates no certificate settled, walked together ----
    if (!unknown) {
        if (tid == 0) {  // (racing waves may have listed a class twice: harmless)
            const int k = min(n_mk, kLnMarked);
            mk_cls[k] = CL;
            n_mk = k + 1;
            // Whatever cluster carries the number: if refineClusterByBoundingBox erased every cluster that could (ssc.cpp:437-467: a
            // class of a regular scan IS a cluster, called after its smallest point, whose type byte says so), no live cluster
            // carries it and there is nothing to walk.  (Classes tied together by irregular points are walked.)
            int all_erased = 1;
            for (int t = 0; t < n_mk; ++t) {
                const int c = mk_cls[t];
                bool tied = false;
                for (int q = 0; q < L.n_names; ++q) tied |= L.crep[q] == c;
                if (tied || A.pt_type[(size_t)base + c] != 0) all_erased = 0;
            }
            bc[6] = all_erased;
        }
        __syncthreads();
        if (bc[6]) {
            if (tid == 0) {
                outp[0] = outp[1] = -1, outp[2] = 0, outp[3] = 0;
                if (A.ln_prof) A.ln_prof[(size_t)s * 8 + 7] = -CAP;
            }
            return;
        }
    }
    }  // MODE != 2
    if (MODE == 1) {
        // how many nodes the set has decides which pass follows it; the set and what the irregular points need go to A.ln_state
        if (unknown) {
            if (tid == 0) {
                outp[0] = outp[1] = -1, outp[2] = 1, outp[3] = 0;
                atomicAdd(&A.ln_stats[0], 1);
            }
            return;
        }
        int cnt = 0;
        for (int v = tid; v < L.nv; v += TH) {
            const int c = L.vcl[v];
            bool in = false;
            for (int t = 0; t < n_mk; ++t) in |= mk_cls[t] == c;
            cnt += in ? 1 : 0;
        }
        for (int j = tid; j < L.n_irr; j += TH) {
            bool in = false;
            for (int t = 0; t < n_mk; ++t) in |= mk_cls[t] == L.irr_cls[j];
            cnt += in ? 1 : 0;
        }
        for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
        __syncthreads();
        if (lane == 0) red[wave] = cnt;
        __syncthreads();
        cnt = 0;
        for (int w = 0; w < TH / 64; ++w) cnt += red[w];
        for (int t = tid; t < n_mk; t += TH) state[8 + t] = mk_cls[t];
        int32_t* sp = state + 8 + kLnMarked + 1;
        for (int j = tid; j < L.n_irr; j += TH) {
            sp[j] = L.irr_pt[j];
            sp[kLnIrr + j] = L.irr_home[j];
            sp[2 * kLnIrr + j] = L.irr_cls[j];
            sp[3 * kLnIrr + j] = L.irr_reg0[j];
        }
        for (int j = tid; j < L.n_xs; j += TH) sp[4 * kLnIrr + j] = L.irr_xs[j];
        for (int j = tid; j < L.n_names; j += TH) {
            sp[5 * kLnIrr + j] = L.cname[j];
            sp[5 * kLnIrr + kLnNames + j] = L.crep[j];
        }
        if (tid == 0) {
            state[0] = n_mk, state[1] = L.n_irr, state[2] = L.n_xs, state[3] = L.n_names;
            if (cnt > kLnCapHugeNodes) {
                outp[0] = outp[1] = -1, outp[2] = 1, outp[3] = cnt;  // (undetermined: the fourth word reports how many nodes the set holds)
                atomicAdd(&A.ln_stats[0], 1);
            } else {
                int32_t* list = redo + (size_t)(cnt <= kLnCapTinyNodes ? 0 : (cnt <= kLnCapSmallNodes ? 1 : (cnt <= kLnCapBigNodes ? 2 : 3))) * (A.n_scans + 1);
                list[atomicAdd(list + A.n_scans, 1)] = s;
            }
        }
        return;
    }
    if (MODE == 2) {
        L.n_irr = state[1], L.n_xs = state[2], L.n_names = state[3];
        if (tid == 0) n_mk = state[0];
        for (int t = tid; t < state[0]; t += TH) mk_cls[t] = state[8 + t];
        const int32_t* sp = state + 8 + kLnMarked + 1;
        for (int j = tid; j < L.n_irr; j += TH) {
            L.irr_pt[j] = sp[j];
            L.irr_home[j] = sp[kLnIrr + j];
            L.irr_cls[j] = sp[2 * kLnIrr + j];
            L.irr_reg0[j] = sp[3 * kLnIrr + j];
        }
        for (int j = tid; j < L.n_xs; j += TH) L.irr_xs[j] = sp[4 * kLnIrr + j];
        for (int j = tid; j < L.n_names; j += TH) {
            L.cname[j] = sp[5 * kLnIrr + j];
            L.crep[j] = sp[5 * kLnIrr + kLnNames + j];
        }
        __syncthreads();
    }
    RpOut best = {-1, 0, -1, -1, 0, 1, 0, 0, 0, 0, 0, 0, 0};
    if (!unknown) {
        best = replay_class<CAP, TH>(L, T, mk_cls, n_mk, -1, wsum, bc);
        unknown = best.too_big != 0;
    }
    const RpOut big = best;
    const int events = best.n_events;
    pt4 = wall_clock64();
    if (tid == 0 && A.ln_prof) {
        int32_t* pp = A.ln_prof + (size_t)s * 8;
        pp[0] = (int)(pt1 - pt0), pp[1] = (int)(pt2 - pt1), pp[2] = (int)(pt3 - pt2), pp[3] = (int)(pt4 - pt3);
        pp[4] = prof_cand, pp[5] = prof_mk, pp[6] = L.n_irr, pp[7] = CAP;
        if (A.ln_prof2) {
            int32_t* p2 = A.ln_prof2 + (size_t)s * 8;
            p2[0] = big.nodes, p2[1] = big.rounds, p2[2] = big.c_build, p2[3] = big.c_jacobi, p2[4] = big.c_open, p2[5] = big.c_cc, p2[6] = big.c_walk, p2[7] = big.n_events;
        }
    }
    if (tid == 0) {
        if (unknown) {
            if (redo) {  // a larger table may hold it
                int32_t* list = (redo_big && best.nodes > big_from) ? redo_big : redo;
                list[atomicAdd(list + A.n_scans, 1)] = s;
                outp[0] = outp[1] = -1, outp[2] = 3, outp[3] = events;
            } else {
                outp[0] = outp[1] = -1, outp[2] = 1, outp[3] = events;
                atomicAdd(&A.ln_stats[0], 1);
            }
        } else {
            outp[0] = best.fin ? best.canon : -1;
            outp[1] = best.fin ? best.slot : -1;
            outp[2] = 0;
            outp[3] = events;
        }
    }
}


template <int CAP, int TH, int MODE, bool GT = false>
__global__ __launch_bounds__(TH) void k_cc_lastname(DevParams P, Arena A, const int32_t* todo, const int32_t* n_todo, int32_t* redo, int32_t* redo_big, int big_from,
                                                    int32_t* n_redo) {
    extern __shared__ __align__(16) unsigned char ln_kernel_smem[];
    int s = blockIdx.x;
    if (todo) {
        if (s >= *n_todo) return;
        s = todo[s];
    }
    ln_scan<CAP, TH, MODE, GT>(P, A, s, ln_kernel_smem, redo, redo_big, big_from);
}

}  // namespace

constexpr int kLnCapTiny = kLnCapTinyNodes, kLnCapSmall = kLnCapSmallNodes, kLnCapBig = kLnCapBigNodes;

// A triage pass over every scan (the set of classes to follow, or the answer when there is nothing to follow), then the three
// passes that follow the sets -- tables of 320 / 1792 / 8192 nodes, each over the scans the triage listed for it on the device
// (the grid stays the batch, idle workgroups leave at once: nothing is read back on the host).  Each of the three is as long as
// its slowest scan, so they run side by side on `st`, `st2`, `st3` when those are given.
void launch_lastname(const DevParams& P, const Arena& A, hipStream_t st, hipStream_t st2, hipStream_t st3, hipEvent_t ev_fork, hipEvent_t ev_join2,
                     hipEvent_t ev_join3, TimerHook th, void* tu) {
    const int B = A.n_scans;
    if (B <= 0 || A.max_scan_pts <= 0) return;
    {  // (function attributes are per device: set on every launch, like launch_track_chain does)
        hipFuncSetAttribute((const void*)k_cc_lastname<kLnCapTiny, 256, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ln_lds_bytes<kLnCapTiny>());
        hipFuncSetAttribute((const void*)k_cc_lastname<kLnCapSmall, 1024, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ln_lds_bytes<kLnCapSmall>());
        hipFuncSetAttribute((const void*)k_cc_lastname<kLnCapTiny, 256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ln_lds_bytes<kLnCapTiny>());
        hipFuncSetAttribute((const void*)k_cc_lastname<kLnCapSmall, 256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ln_lds_bytes<kLnCapSmall>());
        hipFuncSetAttribute((const void*)k_cc_lastname<kLnCapBig, 1024, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ln_lds_bytes<kLnCapBig>());
        hipFuncSetAttribute((const void*)k_cc_lastname<kLnCapHugeNodes, 1024, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ln_lds_bytes<kLnCapBig>());
    }
    int32_t* lists = A.cc_redo;  // 4 x ([B] scans, [B] = how many)
    for (int k = 0; k < 4; ++k) hipMemsetAsync(lists + (size_t)k * (B + 1) + B, 0, sizeof(int32_t), st);
    hipMemsetAsync(A.ln_stats, 0, 4 * sizeof(int32_t), st);
    const bool large = A.max_scan_pts > 200000;  // (128-beam class, 262 144 returns per scan: its tables are passed over by 1024 threads)
    if (th) th(tu, "cc_lastname", 1);
    if (!large)
        hipLaunchKernelGGL((k_cc_lastname<kLnCapTiny, 256, 1>), dim3(B), dim3(256), ln_lds_bytes<kLnCapTiny>(), st, P, A, (const int32_t*)nullptr,
                           (const int32_t*)nullptr, lists, (int32_t*)nullptr, 0, (int32_t*)nullptr);
    else  // (LDS sized like the mid tables: the certificate blocks of 16 waves need the room)
        hipLaunchKernelGGL((k_cc_lastname<kLnCapSmall, 1024, 1>), dim3(B), dim3(1024), ln_lds_bytes<kLnCapSmall>(), st, P, A, (const int32_t*)nullptr,
                           (const int32_t*)nullptr, lists, (int32_t*)nullptr, 0, (int32_t*)nullptr);
    if (th) th(tu, "cc_lastname", 0);
    const bool side = st2 && st3 && ev_fork && ev_join2 && ev_join3 && !th;  // (timed runs keep one stream: the hook records on one)
    if (side) {
        hipEventRecord(ev_fork, st);
        hipStreamWaitEvent(st2, ev_fork, 0);
        hipStreamWaitEvent(st3, ev_fork, 0);
    }
    int32_t* l0 = lists, *l1 = lists + (B + 1), *l2 = lists + 2 * (size_t)(B + 1), *l3 = lists + 3 * (size_t)(B + 1);
    {  // sets of more than 8192 nodes (the facades of a 128-beam scan): tables in arena scratch, the longest pass first (an empty list: the workgroups leave at once)
        if (th) th(tu, "cc_lastname_huge", 1);
        hipLaunchKernelGGL((k_cc_lastname<kLnCapHugeNodes, 1024, 2, true>), dim3(B), dim3(1024), ln_lds_bytes<kLnCapBig>(), side ? st3 : st, P, A, (const int32_t*)l3,
                           (const int32_t*)(l3 + B), (int32_t*)nullptr, (int32_t*)nullptr, 0, (int32_t*)nullptr);
        if (th) th(tu, "cc_lastname_huge", 0);
    }
    if (th) th(tu, "cc_lastname_big", 1);
    hipLaunchKernelGGL((k_cc_lastname<kLnCapBig, 1024, 2>), dim3(B), dim3(1024), ln_lds_bytes<kLnCapBig>(), side ? (large ? st2 : st3) : st, P, A, (const int32_t*)l2,
                       (const int32_t*)(l2 + B), (int32_t*)nullptr, (int32_t*)nullptr, 0, (int32_t*)nullptr);
    if (th) th(tu, "cc_lastname_big", 0);
    if (th) th(tu, "cc_lastname_mid", 1);
    hipLaunchKernelGGL((k_cc_lastname<kLnCapSmall, 256, 2>), dim3(B), dim3(256), ln_lds_bytes<kLnCapSmall>(), side ? st2 : st, P, A, (const int32_t*)l1,
                       (const int32_t*)(l1 + B), (int32_t*)nullptr, (int32_t*)nullptr, 0, (int32_t*)nullptr);
    if (th) th(tu, "cc_lastname_mid", 0);
    if (th) th(tu, "cc_lastname_small", 1);
    hipLaunchKernelGGL((k_cc_lastname<kLnCapTiny, 256, 2>), dim3(B), dim3(256), ln_lds_bytes<kLnCapTiny>(), st, P, A, (const int32_t*)l0,
                       (const int32_t*)(l0 + B), (int32_t*)nullptr, (int32_t*)nullptr, 0, (int32_t*)nullptr);
    if (th) th(tu, "cc_lastname_small", 0);
    if (side) {
        hipEventRecord(ev_join2, st2);
        hipEventRecord(ev_join3, st3);
        hipStreamWaitEvent(st, ev_join2, 0);
        hipStreamWaitEvent(st, ev_join3, 0);
    }
}

}  // namespace scvod
