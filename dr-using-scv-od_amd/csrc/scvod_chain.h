// scvod_chain.h -- job description of the sequential tracking chain (scvod_chain.hip), shared with the C-ABI layer.
#ifndef SCVOD_CHAIN_H_
#define SCVOD_CHAIN_H_
#include "scvod_dev.h"

namespace scvod {

struct ChainWalker {      // one workgroup of k_tk_chain: steps [a, b) of a chain, warmed up from step t0 <= a
    int32_t first;        // offset of the chain's first frame in chain_scans
    int32_t n_frames;     // frames of the chain (steps: n_frames - 1)
    int32_t a, b, t0;
    int32_t chain;
    int32_t ext;          // 1: the chain's first walker, and the chain continues one that another shard walked: its steps [t0, a) are a
                          //    warm-up like any other walker's, the state it has to match at step a arrives later (scvod_batch_track_resume)
    int32_t pad1;
};

struct ChainWs {          // geometry of the walkers' workspace: walker w lives at base + w * stride
    unsigned char* base;
    size_t stride;
    size_t off_hdr, off_ent[3], off_parts[3], off_pool[3], off_chit, off_evr, off_suniq, off_spairs, off_vlab, off_lcnt, off_lfwd,
        off_newent, off_cmeta, off_cparts, off_links, off_dsz, off_eidx, off_rp;
    int32_t cap_pool;     // carried points a state can hold
    int32_t cap_ent;      // clusters (entries, parts, links, created labels) a state / step can hold
    int32_t cap_nv;       // voxels of a table
};

struct ChainJob {
    const int32_t* chain_scans;         // the frames (scan indices of the batch) of every chain, chain after chain
    const ChainWalker* walkers;         // [n_walkers], the walkers of a chain consecutive and in order
    const int32_t* chain_first_walker;  // [n_chains + 1]
    int32_t n_walkers, n_chains;
    ChainWs ws;
    int32_t* stats;       // [0] error bits (1 pool, 2 entries, 4 created labels, 8 bitset words), [1] segments walked again, [2] segments compared
    int32_t words;        // bitset words of an evaluating wave
    int32_t n_eval_waves;
    int32_t force_generic;  // testing: every step through the HBM-resident generic path
    const unsigned char* const* ext_state;  // [n_chains] resume: the state the chain's predecessor (on another shard) really ended in, or nullptr
    int32_t resume;       // k_tk_chain_fix: compare the first walker's warm-up snapshot with ext_state first, walk it again when they differ
                          //    (2: compare only -- count the chains that would be walked again in cmp_out, change nothing)
    int32_t* cmp_out;     // [1] resume == 2
    int32_t literal_max_name;  // 1: a frame's first new cluster re-uses Frame::max_name as ssc.cpp:354 stores it (Arena::cc_last)
    int32_t* step_ticks;  // [frames of all chains, parallel to chain_scans] 10 ns ticks k_tk_chain spent in step t of a chain as an OWN step (0: not
                          //    walked as one): the planner of the stream's next batch cuts the segments by these times; may be nullptr
};

#ifdef __HIPCC__
// Successor-table look-up behind the sampled keys (k_tk_probe, the chain's carried points): the slot among the records
// [a0, a1) whose key is `key` and whose label is set, -1 otherwise (ssc.cpp:1304-1305).  `first` = tab[a0], already loaded.
// A table sampled every record (a1 = a0 + 1: any 64-beam scan) is decided by `first`; a larger one (2^k records behind a
// sample: 16 for the 72 k voxels of a 128-beam scan) is bisected -- k dependent reads instead of 2^(k-1) on average.
__device__ __forceinline__ int tk_find_slot(const int4* __restrict__ tab, int a0, int a1, int4 first, int key) {
    if (first.x >= key) return (first.x == key && first.y != -1) ? a0 : -1;
    int l = a0 + 1, h = a1;  // first record behind a0 whose key is >= key
    while (l < h) {
        const int mid = (l + h) >> 1;
        if (tab[mid].x < key)
            l = mid + 1;
        else
            h = mid;
    }
    if (l >= a1) return -1;
    const int4 rec = tab[l];
    return (rec.x == key && rec.y != -1) ? l : -1;
}
#endif

void launch_track_chain(const DevParams& P, const Arena& A, const TrackBatch& J, const ChainJob& C, int from_apri, hipStream_t st,
                        TimerHook th, void* tu);
// the verification pass again, first walkers against the states of ChainJob::ext_state (C.resume = 1)
void launch_track_chain_resume(const DevParams& P, const Arena& A, const TrackBatch& J, const ChainJob& C, int from_apri, hipStream_t st);
// a chain's boundary state as one record: int32 {entries, carried points, parts, valid}, then entries (2 int4 each), parts (int32,
// padded to 16 bytes), carried points (float4).  which = 0: what the chain's first walker assumed at its first own step (its
// warm-up snapshot), 1: what the chain's last walker ended in.
void launch_chain_export_state(const ChainJob& C, int chain, int which, unsigned char* dst, long long cap_bytes, hipStream_t st);
size_t chain_state_bytes(const ChainWs& ws);

}  // namespace scvod
#endif
