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
    int32_t pad0, pad1;
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
};

void launch_track_chain(const DevParams& P, const Arena& A, const TrackBatch& J, const ChainJob& C, int from_apri, hipStream_t st,
                        TimerHook th, void* tu);

}  // namespace scvod
#endif
