// scvod_capi.hip -- the extern "C" boundary of libscvod.so (declared in include/scvod.h).
// Owns the device arena, host staging and the HIP stream; no torch types, no exceptions
// across the boundary.  GPU-only: every entry point fails loudly without a HIP device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/scvod.h"
#include <algorithm>
#include "scvod_kernels.h"
#include "scvod_chain.h"

using namespace scvod;

namespace {

struct TimingEntry {
    const char* name;
    hipEvent_t e0, e1;
};

}  // namespace

struct UploadSlot {
    void* pinned = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool used = false;
};

struct scvod_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    scvod_params params;
    scvod_pw_params pw;
    DevParams dev;
    int64_t cap_pts = 0;
    int32_t cap_scans = 0;
    void* arena_base = nullptr;
    size_t arena_bytes = 0;
    Arena A;             // carved pointers (pts / scan_off filled per call)
    float4* d_in = nullptr;        // ctx-owned input buffer for the per-scan host API
    int32_t* d_scan_off = nullptr; // [cap_scans+1]
    // tracking buffers
    float4* t_pts = nullptr;
    int32_t* t_hit = nullptr;
    uint64_t* t_work = nullptr;
    int32_t* t_uniq = nullptr;
    int32_t* t_begin = nullptr;
    int32_t* t_pair = nullptr;
    int32_t* t_pairpt = nullptr;
    uint16_t* d_vb_lut = nullptr;   // A.vb_lut is set only for the duration of a VoxelGrid run
    uint32_t* d_labels = nullptr;   // staging of the host API's label array (scvod_voxelgrid): side buffer
    scvod_apri* d_apri = nullptr;   // PointAPRI records: side buffer (Arena::apri)
    size_t side_bytes = 0;          // what ensure_side allocated so far (counted in scvod_arena_bytes)
    int32_t* t_count = nullptr;
    float* t_T = nullptr;
    int32_t* d_next_scan = nullptr;   // [cap_scans] successor table of scvod_batch_track
    const int4** d_ext = nullptr;     // [cap_scans] external (boundary) tables
    std::vector<float> up_T;          // what the device copies of T / next_scan / ext currently hold (re-uploaded on change only)
    std::vector<int32_t> up_next;
    std::vector<const void*> up_ext;
    UploadSlot up_ring[8];                   // (a chain-mode scvod_batch_track can upload six changed tables in one call: ADVICE r3)
    int n_cu = 256;                          // compute units of the device (hipDeviceProp: the chain plans one segment per CU)
    int up_next_slot = 0;
    bool track_valid = false;
    // sequential tracking chain (scvod_chain.hip)
    int track_mode = SCVOD_TRACK_CHAIN;
    int chain_seg = 0, chain_warm = -1;      // steps per segment (0: from the job, ~250 segments), warm-up steps in front of it (-1: chosen per stream, below)
    // warm-up chosen per job: 10 steps to begin with; when a batch walked more than a few segments again (its warm-up did not
    // reproduce their start states) the next batch of the stream warms up 2 steps longer, up to 16.  The counters come back through
    // a pinned word and an event: never a stream drain.
    int chain_warm_auto = 10, chain_warm_used = 10;
    int32_t* h_chain_fb = nullptr;           // pinned [8]
    hipEvent_t chain_fb_ev = nullptr;
    bool chain_fb_pending = false;
    int chain_fb_segments = 0;
    int chain_seg_used = 0;
    bool chain_generic = false;              // testing: every step through the generic (HBM-resident) step function
    bool max_name_literal = true;            // ssc.cpp:354: a frame's first new cluster re-uses the last running number (scvod_lastname.hip)
    bool last_name_valid = false;
    int32_t* d_chain_scans = nullptr;        // [cap_scans]
    ChainWalker* d_chain_walkers = nullptr;  // [cap_scans]
    int32_t* d_chain_fw = nullptr;           // [cap_scans + 1]
    int32_t* d_chain_stats = nullptr;        // [8]
    int32_t* d_step_ticks = nullptr;         // [cap_scans] per frame of the chain plan: what k_tk_chain spent in the step that starts there (ChainJob::step_ticks)
    int32_t* h_step_ticks = nullptr;         // pinned copy of the last batch's, behind step_fb_ev
    hipEvent_t step_fb_ev = nullptr;
    bool step_fb_pending = false;
    std::vector<int32_t> step_fb_scans;      // the plan (frames of every chain) the ticks in flight belong to
    std::vector<int32_t> step_ticks, step_ticks_scans;  // the last ticks that arrived and their plan
    std::vector<int32_t> plan_scans_tmp;
    std::vector<std::vector<int>> cuts_cache;  // the time-balanced segment starts of the plan below
    std::vector<int32_t> cuts_scans;
    std::vector<int> cuts_chain_first, cuts_chain_len;  // (two successor tables can flatten to the same frames and still split them differently)
    std::vector<uint8_t> cuts_halo;
    int cuts_warm = -1;
    int chain_balance = 1;                   // 1: cut the segments of a stream's next batch by these times (scvod_set_track_mode: segment_steps 0)
    int32_t* d_chain_cmp = nullptr;          // [1] scvod_batch_track_compare
    std::vector<uint8_t> halo;               // per scan of a batch: 1 = halo (warmed up over, never decided): scvod_set_track_owned / _halo
    ChainJob last_cj;                        // the chain job of the last scvod_batch_track (export / resume)
    TrackBatch last_tb;
    const unsigned char** d_ext_state = nullptr;  // [cap_scans] device array of the states a resume compares with
    void** h_ext_state_pin = nullptr;        // pinned staging of that table (scvod_batch_track_compare_device)
    bool ext_state_pin_valid = false;
    int ext_state_pin_n = 0;
    std::vector<int32_t> up_chain_scans, up_chain_fw, up_chain_walkers;
    void* chain_ws = nullptr;                // walkers' workspace (own allocation, grows on demand)
    size_t chain_ws_bytes = 0;
    ChainWs chain_geom;
    int chain_geom_pts = -1;                 // max_scan_pts the geometry was laid out for
    int64_t chain_pool_points = 0, chain_geom_pool = -1;  // appended-cloud capacity of a walker (0: 8 x the largest scan)
    bool chain_ran = false;                  // the last scvod_batch_track ran the chain (stats are meaningful)
    bool chain_ws_clean = false;
    int chain_ws_layout_pts = -1;
    bool tables_valid = false;   // successor tables (vox_track) built for the current clustering
    std::vector<int32_t> tk_stage;    // host staging of scvod_batch_fetch_track
    // streaming ingest (scvod_sequence_ingest): two device chunk buffers, a copy stream, pinned offsets
    hipStream_t copy_stream = nullptr;
    // the max_name pass (scvod_lastname.hip) runs beside the kernels that follow the clustering, on a stream of its own; whoever
    // needs its result or its scratch waits for ln_done (join_lastname)
    hipStream_t ln_stream = nullptr;
    hipStream_t ln_stream2 = nullptr, ln_stream3 = nullptr;
    hipEvent_t ln_fork = nullptr, ln_done = nullptr, ln_fork2 = nullptr, ln_join2 = nullptr, ln_join3 = nullptr;
    bool ln_pending = false;
    void* ingest_buf[2] = {nullptr, nullptr};
    size_t ingest_cap = 0;  // points per buffer
    hipEvent_t ingest_copied[2] = {nullptr, nullptr}, ingest_done[2] = {nullptr, nullptr};
    int32_t* ingest_off = nullptr;  // pinned
    size_t ingest_off_cap = 0;
    std::vector<uint8_t> tk_stage_dyn;
    // last batch
    bool batch_valid = false;
    std::vector<int32_t> h_scan_off;
    std::vector<int32_t> h_counts;
    bool counts_valid = false;
    bool have_patchwork = false;
    int batch_mode = 0;          // do_patchwork of the last batch: 1 Patchwork+binning, 0 binning only, 2 caller's apri_vec
    bool apri_compact = false;   // PointAPRI records not materialised yet (k_apri_expand on request)
    bool voxels_valid = false;   // the last batch ran the voxel stage in descriptor mode (not VoxelGrid)
    bool clusters_valid = false;
    bool clustered_this_batch = false;       // the batch's pt_mapcls bytes carry car / dynamic marks of an earlier clustering (cleared before the next one)
    bool types_valid = false;
    hipStream_t last_stream = nullptr;
    // host staging for scvod_scan_result
    void* nn_buf[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // scvod_nn_search scratch (grow-only)
    size_t nn_cap[6] = {0, 0, 0, 0, 0, 0};
    void* stage = nullptr;       // pinned host block holding the arrays of the last scvod_scan_result
    size_t stage_bytes = 0;
    // timing
    bool timing = false;
    std::vector<TimingEntry> tim;
    size_t tim_used = 0;
    std::string err;
};

namespace {

int fail(scvod_ctx* c, int code, const char* fmt, ...) {
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) return fail(ctx, SCVOD_ERR_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Carver {
    unsigned char* base;
    size_t off = 0;
    template <typename T>
    T* take(size_t count) {
        off = align_up(off, 256);
        T* p = base ? (T*)(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

void carve(scvod_ctx* c, unsigned char* base, size_t* total) {
    const size_t N = (size_t)c->cap_pts, B = (size_t)c->cap_scans;
    Carver k{base};
    Arena& A = c->A;
    c->d_scan_off = k.take<int32_t>(B + 1);
    A.pid = k.take<int16_t>(N);
    A.keys = k.take<uint64_t>(N);
    A.seg = k.take<uint32_t>(N);
    A.sorted_xyz = k.take<Xyz>(N + 8);
    A.sorted_idx = k.take<uint32_t>(N);
    A.zkey = k.take<uint32_t>(N);
    A.fit_thd = k.take<float>(B * kMaxPatches);
    A.order = k.take<int4>(B * kMaxPatches);
    A.order_hist = k.take<int32_t>(64);
    A.order_cursor = k.take<int32_t>(64);
    A.order_off = k.take<int32_t>(65);
    A.patch_count = k.take<int32_t>(B * kMaxPatches);
    A.patch_cursor = k.take<int32_t>(B * kMaxPatches);
    A.patch_off = k.take<int32_t>(B * (kMaxPatches + 1));
    A.patch_rec = k.take<PatchRec>(B * kMaxPatches);
    A.planes = k.take<scvod_patch_plane>(B * kMaxPatches);
    A.emit_off = k.take<int32_t>(B * kMaxPatches * 4);
    A.cls = k.take<uint8_t>(N);
    A.ground_idx = k.take<int32_t>(N);
    A.nonground_idx = k.take<int32_t>(N);
    A.apri = c->d_apri;  // (side buffer: ensure_side; nullptr until something asks for PointAPRI records)
    A.apri_src = k.take<int32_t>(N);
    A.apri_key = k.take<int32_t>(N);
    A.apri_int = k.take<float>(N);
    A.apri_idx3 = k.take<int32_t>(N);
    A.rejected_src = k.take<int32_t>(N);
    A.counts = k.take<int32_t>(B * 8);
    A.scan_irr = k.take<int32_t>(B);
    A.irr_list = k.take<int32_t>(B * (kIrrListCap + 1));
    A.cc_perm = k.take<int32_t>(B);
    A.vb_count = k.take<int32_t>(B * kMaxBuckets);
    A.vb_off = k.take<int32_t>(B * (kMaxBuckets + 1));
    A.vb_nvox = k.take<int32_t>(B * kMaxBuckets);
    A.vox_off = k.take<int32_t>(B * (kMaxBuckets + 1));
    A.vorder = k.take<int4>(B * kMaxBuckets);
    A.vorder_hist = k.take<int32_t>(64);
    A.vorder_cursor = k.take<int32_t>(64);
    A.vorder_off = k.take<int32_t>(65);
    A.vkeys = k.take<uint64_t>(N);
    A.tmp_vox_key = k.take<int32_t>(N);
    A.tmp_vox_begin = k.take<int32_t>(N);
    A.tmp_vox_av = k.take<float>(N);
    A.tmp_vox_cov = k.take<float>(N);
    A.vox_key = k.take<int32_t>(N);
    A.vox_pt_begin = k.take<int32_t>(N + B);
    A.vox_pts = k.take<int32_t>(N);
    A.vox_av = k.take<float>(N);
    A.vox_cov = k.take<float>(N);
    A.cc_stats = k.take<int32_t>(8);
    A.cc_again = k.take<int32_t>(B + 1);
    A.cc_help = k.take<int32_t>(kCcHelpWords);
    A.cc_help_blocks = 0;
    A.cc_help_blocks_wanted = 1;
    A.cc_exact_max = 0x3fffffff;  // (round 6: exact whatever the component's size is the default -- the passes are shared with helper blocks, k_cc_exact)
    A.cc_plain_rule = 1;
    A.cc_parent = k.take<int32_t>(N);
    A.cc_touched = k.take<uint8_t>(N);
    A.pt_voxel = k.take<int32_t>(N);
    A.pt_cluster = k.take<int32_t>(N);
    A.cl_bbox = k.take<uint32_t>(7 * N + 64);
    A.cl_count = k.take<int32_t>(N);
    A.pt_type = k.take<uint8_t>(N);
    A.cc_last = k.take<int32_t>(B * 4);
    A.cc_redo = k.take<int32_t>(4 * (B + 1));
    A.ln_state = k.take<int32_t>(B * 2048);
    A.ln_stats = k.take<int32_t>(4);
    A.ln_prof = k.take<int32_t>(B * 8);
    A.ln_prof2 = k.take<int32_t>(B * 8);
    A.vg_par = k.take<int32_t>(B * 16);
    A.vg_range = k.take<int32_t>(2);  // [0] largest cell-index range of the batch, [1] bin shift of the bucket table
    A.vg_outoff = k.take<int32_t>(B + 1);
    c->d_vb_lut = k.take<uint16_t>(B * kVgLutBins);
    A.vb_lut = nullptr;  // hot path: uniform key ranges (DevParams::vb_shift)
    A.vb_lut_shift = nullptr;
    A.vg_labels = nullptr;
    A.vg_max_intensity = 1.f;
    c->t_hit = k.take<int32_t>(N);
    c->t_work = k.take<uint64_t>(N);
    c->t_uniq = k.take<int32_t>(N);
    c->t_begin = k.take<int32_t>(N + 1);
    c->t_pair = k.take<int32_t>(N);
    c->t_pairpt = k.take<int32_t>(B + 1);
    c->t_count = k.take<int32_t>(N);
    c->t_T = k.take<float>(12 * (B + 1));
    // sequence differencing on the device: the per-pair scratch above is never in use at the same time
    A.tk_hit = c->t_hit;
    A.tk_uniq = c->t_uniq;
    A.tk_nuniq = c->t_count;
    A.tk_cursor = c->t_pair;
    A.tk_mbegin = k.take<int32_t>(N + 1);  // (published by the clustering: not shared with the one-shot probe's scratch)
    A.vox_track = k.take<int4>(N);
    A.vox_rep = k.take<int32_t>(N);
    A.tk_crep = k.take<int32_t>(N);
    A.tk_prep = k.take<int32_t>(N);
    A.cl_state = k.take<int8_t>(N);
    A.tk_members = k.take<int32_t>(N);
    A.tk_pairs = k.take<int2>(N);
    A.tk_npairs = k.take<int32_t>(N);
    A.tk_clusters = k.take<int32_t>(N);
    A.tk_scan = k.take<int32_t>(B * 4);
    A.pt_dyn = k.take<uint8_t>(N);
    A.pt_mapcls = k.take<uint8_t>(N);
    c->d_next_scan = k.take<int32_t>(B);
    c->d_ext = k.take<const int4*>(B);
    c->d_chain_scans = k.take<int32_t>(B);
    c->d_chain_walkers = k.take<ChainWalker>(B);
    c->d_chain_fw = k.take<int32_t>(B + 1);
    c->d_chain_stats = k.take<int32_t>(8);
    c->d_step_ticks = k.take<int32_t>(B + 1);
    c->d_chain_cmp = k.take<int32_t>(4);
    c->d_ext_state = k.take<const unsigned char*>(B);
    *total = align_up(k.off, 256);
}

void build_dev_params(scvod_ctx* c) {
    const scvod_params& p = c->params;
    const scvod_pw_params& w = c->pw;
    DevParams& D = c->dev;
    memset(&D, 0, sizeof(D));
    BinParams& b = D.bin;
    b.min_dis = p.min_dis;
    b.max_dis = p.max_dis;
    b.min_angle = p.min_angle;
    b.max_angle = p.max_angle;
    b.min_azimuth = p.min_azimuth;
    b.max_azimuth = p.max_azimuth;
    b.range_res = p.range_res;
    b.sector_res = p.sector_res;
    b.azimuth_res = p.azimuth_res;
    scvod_grid_dims(&p, &b.range_num, &b.sector_num, &b.azimuth_num, &b.bin_num);
    D.keep = keep_fast_of(b);
    D.binfast = bin_fast_of(b);
    CzmParams& z = D.czm;
    z.min_range = w.min_range;
    z.max_range = w.max_range;
    // patchwork.h:83-94
    const double z2 = (7 * w.min_range + w.max_range) / 8.0;
    const double z3 = (3 * w.min_range + w.max_range) / 4.0;
    const double z4 = (w.min_range + w.max_range) / 2.0;
    z.zone_min[0] = w.min_range;
    z.zone_min[1] = z2;
    z.zone_min[2] = z3;
    z.zone_min[3] = z4;
    z.ring_size[0] = (z2 - w.min_range) / w.num_rings_each_zone[0];
    z.ring_size[1] = (z3 - z2) / w.num_rings_each_zone[1];
    z.ring_size[2] = (z4 - z3) / w.num_rings_each_zone[2];
    z.ring_size[3] = (w.max_range - z4) / w.num_rings_each_zone[3];
    int base = 0;
    for (int k = 0; k < 4; ++k) {
        z.sector_size[k] = 2 * M_PI / w.num_sectors_each_zone[k];
        z.num_rings[k] = w.num_rings_each_zone[k];
        z.num_sectors[k] = w.num_sectors_each_zone[k];
        z.patch_base[k] = base;
        base += w.num_rings_each_zone[k] * w.num_sectors_each_zone[k];
        z.elevation_thr[k] = w.elevation_thr[k];
        z.flatness_thr[k] = w.flatness_thr[k];
    }
    z.num_patches = base;
    const double sensor_height = (double)p.sensor_height;  // set_sensor(const double&), ssc.cpp:93
    z.z_cut = -1.8 * sensor_height;
    z.seed_margin_z = w.adaptive_seed_selection_margin * sensor_height;
    z.th_seeds = w.th_seeds;
    z.th_dist = w.th_dist;
    z.uprightness_thr = w.uprightness_thr;
    z.num_iter = w.num_iter;
    z.num_lpr = w.num_lpr;
    z.num_min_pts = w.num_min_pts;
    z.num_rings_of_interest = w.num_rings_of_interest;
    czm_finalize(z);
    D.n_patches = base;
    D.max_z = p.max_z;
    D.min_z = p.min_z;
    D.car_square = p.car_square;
    D.to_be_class = p.toBeClass;
    // voxel buckets: keys of filtered points lie in [-(R*S+S+1), bin_num)
    D.key_off = (int64_t)b.range_num * b.sector_num + b.sector_num + 1;
    int64_t range = (int64_t)b.bin_num + D.key_off + 1;
    int shift = 12;
    while (((range >> shift) + 1) > kMaxBuckets) ++shift;
    D.vb_shift = shift;
    D.n_buckets = (int32_t)((range >> shift) + 1);
}

void timer_hook(void* user, const char* name, int begin) {
    scvod_ctx* c = (scvod_ctx*)user;
    static const bool trace = getenv("SCVOD_TRACE") != nullptr;
    if (trace) {  // debugging aid: name every launch and synchronise after it
        if (!begin) {
            hipError_t e = hipStreamSynchronize(c->last_stream);
            fprintf(stderr, "[scvod] %s done: %s\n", name, hipGetErrorString(e));
        } else {
            fprintf(stderr, "[scvod] %s ...\n", name);
        }
        fflush(stderr);
    }
    if (!c->timing) return;
    hipStream_t st = c->last_stream;
    if (begin) {
        if (c->tim_used == c->tim.size()) {
            TimingEntry t;
            t.name = name;
            hipEventCreate(&t.e0);
            hipEventCreate(&t.e1);
            c->tim.push_back(t);
        }
        c->tim[c->tim_used].name = name;
        hipEventRecord(c->tim[c->tim_used].e0, st);
    } else {
        hipEventRecord(c->tim[c->tim_used].e1, st);
        c->tim_used++;
    }
}

// Makes `st` wait for the max_name pass of the last clustering (if one is still in flight).
static void join_lastname(scvod_ctx* c, hipStream_t st) {
    if (!c->ln_pending) return;
    hipStreamWaitEvent(st, c->ln_done, 0);
    c->ln_pending = false;
}

// Buffers that only the per-scan host API, the one-shot probe, the VoxelGrid run and a fetch of PointAPRI records touch: 80 of
// what used to be 318 bytes per point of the arena, allocated on first use (round 5: a sequence shard that stays in the batch
// API never pays for them -- 26 GB of the seq-05 job's 104).
enum SideBuf { kSideIn = 1, kSideTpts = 2, kSideLabels = 4, kSideApri = 8 };
int ensure_side(scvod_ctx* c, int which) {
    const size_t N = (size_t)c->cap_pts;
    auto need = [&](void** p, size_t bytes) -> int {
        if (*p) return SCVOD_OK;
        HIPCHK(c, hipMalloc(p, bytes));
        c->side_bytes += bytes;
        return SCVOD_OK;
    };
    int rc = SCVOD_OK;
    if ((which & kSideIn) && (rc = need((void**)&c->d_in, sizeof(float4) * N))) return rc;
    if ((which & kSideTpts) && (rc = need((void**)&c->t_pts, sizeof(float4) * N))) return rc;
    if ((which & kSideLabels) && (rc = need((void**)&c->d_labels, sizeof(uint32_t) * N))) return rc;
    if ((which & kSideApri) && (rc = need((void**)&c->d_apri, sizeof(scvod_apri) * N))) return rc;
    c->A.apri = c->d_apri;
    return rc;
}

int run_batch(scvod_ctx* c, const void* d_xyzi, const int32_t* h_off, int32_t n_scans, hipStream_t st,
              int do_patchwork, int apply_filter, int do_voxels, int sync, bool off_pinned = false) {
    if (!c) return SCVOD_ERR_INVALID;
    if (n_scans <= 0 || !h_off || (!d_xyzi && do_patchwork != 2)) return fail(c, SCVOD_ERR_INVALID, "empty batch");
    if (n_scans > c->cap_scans) return fail(c, SCVOD_ERR_CAPACITY, "n_scans %d > capacity %d", n_scans, c->cap_scans);
    int64_t total = (int64_t)h_off[n_scans] - h_off[0];
    if (h_off[0] != 0) return fail(c, SCVOD_ERR_INVALID, "scan_offsets[0] must be 0");
    if (total > c->cap_pts) return fail(c, SCVOD_ERR_CAPACITY, "%lld points > capacity %lld", (long long)total, (long long)c->cap_pts);
    int32_t mx = 0;
    for (int s = 0; s < n_scans; ++s) {
        int32_t n = h_off[s + 1] - h_off[s];
        if (n < 0) return fail(c, SCVOD_ERR_INVALID, "scan_offsets not monotone");
        if (n > mx) mx = n;
    }
    if (mx > SCVOD_MAX_SCAN_POINTS)
        return fail(c, SCVOD_ERR_CAPACITY, "scan of %d points > SCVOD_MAX_SCAN_POINTS (%d)", mx, SCVOD_MAX_SCAN_POINTS);
    HIPCHK(c, hipSetDevice(c->device));
    if (!st) st = c->stream;
    c->last_stream = st;
    join_lastname(c, st);
    c->h_scan_off.assign(h_off, h_off + n_scans + 1);
    if (do_patchwork != 1) {  // (binning without Patchwork writes the PointAPRI records, a caller's apri_vec lives in them)
        const int rc_side = ensure_side(c, kSideApri);
        if (rc_side) return rc_side;
    }
    // (a caller-pinned offset table is read by the copy engine directly: nothing to wait for on the host)
    HIPCHK(c, hipMemcpyAsync(c->d_scan_off, off_pinned ? h_off : c->h_scan_off.data(), sizeof(int32_t) * (n_scans + 1), hipMemcpyHostToDevice, st));
    c->A.pts = (const float4*)d_xyzi;
    c->A.scan_off = c->d_scan_off;
    c->A.n_scans = n_scans;
    c->A.max_scan_pts = mx;
    c->A.total_pts = total;
    {   // 32-bit voxel sort keys when a bucket-relative key and a scan-local index fit together (filtered keys never leave the grid)
        int ib = 1;
        while ((1 << ib) < mx) ++ib;
        c->A.vx_idx_bits = ib;
        c->A.vx_k32 = (do_patchwork != 3 && apply_filter && c->dev.vb_shift + ib <= 32) ? 1 : 0;
    }
    c->tim_used = 0;
    c->batch_valid = false;
    c->counts_valid = false;
    c->voxels_valid = false;
    c->clusters_valid = false;
    c->types_valid = false;
    c->track_valid = false;
    c->tables_valid = false;
    // marks per input point for the static map (car-cluster member / dynamic / list marks): clean for every new batch
    if (mx > 0 && do_patchwork != 2 && do_patchwork != 3) HIPCHK(c, hipMemsetAsync(c->A.pt_mapcls, 0, (size_t)total, st));
    c->clustered_this_batch = false;
    if (mx > 0) launch_process(c->dev, c->A, st, do_patchwork, apply_filter, do_voxels, timer_hook, c);
    HIPCHK(c, hipGetLastError());
    if (mx == 0) {  // every scan empty: no kernel runs (neither here nor in the clustering / tracking launches): clean per-scan words
        HIPCHK(c, hipMemsetAsync(c->A.counts, 0, sizeof(int32_t) * 8 * n_scans, st));
        HIPCHK(c, hipMemsetAsync(c->A.tk_scan, 0, sizeof(int32_t) * 4 * n_scans, st));
    }
    c->batch_valid = true;
    c->voxels_valid = (do_voxels != 0);
    c->have_patchwork = (do_patchwork == 1);
    c->batch_mode = do_patchwork;
    c->apri_compact = (do_patchwork == 1);
    if (sync) HIPCHK(c, hipStreamSynchronize(st));
    return SCVOD_OK;
}

int ensure_counts(scvod_ctx* c) {
    if (!c->batch_valid) return fail(c, SCVOD_ERR_STATE, "no batch has been run");
    if (c->counts_valid) return SCVOD_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    c->h_counts.resize((size_t)c->A.n_scans * 8);
    HIPCHK(c, hipMemcpy(c->h_counts.data(), c->A.counts, sizeof(int32_t) * c->h_counts.size(), hipMemcpyDeviceToHost));
    c->counts_valid = true;
    return SCVOD_OK;
}

int fetch_scan(scvod_ctx* c, int32_t s, scvod_scan_result* out) {
    int rc = ensure_counts(c);
    if (rc) return rc;
    if (s < 0 || s >= c->A.n_scans) return fail(c, SCVOD_ERR_INVALID, "scan %d out of range", s);
    const int32_t* k = &c->h_counts[(size_t)s * 8];
    const size_t base = (size_t)c->h_scan_off[s];
    memset(out, 0, sizeof(*out));
    out->n_points = k[0];
    out->n_ground = k[1];
    out->n_nonground = k[2];
    out->n_dropped = k[3];
    out->n_apri = k[4];
    out->n_rejected = k[5];
    out->n_voxels = k[6];
    out->n_patches = k[7];
    if (c->apri_compact && k[4] > 0) {
        const int rc_side = ensure_side(c, kSideApri);
        if (rc_side) return rc_side;
    }
    const Arena& A = c->A;
    hipStream_t st = c->last_stream;
    if (c->have_patchwork) launch_cls(A, s, base, k[0], st);
    if (c->apri_compact && k[4] > 0) launch_apri_expand(c->dev, A, s, 1, k[4], st);  // PointAPRI records of this scan only
    HIPCHK(c, hipGetLastError());
    // one pinned staging block, twelve asynchronous copies, one synchronisation (the pointers handed out live in it)
    struct Part {
        const void* src;
        size_t bytes;
        size_t at;
    };
    const size_t n_cls = c->have_patchwork ? (size_t)k[0] : 0;
    Part parts[12] = {{A.cls + base, n_cls, 0},
                      {A.ground_idx + base, 4 * (size_t)k[1], 0},
                      {A.nonground_idx + base, 4 * (size_t)k[2], 0},
                      {A.planes + (size_t)s * kMaxPatches, sizeof(scvod_patch_plane) * (size_t)k[7], 0},
                      {A.apri ? A.apri + base : nullptr, A.apri ? sizeof(scvod_apri) * (size_t)k[4] : 0, 0},
                      {A.apri_src + base, 4 * (size_t)k[4], 0},
                      {A.rejected_src + base, 4 * (size_t)k[5], 0},
                      {A.vox_key + base, 4 * (size_t)k[6], 0},
                      {A.vox_pt_begin + base + s, 4 * ((size_t)k[6] + 1), 0},
                      {A.vox_pts + base, 4 * (size_t)k[4], 0},
                      {A.vox_av + base, 4 * (size_t)k[6], 0},
                      {A.vox_cov + base, 4 * (size_t)k[6], 0}};
    size_t need = 0;
    for (Part& p : parts) {
        p.at = need;
        need += (p.bytes + 63) & ~(size_t)63;
    }
    need += 64;
    if (need > c->stage_bytes) {
        if (c->stage) hipHostFree(c->stage);
        c->stage = nullptr;
        c->stage_bytes = 0;
        HIPCHK(c, hipHostMalloc(&c->stage, need + need / 4, hipHostMallocDefault));
        c->stage_bytes = need + need / 4;
    }
    char* h = (char*)c->stage;
    for (const Part& p : parts)
        if (p.bytes) HIPCHK(c, hipMemcpyAsync(h + p.at, p.src, p.bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (k[6] == 0) *(int32_t*)(h + parts[8].at) = 0;
    out->cls = (const uint8_t*)(h + parts[0].at);
    out->ground_idx = (const int32_t*)(h + parts[1].at);
    out->nonground_idx = (const int32_t*)(h + parts[2].at);
    out->planes = (const scvod_patch_plane*)(h + parts[3].at);
    out->apri = (const scvod_apri*)(h + parts[4].at);
    out->apri_src = (const int32_t*)(h + parts[5].at);
    out->rejected_src = (const int32_t*)(h + parts[6].at);
    out->vox_key = (const int32_t*)(h + parts[7].at);
    out->vox_pt_begin = (const int32_t*)(h + parts[8].at);
    out->vox_pts = (const int32_t*)(h + parts[9].at);
    out->vox_av = (const float*)(h + parts[10].at);
    out->vox_cov = (const float*)(h + parts[11].at);
    return SCVOD_OK;
}

int host_scan(scvod_ctx* c, const float* h_xyzi, int32_t n, int do_pw, int filt, int do_vox, scvod_scan_result* out) {
    if (!c || !out || n < 0 || (n > 0 && !h_xyzi)) return fail(c, SCVOD_ERR_INVALID, "bad arguments");
    if (n > c->cap_pts) return fail(c, SCVOD_ERR_CAPACITY, "%d points > capacity %lld", n, (long long)c->cap_pts);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc_side = ensure_side(c, kSideIn)) return rc_side;
    if (n) HIPCHK(c, hipMemcpyAsync(c->d_in, h_xyzi, sizeof(float) * 4 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    int32_t off[2] = {0, n};
    int rc = run_batch(c, c->d_in, off, 1, c->stream, do_pw, filt, do_vox, 1);
    if (rc) return rc;
    rc = fetch_scan(c, 0, out);
    if (rc) return rc;
    if (!do_pw) out->n_patches = 0;
    return SCVOD_OK;
}

// ---- host-side pose algebra (pcl::getTransformation + Eigen::Affine3f inverse / product) ----
void get_transformation(const float p[6], float t[12]) {
    const float x = p[0], y = p[1], z = p[2], roll = p[3], pitch = p[4], yaw = p[5];
    const float A = std::cos(yaw), B = std::sin(yaw), C = std::cos(pitch), D = std::sin(pitch), E = std::cos(roll),
                F = std::sin(roll), DE = D * E, DF = D * F;
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
inline float red3(float a, float b, float c) { return a + (b + c); }  // Eigen fixed-size redux order

// label filter + pcl::VoxelGrid over a batch of scans resident in HBM.  Two phases with one small host
// round trip each: the largest cell-index range picks the bucket shift of the voxel stage; the per-scan output counts
// become the output offsets.
int run_voxelgrid(scvod_ctx* c, const void* d_xyzi, const uint32_t* d_labels, const int32_t* h_off, int32_t n_scans,
                  const float leaf[3], float max_intensity, void* d_out, int64_t out_cap, int32_t* h_out_off, hipStream_t st) {
    if (!c) return SCVOD_ERR_INVALID;
    if (n_scans <= 0 || !h_off || !d_xyzi || !leaf || !d_out || !h_out_off) return fail(c, SCVOD_ERR_INVALID, "bad arguments");
    if (!(leaf[0] > 0.f) || !(leaf[1] > 0.f) || !(leaf[2] > 0.f)) return fail(c, SCVOD_ERR_INVALID, "leaf size must be positive");
    if (n_scans > c->cap_scans) return fail(c, SCVOD_ERR_CAPACITY, "n_scans %d > capacity %d", n_scans, c->cap_scans);
    if (h_off[0] != 0) return fail(c, SCVOD_ERR_INVALID, "scan_offsets[0] must be 0");
    const int64_t total = h_off[n_scans];
    if (total > c->cap_pts) return fail(c, SCVOD_ERR_CAPACITY, "%lld points > capacity %lld", (long long)total, (long long)c->cap_pts);
    int32_t mx = 0;
    for (int s = 0; s < n_scans; ++s) {
        const int32_t n = h_off[s + 1] - h_off[s];
        if (n < 0) return fail(c, SCVOD_ERR_INVALID, "scan_offsets not monotone");
        if (n > mx) mx = n;
    }
    if (mx > SCVOD_MAX_SCAN_POINTS)
        return fail(c, SCVOD_ERR_CAPACITY, "scan of %d points > SCVOD_MAX_SCAN_POINTS (%d)", mx, SCVOD_MAX_SCAN_POINTS);
    HIPCHK(c, hipSetDevice(c->device));
    if (!st) st = c->stream;
    c->last_stream = st;
    join_lastname(c, st);
    c->h_scan_off.assign(h_off, h_off + n_scans + 1);
    HIPCHK(c, hipMemcpyAsync(c->d_scan_off, c->h_scan_off.data(), sizeof(int32_t) * (n_scans + 1), hipMemcpyHostToDevice, st));
    c->A.pts = (const float4*)d_xyzi;
    c->A.scan_off = c->d_scan_off;
    c->A.n_scans = n_scans;
    c->A.max_scan_pts = mx;
    c->A.total_pts = total;
    c->tim_used = 0;
    c->batch_valid = c->counts_valid = c->voxels_valid = c->clusters_valid = c->types_valid = c->track_valid = c->tables_valid = false;  // the arena is reused
    h_out_off[0] = 0;
    if (mx == 0) {
        for (int s = 0; s < n_scans; ++s) h_out_off[s + 1] = 0;
        return SCVOD_OK;
    }
    if (int rc_side = ensure_side(c, kSideApri)) return rc_side;  // (the centroids of a VoxelGrid run live in the PointAPRI array)
    VgJob J;
    J.labels = d_labels;
    J.max_intensity = max_intensity;
    for (int a = 0; a < 3; ++a) J.inv_leaf[a] = 1.0f / leaf[a];
    J.out = (float4*)d_out;
    // everything on the stream, one synchronisation at the end: bounding boxes + cell keys, the population-balanced
    // bucket table, the voxel stage in its centroid mode, output offsets, compaction into the caller's buffer
    Arena Av = c->A;
    Av.vb_lut = c->d_vb_lut;
    Av.vb_lut_shift = c->A.vg_range + 1;  // second word of the two-int scratch
    Av.vg_labels = d_labels;
    Av.vg_max_intensity = max_intensity;
    Av.vx_k32 = 0;  // cell indices are not bucket-relative: 64-bit sort keys
    DevParams D = c->dev;
    D.key_off = 0;
    D.vb_shift = 0;
    D.n_buckets = kMaxBuckets;
    launch_voxelgrid_keys(Av, J, st);
    launch_voxelgrid_lut(Av, st);
    launch_process(D, Av, st, 3, 0, 1, nullptr, nullptr);
    launch_voxelgrid_gather(D, Av, J, (long long)out_cap, st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_out_off, c->A.vg_outoff, sizeof(int32_t) * (n_scans + 1), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (h_out_off[n_scans] > out_cap)
        return fail(c, SCVOD_ERR_CAPACITY, "output buffer too small (%lld < %d points)", (long long)out_cap, h_out_off[n_scans]);
    return SCVOD_OK;
}

// Re-upload a small host table only when its contents changed since the last call (steady-state sequence loops pass the
// same tables every step and upload nothing).
// A changed table travels through one of eight pinned staging slots; a slot is reused only after the copy that read it
// has completed (its event), so a chunked sequence (scvod_sequence_ingest: new transforms every chunk) never drains the stream.
int staged_upload(scvod_ctx* c, const void* src, size_t bytes, void* dst, hipStream_t st) {
    UploadSlot& u = c->up_ring[c->up_next_slot];
    c->up_next_slot = (c->up_next_slot + 1) % 8;
    if (!u.ev) HIPCHK(c, hipEventCreateWithFlags(&u.ev, hipEventDisableTiming));
    if (u.used) HIPCHK(c, hipEventSynchronize(u.ev));
    if (bytes > u.cap) {
        if (u.pinned) hipHostFree(u.pinned);
        u.pinned = nullptr;
        u.cap = 0;
        HIPCHK(c, hipHostMalloc(&u.pinned, bytes + bytes / 2 + 64, hipHostMallocDefault));
        u.cap = bytes + bytes / 2 + 64;
    }
    memcpy(u.pinned, src, bytes);
    HIPCHK(c, hipMemcpyAsync(dst, u.pinned, bytes, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipEventRecord(u.ev, st));
    u.used = true;
    return SCVOD_OK;
}
template <typename T>
int upload_if_changed(scvod_ctx* c, std::vector<T>& held, const T* src, size_t n, void* dst, hipStream_t st) {
    if (held.size() == n && (n == 0 || memcmp(held.data(), src, n * sizeof(T)) == 0)) return SCVOD_OK;
    held.assign(src, src + n);
    return n ? staged_upload(c, held.data(), n * sizeof(T), dst, st) : SCVOD_OK;
}

// ---- sequential tracking chain: plan (chains of the successor table cut into segments) and workspace ----
size_t chain_layout(ChainWs& g, int max_scan_pts, int64_t pool_points) {
    const size_t nv = (size_t)(max_scan_pts > 0 ? max_scan_pts : 1);
    // appended clouds a state can hold: given, or 8 scans' worth (tracking CONSECUTIVE scans of a 10 Hz sequence keeps an object
    // in range for tens of frames and its cloud grows with every one of them -- the reference's own quadratic behaviour)
    size_t cap_pool = pool_points > 0 ? (size_t)pool_points : nv * 8;
    if (cap_pool < 65536) cap_pool = 65536;
    const size_t cap_ent = nv / 4 + 1024;
    g.cap_pool = (int32_t)cap_pool;
    g.cap_ent = (int32_t)cap_ent;
    g.cap_nv = (int32_t)nv;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        off = align_up(off, 256);
        const size_t o = off;
        off += bytes;
        return o;
    };
    g.off_hdr = take(16 * 4);
    for (int i = 0; i < 3; ++i) {
        g.off_ent[i] = take(2 * cap_ent * sizeof(int4));
        g.off_parts[i] = take(cap_ent * 4);
        g.off_pool[i] = take(cap_pool * sizeof(float4));
    }
    g.off_chit = take(cap_pool * 4);
    g.off_evr = take(cap_ent * sizeof(int2));
    g.off_suniq = take((cap_pool + nv) * 4);
    g.off_spairs = take((cap_pool + nv) * sizeof(int2));
    g.off_vlab = take(nv * 8);
    g.off_lcnt = take((nv + cap_ent) * 8);
    g.off_lfwd = take((nv + cap_ent) * 8);
    g.off_newent = take(cap_ent * 4);
    g.off_cmeta = take(cap_ent * sizeof(int4));
    g.off_cparts = take(cap_ent * 4);
    g.off_links = take(cap_ent * sizeof(int4));
    g.off_dsz = take(2 * cap_ent * 4);
    g.off_eidx = take(2 * cap_ent * 4);
    g.off_rp = take(nv * sizeof(int2));
    g.stride = align_up(off, 4096);
    return g.stride;
}

// Chains = maximal runs scan -> successor -> ... inside the batch; every chain is cut into segments of `seg` steps, each
// walked by one workgroup that warms up `warm` steps earlier (scvod_chain.hip).  Returns the number of walkers, or a
// negative status when a scan is the successor of two scans (no sequence looks like that).
int plan_chains(scvod_ctx* c, const std::vector<int32_t>& next, std::vector<int32_t>& scans, std::vector<int32_t>& fw,
                std::vector<int32_t>& walkers) {
    const int B = (int)next.size();
    std::vector<int32_t> pred(B, -1);
    for (int s = 0; s < B; ++s)
        if (next[s] >= 0) {
            if (pred[next[s]] != -1) return fail(c, SCVOD_ERR_INVALID, "scan %d is the successor of scans %d and %d", next[s], pred[next[s]], s);
            if (next[s] == s) return fail(c, SCVOD_ERR_INVALID, "scan %d is its own successor", s);
            pred[next[s]] = s;
        }
    scans.clear();
    fw.assign(1, 0);
    walkers.clear();
    std::vector<char> seen(B, 0);
    // the chains: frames in order
    std::vector<int> chain_first, chain_len;
    for (int h = 0; h < B; ++h) {
        if (pred[h] != -1 || next[h] < 0) continue;  // not a head, or a head without a successor in the batch
        chain_first.push_back((int)scans.size());
        int n = 0;
        for (int s = h; s >= 0 && !seen[s]; s = next[s] >= 0 ? next[s] : -1) {
            seen[s] = 1;
            scans.push_back(s);
            ++n;
        }
        chain_len.push_back(n);
    }
    // segment length: given, or the SHORTEST whose segments still fit the device one per CU (a walker fills a CU; every walker
    // pays the warm-up, so more segments than CUs would run in two rounds) -- 11 steps = 255 walkers for seq 05 with skip_ 5
    int seg = c->chain_seg;
    if (seg <= 0) {
        const int n_cu = c->n_cu;
        int longest = 0;
        for (int n : chain_len) longest = n - 1 > longest ? n - 1 : longest;
        for (seg = 4; seg < longest; ++seg) {  // (more chains than CUs: one segment per chain)
            long long nseg = 0;
            for (int n : chain_len) nseg += (n - 1 + seg - 1) / seg;
            if (nseg <= n_cu) break;
        }
    }
    c->chain_seg_used = seg;
    if (c->chain_fb_pending && hipEventQuery(c->chain_fb_ev) == hipSuccess) {  // how the last batch of this stream fared
        c->chain_fb_pending = false;
        const int rewalked = c->h_chain_fb[1], limit = c->chain_fb_segments / 64 > 2 ? c->chain_fb_segments / 64 : 2;
        if (rewalked > limit && c->chain_warm_auto < 16) c->chain_warm_auto += 2;
    }
    const int warm = c->chain_warm >= 0 ? c->chain_warm : c->chain_warm_auto;
    c->chain_warm_used = warm;
    // the times the walkers of this stream's last batch measured per step (same plan of frames): a kernel lasts as long as its slowest
    // walker, and what a step costs follows the stretch of the sequence (objects tracked for tens of frames carry large clouds), not
    // anything the planner could read off the scans -- so the segments of the NEXT batch are cut to equal measured time (warm-up included)
    if (c->step_fb_pending && hipEventQuery(c->step_fb_ev) == hipSuccess) {
        c->step_fb_pending = false;
        c->step_ticks.assign(c->h_step_ticks, c->h_step_ticks + c->step_fb_scans.size());
        c->step_ticks_scans = c->step_fb_scans;
    }
    std::vector<std::vector<int>> cuts;  // per chain: segment starts (own steps), when cut by time
    // (planned ONCE per plan of frames, halo and warm-up -- from the first times that arrive for it -- and kept: the host pays for the
    //  search once, the walker table is uploaded once, and a stream's batches keep one segmentation)
    if (c->chain_seg <= 0 && c->chain_balance && !c->cuts_cache.empty() && c->cuts_scans == scans && c->cuts_chain_first == chain_first &&
        c->cuts_chain_len == chain_len && c->cuts_warm == warm && c->cuts_halo == c->halo) {
        cuts = c->cuts_cache;
    } else if (c->chain_seg <= 0 && c->chain_balance && !c->step_ticks.empty() && c->step_ticks_scans == scans) {
        std::vector<double> cost(scans.size(), 0.0);
        double sum = 0;
        long long cnt = 0;
        for (size_t i = 0; i < scans.size(); ++i)
            if (c->step_ticks[i] > 0) {
                sum += c->step_ticks[i];
                ++cnt;
            }
        const double mean = cnt ? sum / (double)cnt : 1.0;
        for (size_t i = 0; i < scans.size(); ++i) cost[i] = c->step_ticks[i] > 0 ? (double)c->step_ticks[i] : mean;
        std::vector<int> a0s(chain_len.size());
        double total = 0, biggest = 0;
        for (size_t ci = 0; ci < chain_len.size(); ++ci) {
            const int first = chain_first[ci], steps = chain_len[ci] - 1;
            int a0 = 0;
            while (a0 < steps && (size_t)scans[first + a0] < c->halo.size() && c->halo[scans[first + a0]]) ++a0;
            a0s[ci] = a0;
            for (int k = 0; k < steps; ++k) total += cost[first + k];
        }
        auto plan = [&](double tau, std::vector<std::vector<int>>* out) -> long long {
            long long nseg = 0;
            if (out) out->assign(chain_len.size(), std::vector<int>());
            for (size_t ci = 0; ci < chain_len.size(); ++ci) {
                const int first = chain_first[ci], steps = chain_len[ci] - 1, a0 = a0s[ci];
                int a = a0;
                while (a < steps) {
                    const int t0 = (a == a0 && a0 > 0) ? 0 : (a - warm > 0 ? a - warm : 0);
                    double t = 0;
                    for (int k = t0; k < a; ++k) t += cost[first + k];
                    int b = a;
                    do {
                        t += cost[first + b];
                        ++b;
                    } while (b < steps && t + cost[first + b] <= tau);
                    if (out) (*out)[ci].push_back(a);
                    ++nseg;
                    a = b;
                }
            }
            return nseg;
        };
        double lo_t = 0, hi_t = total + 1.0;  // smallest tau whose plan fits the device one walker per CU
        if (plan(hi_t, nullptr) <= c->n_cu) {
            for (int it = 0; it < 40; ++it) {
                const double mid = 0.5 * (lo_t + hi_t);
                if (plan(mid, nullptr) <= c->n_cu)
                    hi_t = mid;
                else
                    lo_t = mid;
            }
            plan(hi_t, &cuts);
            c->cuts_cache = cuts;
            c->cuts_scans = scans;
            c->cuts_chain_first = chain_first;
            c->cuts_chain_len = chain_len;
            c->cuts_warm = warm;
            c->cuts_halo = c->halo;
        }
    }
    if (!cuts.empty()) {  // cuts that do not describe THIS plan (one list per chain, ascending starts inside the chain's steps): equal-length segments instead
        bool ok = cuts.size() == chain_len.size();
        for (size_t ci = 0; ok && ci < cuts.size(); ++ci) {
            int prev = -1;
            for (int a : cuts[ci]) {
                if (a <= prev || a >= chain_len[ci] - 1) ok = false;
                prev = a;
            }
        }
        if (!ok) {
            cuts.clear();
            c->cuts_cache.clear();
        }
    }
    int n_chains = 0;
    for (size_t ci = 0; ci < chain_len.size(); ++ci) {
        const int first = chain_first[ci], n = chain_len[ci];
        const int steps = n - 1;
        // frames of a halo (scvod_set_track_owned): the chain's own steps start at its first frame that is not one; the steps before
        // are only ever a warm-up (the whole halo for the first walker: its start state has no other source on this shard)
        int a0 = 0;
        while (a0 < steps && (size_t)scans[first + a0] < c->halo.size() && c->halo[scans[first + a0]]) ++a0;
        for (int a = a0; a < n; ++a)  // a halo flag behind a scan of the chain that is not halo would be ignored silently: refuse it
            if ((size_t)scans[first + a] < c->halo.size() && c->halo[scans[first + a]] && a > a0)
                return fail(c, SCVOD_ERR_INVALID, "scan %d is marked halo but follows scan %d of its chain, which is not: a chain's halo scans must be its first ones", scans[first + a], scans[first + a0]);
        if (!cuts.empty()) {
            const std::vector<int>& cs = cuts[ci];
            for (size_t k = 0; k < cs.size(); ++k) {
                const int a = cs[k], b = k + 1 < cs.size() ? cs[k + 1] : steps;
                const int t0 = (a == a0 && a0 > 0) ? 0 : (a - warm > 0 ? a - warm : 0);
                const int32_t w[8] = {first, n, a, b, t0, n_chains, (a == a0 && a0 > 0) ? 1 : 0, 0};
                walkers.insert(walkers.end(), w, w + 8);
            }
            fw.push_back((int32_t)(walkers.size() / 8));
            ++n_chains;
            continue;
        }
        for (int a = a0; a < steps; a += seg) {
            const int b = a + seg < steps ? a + seg : steps;
            const int t0 = (a == a0 && a0 > 0) ? 0 : (a - warm > 0 ? a - warm : 0);
            const int32_t w[8] = {first, n, a, b, t0, n_chains, (a == a0 && a0 > 0) ? 1 : 0, 0};
            walkers.insert(walkers.end(), w, w + 8);
        }
        fw.push_back((int32_t)(walkers.size() / 8));
        ++n_chains;
    }
    // (a cycle has no head: its scans keep their first-order result; rejected here to be explicit)
    for (int s = 0; s < B; ++s)
        if (next[s] >= 0 && !seen[s]) return fail(c, SCVOD_ERR_INVALID, "the successor table holds a cycle through scan %d", s);
    return (int)(walkers.size() / 8);
}

}  // namespace

// internal bridge for scvod_map.hip (not part of the public header)
extern "C" int scvod__ctx_types_valid(scvod_ctx* c) { return (c->clusters_valid && c->types_valid) ? 1 : 0; }
extern "C" int scvod__ctx_view(scvod_ctx* c, Arena* arena, int* device, int* track_valid, int* batch_valid, int* n_scans,
                               int* max_scan_pts, int* batch_mode, int* num_min_pts) {
    *batch_mode = c->batch_mode;
    *num_min_pts = c->pw.num_min_pts;
    *arena = c->A;
    *device = c->device;
    *track_valid = c->track_valid ? 1 : 0;
    *batch_valid = (c->batch_valid && c->batch_mode != 3) ? 1 : 0;
    *n_scans = c->A.n_scans;
    *max_scan_pts = c->A.max_scan_pts;
    return 0;
}

extern "C" {

void scvod_params_default(scvod_params* p) {
    memset(p, 0, sizeof(*p));
    p->sensor_height = 2.0f;
    p->min_dis = 0.0f;
    p->max_dis = 50.0f;
    p->min_angle = 0.0f;
    p->max_angle = 360.0f;
    p->min_azimuth = -30.0f;
    p->max_azimuth = 60.0f;
    p->range_res = 0.2f;
    p->sector_res = 1.2f;
    p->azimuth_res = 2.0f;
    p->occupancy = 0.6f;
    p->max_z = 1.0f;       // utility.h:294
    p->min_z = -1.0f;      // utility.h:295
    p->car_square = 2.0f;  // utility.h:298
    p->toBeClass = 1;      // utility.h:306
}

void scvod_pw_params_default(scvod_pw_params* p) {
    memset(p, 0, sizeof(*p));
    p->num_iter = 3;
    p->num_lpr = 20;
    p->num_min_pts = 10;
    p->num_rings_of_interest = 4;
    const int sec[4] = {16, 32, 54, 32}, rng[4] = {2, 4, 4, 4};
    const double el[4] = {-1.2, -0.9984, -0.851, -0.605}, fl[4] = {0.0, 0.000125, 0.000185, 0.000185};
    for (int i = 0; i < 4; ++i) {
        p->num_sectors_each_zone[i] = sec[i];
        p->num_rings_each_zone[i] = rng[i];
        p->elevation_thr[i] = el[i];
        p->flatness_thr[i] = fl[i];
    }
    p->th_seeds = 0.3;
    p->th_dist = 0.1;
    p->max_range = 80.0;
    p->min_range = 2.7;
    p->uprightness_thr = 0.707;
    p->adaptive_seed_selection_margin = -1.1;
}

void scvod_grid_dims(const scvod_params* p, int32_t* range_num, int32_t* sector_num, int32_t* azimuth_num,
                     int32_t* bin_num) {
    // src/ssc.cpp:36-39: float arithmetic, std::ceil(float)
    const int r = (int)std::ceil((p->max_dis - p->min_dis) / p->range_res);
    const int s = (int)std::ceil((p->max_angle - p->min_angle) / p->sector_res);
    const int a = (int)std::ceil((p->max_azimuth - p->min_azimuth) / p->azimuth_res);
    if (range_num) *range_num = r;
    if (sector_num) *sector_num = s;
    if (azimuth_num) *azimuth_num = a;
    if (bin_num) *bin_num = r * s * a;
}

int scvod_create(const scvod_params* params, const scvod_pw_params* pw, int device, int64_t max_points_total,
                 int32_t max_scans, scvod_ctx** out) {
    if (!params || !out || max_points_total <= 0 || max_scans <= 0) return SCVOD_ERR_INVALID;
    *out = nullptr;
    if (max_points_total > 2147483583ll) return SCVOD_ERR_CAPACITY;  // scan offsets and point indices are int32
    scvod_ctx* c = new scvod_ctx();
    c->device = device;
    c->params = *params;
    if (pw)
        c->pw = *pw;
    else
        scvod_pw_params_default(&c->pw);
    int np = 0;
    for (int k = 0; k < 4; ++k) {
        if (c->pw.num_rings_each_zone[k] <= 0 || c->pw.num_sectors_each_zone[k] <= 0) {
            delete c;
            return SCVOD_ERR_INVALID;
        }
        np += c->pw.num_rings_each_zone[k] * c->pw.num_sectors_each_zone[k];
    }
    // the elevation / flatness gates are indexed ring + 2 * zone for the patches of the first num_rings_of_interest
    // concentric rings (patchwork.h:351-353) and hold four entries (patchwork.h:50-51): a layout that would index past
    // them reads out of bounds in the reference; refused here
    {
        int conc = 0;
        for (int k = 0; k < 4; ++k)
            for (int r = 0; r < c->pw.num_rings_each_zone[k]; ++r, ++conc)
                if (conc < c->pw.num_rings_of_interest && r + 2 * k > 3) {
                    delete c;
                    return SCVOD_ERR_INVALID;
                }
    }
    // th_seeds < 0 or a near-zero th_dist could leave a plane iteration without points: the reference then fits with the
    // stale moments of the previous patch (PCL leaves cov / mean untouched for an empty cloud) -- not modelled, refused
    if (np > kMaxPatches || !(params->range_res > 0) || !(params->sector_res > 0) || !(params->azimuth_res > 0) ||
        !(c->pw.th_seeds >= 0.0) || !(c->pw.th_dist >= 0.01) || c->pw.num_iter < 1 || c->pw.num_lpr < 1) {
        delete c;
        return SCVOD_ERR_INVALID;
    }
    int ndev = 0;  // (argument errors are reported before the device is looked for)
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        delete c;
        return SCVOD_ERR_NO_DEVICE;
    }
    c->cap_pts = max_points_total;
    c->cap_scans = max_scans;
    build_dev_params(c);
    if (hipSetDevice(device) != hipSuccess) {
        delete c;
        return SCVOD_ERR_NO_DEVICE;
    }
    size_t total = 0;
    carve(c, nullptr, &total);
    if (hipMalloc(&c->arena_base, total) != hipSuccess) {
        delete c;
        return SCVOD_ERR_HIP;
    }
    c->arena_bytes = total;
    carve(c, (unsigned char*)c->arena_base, &total);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
        if (const char* e = getenv("SCVOD_CHAIN_BALANCE")) c->chain_balance = atoi(e);  // (development: 0 = equal-length segments, as rounds 3-4)
    }
    if (hipStreamCreate(&c->stream) != hipSuccess) {
        hipFree(c->arena_base);
        delete c;
        return SCVOD_ERR_HIP;
    }
    *out = c;
    return SCVOD_OK;
}

void scvod_destroy(scvod_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (auto& t : c->tim) {
        hipEventDestroy(t.e0);
        hipEventDestroy(t.e1);
    }
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->ln_stream) hipStreamDestroy(c->ln_stream);
    if (c->h_chain_fb) hipHostFree(c->h_chain_fb);
    if (c->h_step_ticks) hipHostFree(c->h_step_ticks);
    if (c->h_ext_state_pin) hipHostFree(c->h_ext_state_pin);
    if (c->step_fb_ev) hipEventDestroy(c->step_fb_ev);
    if (c->chain_fb_ev) hipEventDestroy(c->chain_fb_ev);
    if (c->ln_stream2) hipStreamDestroy(c->ln_stream2);
    if (c->ln_stream3) hipStreamDestroy(c->ln_stream3);
    if (c->ln_join3) hipEventDestroy(c->ln_join3);
    if (c->ln_fork2) hipEventDestroy(c->ln_fork2);
    if (c->ln_join2) hipEventDestroy(c->ln_join2);
    if (c->ln_fork) hipEventDestroy(c->ln_fork);
    if (c->ln_done) hipEventDestroy(c->ln_done);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    for (int k = 0; k < 2; ++k) {
        if (c->ingest_buf[k]) hipFree(c->ingest_buf[k]);
        if (c->ingest_copied[k]) hipEventDestroy(c->ingest_copied[k]);
        if (c->ingest_done[k]) hipEventDestroy(c->ingest_done[k]);
    }
    if (c->ingest_off) hipHostFree(c->ingest_off);
    for (UploadSlot& u : c->up_ring) {
        if (u.pinned) hipHostFree(u.pinned);
        if (u.ev) hipEventDestroy(u.ev);
    }
    if (c->arena_base) hipFree(c->arena_base);
    if (c->d_in) hipFree(c->d_in);
    if (c->t_pts) hipFree(c->t_pts);
    if (c->d_labels) hipFree(c->d_labels);
    if (c->d_apri) hipFree(c->d_apri);
    if (c->chain_ws) hipFree(c->chain_ws);
    if (c->stage) hipHostFree(c->stage);
    for (void* b : c->nn_buf)
        if (b) hipFree(b);
    delete c;
}

const char* scvod_last_error(const scvod_ctx* c) { return c ? c->err.c_str() : "null ctx"; }
int64_t scvod_arena_bytes(const scvod_ctx* c) { return c ? (int64_t)(c->arena_bytes + c->side_bytes) : 0; }

int scvod_process_scan(scvod_ctx* c, const float* h_xyzi, int32_t n, scvod_scan_result* out) {
    return host_scan(c, h_xyzi, n, 1, 1, 1, out);
}
int scvod_patchwork(scvod_ctx* c, const float* h_xyzi, int32_t n, scvod_scan_result* out) {
    int rc = host_scan(c, h_xyzi, n, 1, 1, 0, out);
    if (rc == SCVOD_OK) {
        out->n_apri = out->n_rejected = out->n_voxels = 0;
    }
    return rc;
}
int scvod_bin_scan(scvod_ctx* c, const float* h_xyzi, int32_t n, int32_t apply_filter, int32_t with_voxels,
                   scvod_scan_result* out) {
    return host_scan(c, h_xyzi, n, 0, apply_filter ? 1 : 0, with_voxels ? 1 : 0, out);
}

int scvod_voxelize(scvod_ctx* c, const scvod_apri* h_apri, int32_t n, scvod_scan_result* out) {
    if (!c || !out || n < 0 || (n > 0 && !h_apri)) return fail(c, SCVOD_ERR_INVALID, "bad arguments");
    if (n > c->cap_pts) return fail(c, SCVOD_ERR_CAPACITY, "%d points > capacity %lld", n, (long long)c->cap_pts);
    HIPCHK(c, hipSetDevice(c->device));
    const int32_t counts[8] = {n, 0, 0, 0, n, 0, 0, 0};
    if (int rc_side = ensure_side(c, kSideApri)) return rc_side;
    if (n) HIPCHK(c, hipMemcpyAsync(c->A.apri, h_apri, sizeof(scvod_apri) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->A.counts, counts, sizeof(counts), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int32_t off[2] = {0, n};
    int rc = run_batch(c, nullptr, off, 1, c->stream, 2, 0, 1, 1);
    if (rc) return rc;
    rc = fetch_scan(c, 0, out);
    if (rc) return rc;
    out->n_patches = 0;
    return SCVOD_OK;
}

void scvod_pose_delta(const float pose_pre[6], const float pose_next[6], float T_out[12]) {
    float tn[12], tp[12];
    get_transformation(pose_next, tn);
    get_transformation(pose_pre, tp);
    // Eigen::Affine3f::inverse(): cofactor inverse of the linear part, then -R^-1 * t
    auto M = [&](int i, int j) { return tn[4 * i + j]; };
    auto cof = [&](int i, int j) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return M(i1, j1) * M(i2, j2) - M(i1, j2) * M(i2, j1);
    };
    const float c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
    const float det = red3(c0 * M(0, 0), c1 * M(1, 0), c2 * M(2, 0));
    const float invdet = 1.f / det;
    float R[3][3];
    R[0][0] = c0 * invdet;
    R[0][1] = c1 * invdet;
    R[0][2] = c2 * invdet;
    R[1][0] = cof(0, 1) * invdet;
    R[1][1] = cof(1, 1) * invdet;
    R[1][2] = cof(2, 1) * invdet;
    R[2][0] = cof(0, 2) * invdet;
    R[2][1] = cof(1, 2) * invdet;
    R[2][2] = cof(2, 2) * invdet;
    float ti[12];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) ti[4 * i + j] = R[i][j];
        ti[4 * i + 3] = red3((-R[i][0]) * tn[3], (-R[i][1]) * tn[7], (-R[i][2]) * tn[11]);
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            T_out[4 * i + j] = red3(ti[4 * i] * tp[j], ti[4 * i + 1] * tp[4 + j], ti[4 * i + 2] * tp[8 + j]);
        T_out[4 * i + 3] = red3(ti[4 * i] * tp[3], ti[4 * i + 1] * tp[7], ti[4 * i + 2] * tp[11]) + ti[4 * i + 3];
    }
}

int scvod_track_probe(scvod_ctx* c, const float* h_xyzi, const int32_t* h_offsets, int32_t n_clusters, const float T[12],
                      const int32_t* h_next_keys, const int32_t* h_next_labels, int32_t n_next_vox, int32_t* h_hit_slot,
                      int32_t* h_uniq_slots, int32_t* h_uniq_begin) {
    if (!c || n_clusters < 0 || !h_offsets || !T || n_next_vox < 0) return fail(c, SCVOD_ERR_INVALID, "bad arguments");
    if (h_offsets[0] != 0) return fail(c, SCVOD_ERR_INVALID, "cluster offsets must start at 0");
    for (int k = 0; k < n_clusters; ++k)
        if (h_offsets[k + 1] < h_offsets[k]) return fail(c, SCVOD_ERR_INVALID, "cluster offsets not monotone");
    const int32_t n_pts = h_offsets[n_clusters];
    if ((n_pts > 0 && !h_xyzi) || (n_next_vox > 0 && !h_next_keys)) return fail(c, SCVOD_ERR_INVALID, "null input array");
    if (n_pts > c->cap_pts || n_clusters > c->cap_pts || n_next_vox > c->cap_pts)
        return fail(c, SCVOD_ERR_CAPACITY, "track probe larger than the ctx capacity");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    c->last_stream = st;
    join_lastname(c, st);
    c->tim_used = 0;
    // the one-shot probe writes t_T and scratch that scvod_batch_track's results alias: forget the cached upload of T and
    // the batch's tracking result (a following scvod_batch_track uploads and decides again)
    c->up_T.clear();
    c->track_valid = false;
    if (int rc_side = ensure_side(c, kSideTpts)) return rc_side;
    float4* d_pts = c->t_pts;
    if (n_pts) HIPCHK(c, hipMemcpyAsync(d_pts, h_xyzi, sizeof(float) * 4 * (size_t)n_pts, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->t_begin, h_offsets, sizeof(int32_t) * (n_clusters + 1), hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemcpyAsync(c->t_T, T, sizeof(float) * 12, hipMemcpyHostToDevice, st));
    if (n_next_vox) {
        HIPCHK(c, hipMemcpyAsync(c->A.tmp_vox_key, h_next_keys, sizeof(int32_t) * n_next_vox, hipMemcpyHostToDevice, st));
        if (h_next_labels)
            HIPCHK(c, hipMemcpyAsync(c->A.tmp_vox_begin, h_next_labels, sizeof(int32_t) * n_next_vox, hipMemcpyHostToDevice, st));
    }
    TrackJob J;
    memset(&J, 0, sizeof(J));
    J.pts = d_pts;
    J.pt_cluster_begin = c->t_begin;
    J.n_clusters = n_clusters;
    J.n_pts = n_pts;
    J.T = c->t_T;
    J.next_keys = c->A.tmp_vox_key;
    J.next_labels = h_next_labels ? c->A.tmp_vox_begin : nullptr;
    J.n_next_vox = n_next_vox;
    J.hit_slot = c->t_hit;
    J.work = c->t_work;
    J.uniq_slots = c->t_uniq;
    J.uniq_count = c->t_count;
    launch_track(c->dev, c->A, J, 0, st, timer_hook, c);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));
    if (n_pts && h_hit_slot) HIPCHK(c, hipMemcpy(h_hit_slot, c->t_hit, sizeof(int32_t) * n_pts, hipMemcpyDeviceToHost));
    std::vector<int32_t> cnt(n_clusters > 0 ? n_clusters : 1), uq(n_pts > 0 ? n_pts : 1);
    if (n_clusters) HIPCHK(c, hipMemcpy(cnt.data(), c->t_count, sizeof(int32_t) * n_clusters, hipMemcpyDeviceToHost));
    if (n_pts) HIPCHK(c, hipMemcpy(uq.data(), c->t_uniq, sizeof(int32_t) * n_pts, hipMemcpyDeviceToHost));
    int32_t o = 0;
    for (int k = 0; k < n_clusters; ++k) {
        if (h_uniq_begin) h_uniq_begin[k] = o;
        for (int j = 0; j < cnt[k]; ++j) {
            if (h_uniq_slots) h_uniq_slots[o] = uq[h_offsets[k] + j];
            ++o;
        }
    }
    if (h_uniq_begin) h_uniq_begin[n_clusters] = o;
    return SCVOD_OK;
}

int scvod_batch_process(scvod_ctx* c, const void* d_xyzi, const int32_t* h_scan_offsets, int32_t n_scans, void* stream,
                        int32_t sync) {
    return run_batch(c, d_xyzi, h_scan_offsets, n_scans, (hipStream_t)stream, 1, 1, 1, sync);
}

int scvod_batch_counts(scvod_ctx* c, int32_t* h_out) {
    if (!c || !h_out) return SCVOD_ERR_INVALID;
    int rc = ensure_counts(c);
    if (rc) return rc;
    memcpy(h_out, c->h_counts.data(), sizeof(int32_t) * c->h_counts.size());
    return SCVOD_OK;
}

int scvod_batch_fetch(scvod_ctx* c, int32_t s, scvod_scan_result* out) {
    if (!c || !out) return SCVOD_ERR_INVALID;
    return fetch_scan(c, s, out);
}

int scvod_batch_cluster(scvod_ctx* c, void* stream, int32_t sync) {
    if (!c) return SCVOD_ERR_INVALID;
    if (!c->batch_valid || !c->voxels_valid)
        return fail(c, SCVOD_ERR_STATE, "scvod_batch_cluster needs a processed batch with voxel descriptors");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    c->last_stream = st;
    c->tim_used = 0;
    if (c->dev.bin.range_num > 2040 || c->dev.bin.sector_num > 2040 || c->dev.bin.azimuth_num > 1016)
        return fail(c, SCVOD_ERR_INVALID, "grid of %d x %d x %d bins is finer than the clustering's packed index triples hold (2040 x 2040 x 1016)",
                    c->dev.bin.range_num, c->dev.bin.sector_num, c->dev.bin.azimuth_num);
    join_lastname(c, st);
    // a second clustering of the same batch (other mode, scvod_set_cluster_exact, cluster - track - cluster): the marks the first one
    // and its tracking left per input point would otherwise survive on points that are no car members any more
    if (c->clustered_this_batch && c->A.total_pts > 0) HIPCHK(c, hipMemsetAsync(c->A.pt_mapcls, 0, (size_t)c->A.total_pts, st));
    c->clustered_this_batch = true;
    launch_cluster(c->dev, c->A, c->batch_mode == 2 ? 1 : 0, st, timer_hook, c);
    c->last_name_valid = false;
    // Frame::max_name of every scan (scvod_lastname.hip): launches of its own beside the tracking kernels.  (Run by the clustering
    // workgroup itself -- 1024 threads, one workgroup per CU -- the pass is a chain of dependent look-ups nobody hides: measured
    // in round 4, k_cc_scan 3.1 -> 7.6 ms; as separate passes with eight small workgroups per CU it costs 3 ms, partly hidden.)
    if (c->max_name_literal) {
        if (!c->ln_stream) {
            // (highest priority: the few workgroups of these passes -- one per scan that needs one, each a chain of dependent
            // look-ups -- are dispatched ahead of the tracking kernels' thousands and finish while those stream)
            int prio_low = 0, prio_high = 0;
            HIPCHK(c, hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
            if (getenv("SCVOD_LN_NO_PRIORITY")) prio_high = prio_low;
            HIPCHK(c, hipStreamCreateWithPriority(&c->ln_stream, hipStreamNonBlocking, prio_high));
            HIPCHK(c, hipEventCreateWithFlags(&c->ln_fork, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ln_done, hipEventDisableTiming));
            HIPCHK(c, hipStreamCreateWithPriority(&c->ln_stream2, hipStreamNonBlocking, prio_high));
            HIPCHK(c, hipEventCreateWithFlags(&c->ln_fork2, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ln_join2, hipEventDisableTiming));
            HIPCHK(c, hipStreamCreateWithPriority(&c->ln_stream3, hipStreamNonBlocking, prio_high));
            HIPCHK(c, hipEventCreateWithFlags(&c->ln_join3, hipEventDisableTiming));
        }
        if (c->timing) {  // attribution runs: one stream, one pair of events per pass
            launch_lastname(c->dev, c->A, st, nullptr, nullptr, nullptr, nullptr, nullptr, timer_hook, c);
        } else {
            HIPCHK(c, hipEventRecord(c->ln_fork, st));
            HIPCHK(c, hipStreamWaitEvent(c->ln_stream, c->ln_fork, 0));
            launch_lastname(c->dev, c->A, c->ln_stream, c->ln_stream2, c->ln_stream3, c->ln_fork2, c->ln_join2, c->ln_join3, nullptr, c);
            HIPCHK(c, hipEventRecord(c->ln_done, c->ln_stream));
            c->ln_pending = true;
        }
        c->last_name_valid = true;
        if (sync) join_lastname(c, st);
    }
    HIPCHK(c, hipGetLastError());
    c->clusters_valid = true;
    c->types_valid = false;  // scvod_batch_cluster_types publishes them
    c->track_valid = false;
    c->tables_valid = false;
    if (sync) HIPCHK(c, hipStreamSynchronize(st));
    return SCVOD_OK;
}

int scvod_batch_fetch_clusters(scvod_ctx* c, int32_t s, int32_t* h_pt_cluster, int32_t cap) {
    if (!c || !h_pt_cluster) return SCVOD_ERR_INVALID;
    if (!c->clusters_valid) return fail(c, SCVOD_ERR_STATE, "no clusters computed for the last batch");
    int rc = ensure_counts(c);
    if (rc) return rc;
    if (s < 0 || s >= c->A.n_scans) return fail(c, SCVOD_ERR_INVALID, "scan %d out of range", s);
    const int32_t n = c->h_counts[(size_t)s * 8 + 4];
    if (n > cap) return fail(c, SCVOD_ERR_CAPACITY, "output buffer too small (%d < %d)", cap, n);
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    if (n) HIPCHK(c, hipMemcpy(h_pt_cluster, c->A.pt_cluster + c->h_scan_off[s], sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    return n;
}

int scvod_batch_cluster_types(scvod_ctx* c, void* stream, int32_t sync) {
    if (!c) return SCVOD_ERR_INVALID;
    if (!c->clusters_valid) return fail(c, SCVOD_ERR_STATE, "scvod_batch_cluster_types needs scvod_batch_cluster first");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    c->last_stream = st;
    // (nothing is launched here: the max_name pass keeps running beside what follows)
    c->tim_used = 0;
    // the boxes and the type rules are evaluated by the clustering kernel itself (same workgroup, boxes in LDS): nothing to launch
    c->types_valid = true;
    c->track_valid = false;
    c->tables_valid = false;
    if (sync) HIPCHK(c, hipStreamSynchronize(st));
    return SCVOD_OK;
}

int scvod_batch_fetch_cluster_types(scvod_ctx* c, int32_t s, int32_t car_label, int32_t other_label, int32_t* h_type,
                                    int32_t cap) {
    if (!c || !h_type) return SCVOD_ERR_INVALID;
    if (!c->types_valid) return fail(c, SCVOD_ERR_STATE, "no cluster types computed for the last batch");
    int rc = ensure_counts(c);
    if (rc) return rc;
    if (s < 0 || s >= c->A.n_scans) return fail(c, SCVOD_ERR_INVALID, "scan %d out of range", s);
    const int32_t n = c->h_counts[(size_t)s * 8 + 4];
    if (n > cap) return fail(c, SCVOD_ERR_CAPACITY, "output buffer too small (%d < %d)", cap, n);
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    std::vector<uint8_t> t(n ? n : 1);
    if (n) HIPCHK(c, hipMemcpy(t.data(), c->A.pt_type + c->h_scan_off[s], (size_t)n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) h_type[i] = t[i] == 0 ? -1 : (t[i] == 2 ? car_label : other_label);
    return n;
}

int scvod_cluster(scvod_ctx* c, const scvod_apri* h_apri, int32_t n, int32_t* h_pt_cluster) {
    scvod_scan_result r;
    int rc = scvod_voxelize(c, h_apri, n, &r);
    if (rc) return rc;
    rc = scvod_batch_cluster(c, nullptr, 1);
    if (rc) return rc;
    rc = scvod_batch_fetch_clusters(c, 0, h_pt_cluster, n);
    return rc < 0 ? rc : SCVOD_OK;
}

int scvod_batch_track(scvod_ctx* c, const float* h_T, const int32_t* h_next_scan, const void* const* h_ext_tables, int32_t n_ext,
                      void* stream, int32_t sync) {
    if (!c || !h_T || n_ext < 0 || (n_ext > 0 && !h_ext_tables)) return fail(c, SCVOD_ERR_INVALID, "bad arguments");
    if (!c->batch_valid || !c->voxels_valid || !c->clusters_valid || !c->types_valid)
        return fail(c, SCVOD_ERR_STATE, "scvod_batch_track needs scvod_batch_process, scvod_batch_cluster and scvod_batch_cluster_types first");
    if (c->batch_mode == 3) return fail(c, SCVOD_ERR_STATE, "the last batch was a VoxelGrid run");
    const int B = c->A.n_scans;
    if (n_ext > c->cap_scans) return fail(c, SCVOD_ERR_CAPACITY, "too many external tables");
    if ((int)c->halo.size() > B)  // (a mask left over from a batch with another layout would silently turn leading scans into warm-up only)
        return fail(c, SCVOD_ERR_INVALID, "the halo mask covers %d scans, the batch holds %d: call scvod_set_track_owned / scvod_set_track_halo for this batch", (int)c->halo.size(), B);
    std::vector<int32_t> next(B);
    for (int s = 0; s < B; ++s) {
        const int32_t v = h_next_scan ? h_next_scan[s] : (s + 1 < B ? s + 1 : -1);
        if (v >= B || (v <= -2 && -2 - v >= n_ext)) return fail(c, SCVOD_ERR_INVALID, "next_scan[%d] = %d out of range", s, v);
        next[s] = v;
    }
    for (int e = 0; e < n_ext; ++e)
        if (!h_ext_tables[e]) return fail(c, SCVOD_ERR_INVALID, "external table %d is NULL", e);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    c->last_stream = st;
    c->tim_used = 0;
    int rc;
    if ((rc = upload_if_changed(c, c->up_T, h_T, (size_t)12 * B, c->t_T, st))) return rc;
    if ((rc = upload_if_changed(c, c->up_next, next.data(), (size_t)B, c->d_next_scan, st))) return rc;
    if ((rc = upload_if_changed(c, c->up_ext, h_ext_tables, (size_t)n_ext, (void*)c->d_ext, st))) return rc;
    TrackBatch J;
    J.next_scan = c->d_next_scan;
    J.ext_tables = c->d_ext;
    J.n_ext = n_ext;
    J.T = c->t_T;
    J.occupancy = c->params.occupancy;
    ChainJob CJ;
    memset(&CJ, 0, sizeof(CJ));
    c->chain_ran = false;
    if (c->track_mode == SCVOD_TRACK_CHAIN && c->A.max_scan_pts > 0) {
        std::vector<int32_t> scans, fw, walkers;
        const int nw = plan_chains(c, next, scans, fw, walkers);
        if (nw < 0) return nw;
        if (nw > 0) {
            // The layout follows the LARGEST scan this ctx has tracked so far (rounded up to 4096 points), not the current batch's:
            // a stream of batches whose largest scan wobbles keeps one layout, one workspace, and is not cleared again
            const int lay_pts = ((c->A.max_scan_pts + 4095) / 4096) * 4096;
            if (lay_pts > c->chain_geom_pts || c->chain_geom_pool != c->chain_pool_points) {
                const int pts = lay_pts > c->chain_geom_pts ? lay_pts : c->chain_geom_pts;
                chain_layout(c->chain_geom, pts, c->chain_pool_points);
                c->chain_geom_pts = pts;
                c->chain_geom_pool = c->chain_pool_points;
                c->chain_ws_clean = false;
            }
            // (the plan cut by measured time -- the stream's second or third batch -- has up to one walker per CU: room for them from the first call,
            //  or that batch would stop the device, free, allocate and clear ~18 GB again: 3.7 ms of a one-shot job's second step, round 6)
            const int nw_room = (c->chain_seg <= 0 && c->chain_balance && nw < c->n_cu) ? c->n_cu : nw;
            const size_t need = (size_t)nw_room * c->chain_geom.stride;
            if (need > c->chain_ws_bytes) {  // first call / larger job: the only allocation, outside any steady-state step
                HIPCHK(c, hipDeviceSynchronize());
                if (c->chain_ws) hipFree(c->chain_ws);
                c->chain_ws = nullptr;
                c->chain_ws_bytes = 0;
                HIPCHK(c, hipMalloc(&c->chain_ws, need));
                c->chain_ws_bytes = need;
                c->chain_ws_clean = false;
            }
            // the stamped override words compare against a per-walker epoch that only grows: a zeroed workspace is a valid one
            if (!c->chain_ws_clean || c->chain_ws_layout_pts != c->chain_geom_pts) {
                HIPCHK(c, hipMemsetAsync(c->chain_ws, 0, c->chain_ws_bytes, st));
                c->chain_ws_clean = true;
                c->chain_ws_layout_pts = c->chain_geom_pts;
            }
            if ((rc = upload_if_changed(c, c->up_chain_scans, scans.data(), scans.size(), c->d_chain_scans, st))) return rc;
            if ((rc = upload_if_changed(c, c->up_chain_fw, fw.data(), fw.size(), c->d_chain_fw, st))) return rc;
            if ((rc = upload_if_changed(c, c->up_chain_walkers, walkers.data(), walkers.size(), (void*)c->d_chain_walkers, st))) return rc;
            HIPCHK(c, hipMemsetAsync(c->d_chain_stats, 0, 8 * sizeof(int32_t), st));
            CJ.chain_scans = c->d_chain_scans;
            CJ.walkers = c->d_chain_walkers;
            CJ.chain_first_walker = c->d_chain_fw;
            CJ.n_walkers = nw;
            CJ.n_chains = (int)fw.size() - 1;
            CJ.ws = c->chain_geom;
            CJ.ws.base = (unsigned char*)c->chain_ws;
            CJ.stats = c->d_chain_stats;
            CJ.words = (c->chain_geom_pts + 31) / 32 + 1;  // (sized like the layout: one value for the workspace and the bitsets)
            const size_t lds_bits = 72 * 1024;  // next to the 80 KB of per-step tables
            int ev = (int)(lds_bits / ((size_t)CJ.words * 4));
            CJ.n_eval_waves = ev < 1 ? 1 : (ev > 16 ? 16 : ev);
            if ((size_t)CJ.words * 4 > lds_bits) return fail(c, SCVOD_ERR_CAPACITY, "scan too large for the chain's LDS bitset");
            CJ.force_generic = c->chain_generic ? 1 : 0;
            CJ.literal_max_name = (c->max_name_literal && c->last_name_valid) ? 1 : 0;
            CJ.ext_state = nullptr;
            CJ.resume = 0;
            CJ.step_ticks = c->d_step_ticks;
            HIPCHK(c, hipMemsetAsync(c->d_step_ticks, 0, sizeof(int32_t) * scans.size(), st));
            c->plan_scans_tmp = scans;
            c->last_cj = CJ;
            c->last_tb = J;
            c->chain_ran = true;
        }
    }
    launch_track_batch(c->dev, c->A, J, c->batch_mode == 2 ? 1 : 0, c->tables_valid ? 2 : 3, st, timer_hook, c, c->chain_ran ? &CJ : nullptr,
                       c->ln_pending ? c->ln_done : nullptr);
    if (c->chain_ran) c->ln_pending = false;  // (the chain waited for it)
    if (c->chain_ran) {  // the counters of this chain, for the warm-up of the stream's next batch
        if (!c->h_chain_fb) {
            HIPCHK(c, hipHostMalloc((void**)&c->h_chain_fb, 8 * sizeof(int32_t)));
            HIPCHK(c, hipEventCreateWithFlags(&c->chain_fb_ev, hipEventDisableTiming));
        }
        if (!c->chain_fb_pending) {
            HIPCHK(c, hipMemcpyAsync(c->h_chain_fb, c->d_chain_stats, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipEventRecord(c->chain_fb_ev, st));
            c->chain_fb_pending = true;
            c->chain_fb_segments = CJ.n_walkers;
        }
        if (!c->h_step_ticks) {
            HIPCHK(c, hipHostMalloc((void**)&c->h_step_ticks, sizeof(int32_t) * (size_t)(c->cap_scans + 1)));
            HIPCHK(c, hipEventCreateWithFlags(&c->step_fb_ev, hipEventDisableTiming));
        }
        if (!c->step_fb_pending && c->chain_balance && !(c->cuts_scans == c->plan_scans_tmp && !c->cuts_cache.empty())) {  // the per-step times, for the cuts of the stream's next batch
            HIPCHK(c, hipMemcpyAsync(c->h_step_ticks, c->d_step_ticks, sizeof(int32_t) * c->plan_scans_tmp.size(), hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipEventRecord(c->step_fb_ev, st));
            c->step_fb_pending = true;
            c->step_fb_scans = c->plan_scans_tmp;
        }
    }
    HIPCHK(c, hipGetLastError());
    c->tables_valid = true;
    c->track_valid = true;
    if (sync) HIPCHK(c, hipStreamSynchronize(st));
    return SCVOD_OK;
}

int scvod_get_params(const scvod_ctx* c, scvod_params* out) {
    if (!c || !out) return SCVOD_ERR_INVALID;
    *out = c->params;
    return SCVOD_OK;
}

int64_t scvod_chain_workspace_bytes(scvod_ctx* c) { return c ? (int64_t)c->chain_ws_bytes : 0; }

int scvod_set_track_owned(scvod_ctx* c, int32_t first_owned_scan) {
    if (!c || first_owned_scan < 0) return fail(c, SCVOD_ERR_INVALID, "bad first owned scan");
    c->halo.assign((size_t)first_owned_scan, 1);
    c->track_valid = false;
    return SCVOD_OK;
}

int scvod_set_track_halo(scvod_ctx* c, const uint8_t* h_is_halo, int32_t n_scans) {
    if (!c || n_scans < 0 || (n_scans > 0 && !h_is_halo)) return fail(c, SCVOD_ERR_INVALID, "bad halo mask");
    c->halo.assign(h_is_halo, h_is_halo + n_scans);
    c->track_valid = false;
    return SCVOD_OK;
}

int64_t scvod_chain_state_bytes(scvod_ctx* c) {
    if (!c || !c->chain_ran) return 0;
    return (int64_t)chain_state_bytes(c->last_cj.ws);
}

int scvod_chain_export_state(scvod_ctx* c, int32_t chain, int32_t which, void* d_dst, int64_t cap_bytes, void* stream) {
    if (!c || !d_dst || (which != 0 && which != 1)) return SCVOD_ERR_INVALID;
    if (!c->track_valid || !c->chain_ran) return fail(c, SCVOD_ERR_STATE, "no tracking chain of the last batch");
    if (chain < 0 || chain >= c->last_cj.n_chains) return fail(c, SCVOD_ERR_INVALID, "chain %d out of range (%d chains)", chain, c->last_cj.n_chains);
    if (cap_bytes < 16) return fail(c, SCVOD_ERR_CAPACITY, "state buffer too small");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : (c->last_stream ? c->last_stream : c->stream);  // (behind the walkers of the last scvod_batch_track)
    launch_chain_export_state(c->last_cj, chain, which, (unsigned char*)d_dst, cap_bytes, st);
    HIPCHK(c, hipGetLastError());
    return SCVOD_OK;
}

int scvod_batch_track_chains(scvod_ctx* c, int32_t* h_first_scan, int32_t cap) {
    if (!c || !h_first_scan) return SCVOD_ERR_INVALID;
    if (!c->track_valid || !c->chain_ran) return fail(c, SCVOD_ERR_STATE, "no tracking chain of the last batch");
    const int nc = c->last_cj.n_chains;
    if (cap < nc) return fail(c, SCVOD_ERR_CAPACITY, "output buffer too small (%d < %d chains)", cap, nc);
    // a chain without a step of its own has no walker: its head is reported all the same (up_chain_fw / up_chain_scans hold the plan)
    int k = 0;
    std::vector<char> is_succ(c->up_next.size(), 0);
    for (size_t s = 0; s < c->up_next.size(); ++s)
        if (c->up_next[s] >= 0) is_succ[c->up_next[s]] = 1;
    for (size_t s = 0; s < c->up_next.size() && k < nc; ++s)
        if (!is_succ[s] && c->up_next[s] >= 0) h_first_scan[k++] = (int32_t)s;
    return nc;
}

int scvod_batch_track_resume(scvod_ctx* c, const void* const* h_d_states, int32_t n_states, void* stream, int32_t sync) {
    if (!c || !h_d_states) return SCVOD_ERR_INVALID;
    if (!c->track_valid || !c->chain_ran) return fail(c, SCVOD_ERR_STATE, "scvod_batch_track_resume needs a chain-mode scvod_batch_track first");
    if (n_states != c->last_cj.n_chains) return fail(c, SCVOD_ERR_INVALID, "%d states for %d chains", n_states, c->last_cj.n_chains);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : (c->last_stream ? c->last_stream : c->stream);
    c->last_stream = st;
    c->ext_state_pin_valid = false;
    HIPCHK(c, hipMemcpyAsync((void*)c->d_ext_state, h_d_states, sizeof(void*) * (size_t)n_states, hipMemcpyHostToDevice, st));
    ChainJob CJ = c->last_cj;
    CJ.ext_state = c->d_ext_state;
    CJ.resume = 1;
    launch_track_chain_resume(c->dev, c->A, c->last_tb, CJ, c->batch_mode == 2 ? 1 : 0, st);
    launch_track_dyn(c->A, c->batch_mode == 2 ? 1 : 0, st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(st));  // (the pointer table is a pageable host array)
    (void)sync;
    return SCVOD_OK;
}

int scvod_batch_track_compare(scvod_ctx* c, const void* const* h_d_states, int32_t n_states, int32_t* h_differ, void* stream) {
    if (!c || !h_d_states || !h_differ) return SCVOD_ERR_INVALID;
    if (!c->track_valid || !c->chain_ran) return fail(c, SCVOD_ERR_STATE, "scvod_batch_track_compare needs a chain-mode scvod_batch_track first");
    if (n_states != c->last_cj.n_chains) return fail(c, SCVOD_ERR_INVALID, "%d states for %d chains", n_states, c->last_cj.n_chains);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : (c->last_stream ? c->last_stream : c->stream);
    c->last_stream = st;
    c->ext_state_pin_valid = false;
    HIPCHK(c, hipMemcpyAsync((void*)c->d_ext_state, h_d_states, sizeof(void*) * (size_t)n_states, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemsetAsync(c->d_chain_cmp, 0, sizeof(int32_t), st));
    ChainJob CJ = c->last_cj;
    CJ.ext_state = c->d_ext_state;
    CJ.resume = 2;
    CJ.cmp_out = c->d_chain_cmp;
    launch_track_chain_resume(c->dev, c->A, c->last_tb, CJ, c->batch_mode == 2 ? 1 : 0, st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_differ, c->d_chain_cmp, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    return SCVOD_OK;
}

int scvod_batch_track_compare_device(scvod_ctx* c, const void* const* h_d_states, int32_t n_states, int32_t* d_differ, void* stream) {
    if (!c || !h_d_states || !d_differ) return SCVOD_ERR_INVALID;
    if (!c->track_valid || !c->chain_ran) return fail(c, SCVOD_ERR_STATE, "scvod_batch_track_compare_device needs a chain-mode scvod_batch_track first");
    if (n_states != c->last_cj.n_chains) return fail(c, SCVOD_ERR_INVALID, "%d states for %d chains", n_states, c->last_cj.n_chains);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : (c->last_stream ? c->last_stream : c->stream);
    c->last_stream = st;
    // the pointer table goes through a pinned copy of the ctx (nothing waits on the host, the caller's array may go away); it is
    // uploaded again only when it changed (a shard's receive buffers stay where they are from step to step)
    if (!c->h_ext_state_pin) HIPCHK(c, hipHostMalloc((void**)&c->h_ext_state_pin, sizeof(void*) * (size_t)c->cap_scans));
    if (!c->ext_state_pin_valid || c->ext_state_pin_n != n_states || memcmp(c->h_ext_state_pin, h_d_states, sizeof(void*) * (size_t)n_states) != 0) {
        if (c->ext_state_pin_valid) HIPCHK(c, hipStreamSynchronize(st));  // (a table in flight is still being read: rare, the table changed)
        memcpy(c->h_ext_state_pin, h_d_states, sizeof(void*) * (size_t)n_states);
        HIPCHK(c, hipMemcpyAsync((void*)c->d_ext_state, c->h_ext_state_pin, sizeof(void*) * (size_t)n_states, hipMemcpyHostToDevice, st));
        c->ext_state_pin_valid = true;
        c->ext_state_pin_n = n_states;
    }
    ChainJob CJ = c->last_cj;
    CJ.ext_state = c->d_ext_state;
    CJ.resume = 2;
    CJ.cmp_out = d_differ;
    launch_track_chain_resume(c->dev, c->A, c->last_tb, CJ, c->batch_mode == 2 ? 1 : 0, st);
    HIPCHK(c, hipGetLastError());
    return SCVOD_OK;
}

int scvod_set_cluster_exact(scvod_ctx* c, int32_t on) {
    if (!c) return SCVOD_ERR_INVALID;
    c->A.cc_exact_max = on ? 0x3fffffff : 4096;
    c->A.cc_plain_rule = on == 2 ? 0 : 1;
    c->A.cc_help_blocks_wanted = on == 3 ? 0 : 1;  // 3: exact, every scan's workgroup on its own (the round-5 form: A/B runs, tests of the shared rounds)
    c->clusters_valid = c->types_valid = c->tables_valid = c->track_valid = false;
    return SCVOD_OK;
}

int scvod_batch_cluster_stats(scvod_ctx* c, int32_t* h_out4) {
    if (!c || !h_out4) return SCVOD_ERR_INVALID;
    if (!c->clusters_valid) return fail(c, SCVOD_ERR_STATE, "no clustering of the last batch");
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    HIPCHK(c, hipMemcpy(h_out4, c->A.cc_stats, 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
    h_out4[2] = c->A.cc_exact_max > 4096 ? 1 : 0;
    return SCVOD_OK;
}

int scvod_batch_cluster_rule_stats(scvod_ctx* c, int32_t* h_out2) {
    if (!c || !h_out2) return SCVOD_ERR_INVALID;
    if (!c->clusters_valid) return fail(c, SCVOD_ERR_STATE, "no clustering of the last batch");
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    HIPCHK(c, hipMemcpy(h_out2, c->A.cc_stats + 4, 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return SCVOD_OK;
}

int scvod_batch_cluster_help_stats(scvod_ctx* c, int32_t* h_out2) {
    if (!c || !h_out2) return SCVOD_ERR_INVALID;
    if (!c->clusters_valid) return fail(c, SCVOD_ERR_STATE, "no clustering of the last batch");
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    HIPCHK(c, hipMemcpy(h_out2, c->A.cc_help, sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(h_out2 + 1, c->A.cc_stats + 6, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (getenv("SCVOD_CC_HELP_DUMP")) {  // development: the board of the last launch (10 ns ticks relative to the first scan block)
        std::vector<int32_t> b(kCcHelpWords);
        HIPCHK(c, hipMemcpy(b.data(), c->A.cc_help, b.size() * 4, hipMemcpyDeviceToHost));
        auto t64 = [&](size_t w) { return *reinterpret_cast<long long*>(&b[w]); };
        long long t0 = 0x7fffffffffffffffll;
        for (int k = 0; k < b[0] && k < kCcHelpSlots; ++k) t0 = std::min(t0, t64(kCcHelpHdr + (size_t)k * kCcHelpSlotWords + 20));
        int32_t nr = 0;
        HIPCHK(c, hipMemcpy(&nr, c->A.cc_again, 4, hipMemcpyDeviceToHost));
        fprintf(stderr, "cc_help: %d scans handed to k_cc_exact, %d asked for help, %d past their shared passes (clocks: x10 ns after the first rounds began)\n", nr, b[0], b[1]);
        for (int k = 0; k < b[0] && k < kCcHelpSlots; ++k) {
            const size_t w = kCcHelpHdr + (size_t)k * kCcHelpSlotWords;
            fprintf(stderr, "  slot %d scan %d: na %d chunks/round %d rounds %d leader chunks %d | entry +%lld rule done +%lld listed +%lld rounds start +%lld end +%lld unions end +%lld scan done +%lld\n", k, b[w + 18], b[w + 3], b[w + 4], b[w + 19], b[w + 5],
                    t64(w + 38) - t0, t64(w + 40) - t0, t64(w + 42) - t0, t64(w + 20) - t0, t64(w + 22) - t0, t64(w + 24) - t0, t64(w + 54) - t0);
        }
    }
    return SCVOD_OK;
}

int scvod_set_max_name_literal(scvod_ctx* c, int32_t literal) {
    if (!c) return SCVOD_ERR_INVALID;
    c->max_name_literal = literal != 0;
    c->clusters_valid = c->types_valid = c->tables_valid = c->track_valid = false;  // (the clustering publishes the name)
    return SCVOD_OK;
}

int scvod_batch_cluster_last_name(scvod_ctx* c, int32_t* h_out4, int32_t cap_scans, int32_t* h_stats4) {
    if (!c || !h_out4) return SCVOD_ERR_INVALID;
    if (!c->clusters_valid || !c->last_name_valid) return fail(c, SCVOD_ERR_STATE, "no clustering with max_name tracking for the last batch");
    if (cap_scans < c->A.n_scans) return fail(c, SCVOD_ERR_CAPACITY, "output buffer too small (%d < %d scans)", cap_scans, c->A.n_scans);
    join_lastname(c, c->last_stream);
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    HIPCHK(c, hipMemcpy(h_out4, c->A.cc_last, sizeof(int32_t) * 4 * (size_t)c->A.n_scans, hipMemcpyDeviceToHost));
    if (h_stats4) HIPCHK(c, hipMemcpy(h_stats4, c->A.ln_stats, 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (getenv("SCVOD_LN_PROF")) {  // development: phase clocks of the first scans
        std::vector<int32_t> pr(8 * (size_t)c->A.n_scans);
        HIPCHK(c, hipMemcpy(pr.data(), c->A.ln_prof, pr.size() * 4, hipMemcpyDeviceToHost));
        std::vector<int32_t> pr2(pr.size());
        HIPCHK(c, hipMemcpy(pr2.data(), c->A.ln_prof2, pr2.size() * 4, hipMemcpyDeviceToHost));
        for (int s = 0; s < c->A.n_scans && s < 64; ++s)
            fprintf(stderr, "ln_big scan %d: nodes %d rounds %d build %d jacobi %d open %d cc %d walk %d events %d\n", s, pr2[8 * s], pr2[8 * s + 1], pr2[8 * s + 2], pr2[8 * s + 3], pr2[8 * s + 4], pr2[8 * s + 5], pr2[8 * s + 6], pr2[8 * s + 7]),
            fprintf(stderr, "ln_prof scan %d: setup %d A+irr.. CL %d cand %d marked %d (x10ns) n_cand %d n_mk %d n_irr %d cap %d\n", s, pr[8 * s], pr[8 * s + 1], pr[8 * s + 2], pr[8 * s + 3], pr[8 * s + 4], pr[8 * s + 5], pr[8 * s + 6], pr[8 * s + 7]);
    }
    return c->A.n_scans;
}

int scvod_set_chain_capacity(scvod_ctx* c, int64_t pool_points) {
    if (!c || pool_points < 0 || pool_points > (1ll << 30)) return fail(c, SCVOD_ERR_INVALID, "bad chain capacity");
    c->chain_pool_points = pool_points;
    return SCVOD_OK;
}

int scvod_set_track_mode(scvod_ctx* c, int32_t mode, int32_t segment_steps, int32_t warmup_steps) {
    if (!c || (mode != SCVOD_TRACK_CHAIN && mode != SCVOD_TRACK_FIRST_ORDER && mode != SCVOD_TRACK_CHAIN_GENERIC))
        return fail(c, SCVOD_ERR_INVALID, "unknown tracking mode");
    if (segment_steps < 0 || warmup_steps < -1) return fail(c, SCVOD_ERR_INVALID, "negative segment / warm-up length");
    c->chain_generic = (mode == SCVOD_TRACK_CHAIN_GENERIC);
    if (mode == SCVOD_TRACK_CHAIN_GENERIC) mode = SCVOD_TRACK_CHAIN;
    c->track_mode = mode;
    c->chain_seg = segment_steps;                   // 0: chosen per job
    if (warmup_steps >= 0) c->chain_warm = warmup_steps;  // -1: keep
    c->track_valid = false;
    return SCVOD_OK;
}

// error bits / counters of the chain of the last scvod_batch_track (synchronises its stream)
static int chain_status(scvod_ctx* c, int32_t out[8]) {
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (!c->chain_ran) return SCVOD_OK;
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    HIPCHK(c, hipMemcpy(out, c->d_chain_stats, 8 * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (out[0])
        return fail(c, SCVOD_ERR_CAPACITY, "tracking chain overflow (bits %d: 1 appended clouds > %d points (scvod_set_chain_capacity), 2 more than %d clusters, 4 created labels, 8 table > bitset)",
                    out[0], c->chain_geom.cap_pool, c->chain_geom.cap_ent);
    return SCVOD_OK;
}

int scvod_batch_track_stats(scvod_ctx* c, int32_t* h_out8) {
    if (!c || !h_out8) return SCVOD_ERR_INVALID;
    if (!c->track_valid) return fail(c, SCVOD_ERR_STATE, "no tracking result for the last batch");
    int32_t st[8];
    const int rc = chain_status(c, st);
    h_out8[0] = c->chain_ran ? SCVOD_TRACK_CHAIN : SCVOD_TRACK_FIRST_ORDER;
    h_out8[1] = (int32_t)(c->up_chain_walkers.size() / 8) * (c->chain_ran ? 1 : 0);
    h_out8[2] = st[2];
    h_out8[3] = st[1];
    h_out8[4] = st[0];
    h_out8[5] = c->chain_seg_used;
    h_out8[6] = c->chain_warm_used;
    h_out8[7] = 0;
    if (c->chain_ran && c->max_name_literal && c->last_name_valid) {  // scans whose max_name the clustering could not determine: the chain handed out a fresh number there
        int32_t ls[4] = {0, 0, 0, 0};
        HIPCHK(c, hipMemcpy(ls, c->A.ln_stats, sizeof(ls), hipMemcpyDeviceToHost));
        h_out8[7] = ls[0] + ls[1];
    }
#ifdef SCVOD_PROFILE
    fprintf(stderr, "[chain phases, 10 ns ticks summed over walkers x steps] fetch %d  carried %d  eval %d  walk+state %d  copy %d\n", st[3], st[4], st[5], st[6], st[7]);
    if (c->chain_ran) {
        const int nw = (int)(c->up_chain_walkers.size() / 8);
        std::vector<int32_t> hd(16);
        long long sum = 0;
        int mn = 0x7fffffff, mx = 0;
        std::vector<int> all;
        for (int w = 0; w < nw; ++w) {
            hipMemcpy(hd.data(), (unsigned char*)c->chain_ws + (size_t)w * c->chain_geom.stride + c->chain_geom.off_hdr, 64, hipMemcpyDeviceToHost);
            sum += hd[15];
            mn = hd[15] < mn ? hd[15] : mn;
            mx = hd[15] > mx ? hd[15] : mx;
            all.push_back(hd[15]);
        }
        std::sort(all.begin(), all.end());
        fprintf(stderr, "[chain walkers] %d: wall min %.1f us  median %.1f  p90 %.1f  max %.1f  mean %.1f\n", nw, mn * 0.01, all[nw / 2] * 0.01, all[nw * 9 / 10] * 0.01,
                mx * 0.01, sum * 0.01 / nw);
    }
#endif
    return rc;
}

int scvod_batch_track_tables(scvod_ctx* c, void* stream) {
    if (!c) return SCVOD_ERR_INVALID;
    if (!c->batch_valid || !c->voxels_valid || !c->clusters_valid || !c->types_valid)
        return fail(c, SCVOD_ERR_STATE, "scvod_batch_track_tables needs scvod_batch_process, scvod_batch_cluster and scvod_batch_cluster_types first");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    c->last_stream = st;
    join_lastname(c, st);
    c->tim_used = 0;
    TrackBatch J;
    memset(&J, 0, sizeof(J));
    launch_track_batch(c->dev, c->A, J, 0, 1, st, timer_hook, c);
    HIPCHK(c, hipGetLastError());
    c->tables_valid = true;
    c->track_valid = false;
    return SCVOD_OK;
}

int scvod_batch_export_table(scvod_ctx* c, int32_t s, void* d_out, int64_t cap_records, void* stream) {
    if (!c || !d_out || cap_records < 1) return fail(c, SCVOD_ERR_INVALID, "bad arguments");
    if (!c->tables_valid) return fail(c, SCVOD_ERR_STATE, "scvod_batch_export_table needs scvod_batch_track_tables or scvod_batch_track first");
    if (s < 0 || s >= c->A.n_scans) return fail(c, SCVOD_ERR_INVALID, "scan %d out of range", s);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    launch_export_table(c->A, s, (int4*)d_out, (long long)cap_records, st);
    HIPCHK(c, hipGetLastError());
    return SCVOD_OK;
}

int scvod_batch_fetch_track(scvod_ctx* c, int32_t s, scvod_track_result* out) {
    if (!c || !out) return SCVOD_ERR_INVALID;
    if (!c->track_valid) return fail(c, SCVOD_ERR_STATE, "no tracking result for the last batch");
    int rc = ensure_counts(c);
    if (rc) return rc;
    if (s < 0 || s >= c->A.n_scans) return fail(c, SCVOD_ERR_INVALID, "scan %d out of range", s);
    HIPCHK(c, hipStreamSynchronize(c->last_stream));
    {
        int32_t cst[8];
        if ((rc = chain_status(c, cst))) return rc;
    }
    const Arena& A = c->A;
    const size_t base = (size_t)c->h_scan_off[s];
    const int n = c->h_counts[(size_t)s * 8 + 4];
    int32_t sc[4];
    HIPCHK(c, hipMemcpy(sc, A.tk_scan + (size_t)s * 4, sizeof(sc), hipMemcpyDeviceToHost));
    const int ncl = sc[0] < 0 ? 0 : (sc[0] > n ? n : sc[0]);
    memset(out, 0, sizeof(*out));
    out->n_apri = n;
    out->n_clusters = ncl;
    out->n_car_points = sc[1];
    out->n_dynamic_clusters = sc[2];
    out->n_dynamic_points = sc[3];
    // per-root device arrays of this scan -> compact per-cluster host arrays
    const size_t nn = n > 0 ? (size_t)n : 1;
    std::vector<int32_t> roots(nn), cnt(nn), nuq(nn), npr(nn), mbeg(nn);
    std::vector<int8_t> state(nn);
    std::vector<int2> pairs(nn);
    c->tk_stage_dyn.resize(nn);
    if (n) {
        HIPCHK(c, hipMemcpy(roots.data(), A.tk_clusters + base, 4 * (size_t)(ncl > 0 ? ncl : 0), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(cnt.data(), A.cl_count + base, 4 * nn, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(nuq.data(), A.tk_nuniq + base, 4 * nn, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(npr.data(), A.tk_npairs + base, 4 * nn, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(mbeg.data(), A.tk_mbegin + base, 4 * nn, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(state.data(), A.cl_state + base, nn, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(pairs.data(), A.tk_pairs + base, 8 * nn, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(c->tk_stage_dyn.data(), A.pt_dyn + base, nn, hipMemcpyDeviceToHost));
    }
    size_t n_pairs = 0;
    for (int k = 0; k < ncl; ++k) n_pairs += (size_t)npr[roots[k]];
    std::vector<int32_t>& h = c->tk_stage;
    h.assign((size_t)5 * ncl + 1 + 2 * n_pairs + 8, 0);
    int32_t* p_root = h.data();
    int32_t* p_size = p_root + ncl;
    int32_t* p_state = p_size + ncl;
    int32_t* p_nuq = p_state + ncl;
    int32_t* p_pbeg = p_nuq + ncl;
    int32_t* p_plab = p_pbeg + ncl + 1;
    int32_t* p_pcnt = p_plab + n_pairs;
    size_t o = 0;
    for (int k = 0; k < ncl; ++k) {
        const int r = roots[k];
        p_root[k] = r;
        p_size[k] = cnt[r];
        p_state[k] = state[r];
        p_nuq[k] = nuq[r];
        p_pbeg[k] = (int32_t)o;
        for (int j = 0; j < npr[r]; ++j, ++o) {
            p_plab[o] = pairs[(size_t)mbeg[r] + j].x;
            p_pcnt[o] = pairs[(size_t)mbeg[r] + j].y;
        }
    }
    p_pbeg[ncl] = (int32_t)o;
    out->cluster_root = p_root;
    out->cluster_size = p_size;
    out->cluster_state = p_state;
    out->n_unique = p_nuq;
    out->pair_begin = p_pbeg;
    out->pair_label = p_plab;
    out->pair_count = p_pcnt;
    out->pt_dyn = c->tk_stage_dyn.data();
    return SCVOD_OK;
}

// Streaming ingest of a sequence that lives in HOST memory (the reference reads one .bin per scan from disk,
// SSC::getCloud ssc.cpp:1040-1125): chunks of scans are copied host -> device on a copy stream into one of two buffers
// while the previous chunk is processed on the compute stream; `fn` is called once per chunk, after its
// scvod_batch_process launches are enqueued, to enqueue the consumers of that chunk's results (clustering, tracking, map
// accumulation ...) on the same stream -- the arena holds one chunk at a time.
int scvod_sequence_ingest(scvod_ctx* c, const float* h_xyzi, const int32_t* h_scan_offsets, int32_t n_scans, int32_t chunk_scans,
                          int32_t flags, scvod_chunk_fn fn, void* user) {
    if (!c || !h_xyzi || !h_scan_offsets || n_scans <= 0 || chunk_scans <= 0) return fail(c, SCVOD_ERR_INVALID, "bad arguments");
    if (chunk_scans > c->cap_scans) return fail(c, SCVOD_ERR_CAPACITY, "chunk of %d scans > capacity %d", chunk_scans, c->cap_scans);
    const int n_chunks = (n_scans + chunk_scans - 1) / chunk_scans;
    int64_t max_chunk_pts = 0;
    for (int k = 0; k < n_chunks; ++k) {
        const int s0 = k * chunk_scans, s1 = s0 + chunk_scans < n_scans ? s0 + chunk_scans : n_scans;
        const int64_t p = (int64_t)h_scan_offsets[s1] - h_scan_offsets[s0];
        if (p < 0) return fail(c, SCVOD_ERR_INVALID, "scan_offsets not monotone");
        if (p > max_chunk_pts) max_chunk_pts = p;
    }
    if (max_chunk_pts > c->cap_pts) return fail(c, SCVOD_ERR_CAPACITY, "chunk of %lld points > capacity %lld", (long long)max_chunk_pts, (long long)c->cap_pts);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        if (!c->ingest_copied[k]) HIPCHK(c, hipEventCreateWithFlags(&c->ingest_copied[k], hipEventDisableTiming));
        if (!c->ingest_done[k]) HIPCHK(c, hipEventCreateWithFlags(&c->ingest_done[k], hipEventDisableTiming));
    }
    if ((size_t)max_chunk_pts > c->ingest_cap) {
        HIPCHK(c, hipDeviceSynchronize());
        for (int k = 0; k < 2; ++k) {
            if (c->ingest_buf[k]) hipFree(c->ingest_buf[k]);
            c->ingest_buf[k] = nullptr;
        }
        c->ingest_cap = 0;
        for (int k = 0; k < 2; ++k) HIPCHK(c, hipMalloc(&c->ingest_buf[k], 16 * (size_t)max_chunk_pts));
        c->ingest_cap = (size_t)max_chunk_pts;
    }
    const size_t off_need = (size_t)n_scans + (size_t)n_chunks + 8;
    if (off_need > c->ingest_off_cap) {
        HIPCHK(c, hipDeviceSynchronize());
        if (c->ingest_off) hipHostFree(c->ingest_off);
        c->ingest_off = nullptr;
        c->ingest_off_cap = 0;
        HIPCHK(c, hipHostMalloc((void**)&c->ingest_off, sizeof(int32_t) * off_need, hipHostMallocDefault));
        c->ingest_off_cap = off_need;
    }
    const size_t bytes_all = 16 * (size_t)(h_scan_offsets[n_scans] - h_scan_offsets[0]);
    bool registered = false;
    if ((flags & SCVOD_INGEST_REGISTER) && bytes_all) {  // pin the caller's buffer for the duration of the call
        if (hipHostRegister((void*)(h_xyzi + 4 * (size_t)h_scan_offsets[0]), bytes_all, hipHostRegisterDefault) == hipSuccess)
            registered = true;
        else
            (void)hipGetLastError();  // already pinned or not registrable: the copies still work (staged)
    }
    hipStream_t st = c->stream;
    int rc = SCVOD_OK;
    int32_t* po = c->ingest_off;
    for (int k = 0; k < n_chunks && rc == SCVOD_OK; ++k) {
        const int b = k & 1;
        const int s0 = k * chunk_scans, s1 = s0 + chunk_scans < n_scans ? s0 + chunk_scans : n_scans;
        const int ns = s1 - s0;
        for (int j = 0; j <= ns; ++j) po[j] = h_scan_offsets[s0 + j] - h_scan_offsets[s0];
        const size_t npts = (size_t)po[ns];
        // the buffer is free once the consumers of the chunk two steps back are done
        if (k >= 2 && hipStreamWaitEvent(c->copy_stream, c->ingest_done[b], 0) != hipSuccess) rc = fail(c, SCVOD_ERR_HIP, "hipStreamWaitEvent");
        if (rc == SCVOD_OK && npts &&
            hipMemcpyAsync(c->ingest_buf[b], h_xyzi + 4 * (size_t)h_scan_offsets[s0], 16 * npts, hipMemcpyHostToDevice, c->copy_stream) != hipSuccess)
            rc = fail(c, SCVOD_ERR_HIP, "hipMemcpyAsync (ingest)");
        if (rc == SCVOD_OK && (hipEventRecord(c->ingest_copied[b], c->copy_stream) != hipSuccess ||
                               hipStreamWaitEvent(st, c->ingest_copied[b], 0) != hipSuccess))
            rc = fail(c, SCVOD_ERR_HIP, "ingest event");
        if (rc == SCVOD_OK) rc = run_batch(c, c->ingest_buf[b], po, ns, st, 1, 1, 1, 0, true);
        if (rc == SCVOD_OK && fn) {
            const int frc = fn(user, c, s0, ns, (void*)st);
            if (frc != SCVOD_OK) rc = fail(c, frc, "chunk callback returned %d", frc);
        }
        if (rc == SCVOD_OK && hipEventRecord(c->ingest_done[b], st) != hipSuccess) rc = fail(c, SCVOD_ERR_HIP, "hipEventRecord");
        po += ns + 1;
    }
    hipError_t e1 = hipStreamSynchronize(c->copy_stream), e2 = hipStreamSynchronize(st);
    if (registered) hipHostUnregister((void*)(h_xyzi + 4 * (size_t)h_scan_offsets[0]));
    if (rc == SCVOD_OK && (e1 != hipSuccess || e2 != hipSuccess)) rc = fail(c, SCVOD_ERR_HIP, "ingest: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return rc;
}

int scvod_set_timing(scvod_ctx* c, int32_t enabled) {
    if (!c) return SCVOD_ERR_INVALID;
    c->timing = enabled != 0;
    return SCVOD_OK;
}

int scvod_batch_timings(scvod_ctx* c, const char** names, float* ms, int32_t cap) {
    if (!c) return SCVOD_ERR_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return SCVOD_ERR_HIP;
    if (c->last_stream) hipStreamSynchronize(c->last_stream);
    int n = 0;
    for (size_t i = 0; i < c->tim_used && n < cap; ++i, ++n) {
        float t = 0.f;
        hipEventElapsedTime(&t, c->tim[i].e0, c->tim[i].e1);
        if (names) names[n] = c->tim[i].name;
        if (ms) ms[n] = t;
    }
    return n;
}

int scvod_batch_voxelgrid(scvod_ctx* c, const void* d_xyzi, const uint32_t* d_labels, const int32_t* h_scan_offsets,
                          int32_t n_scans, const float leaf[3], float max_intensity, void* d_out_xyzi, int64_t out_capacity,
                          int32_t* h_out_offsets, void* stream) {
    return run_voxelgrid(c, d_xyzi, d_labels, h_scan_offsets, n_scans, leaf, max_intensity, d_out_xyzi, out_capacity,
                         h_out_offsets, (hipStream_t)stream);
}

int scvod_voxelgrid(scvod_ctx* c, const float* h_xyzi, const uint32_t* h_labels, int32_t n, const float leaf[3],
                    float max_intensity, float* h_out_xyzi, int32_t out_capacity, int32_t* n_out) {
    if (!c || !n_out || n < 0 || (n > 0 && (!h_xyzi || !h_out_xyzi))) return fail(c, SCVOD_ERR_INVALID, "bad arguments");
    *n_out = 0;
    if (n == 0) return SCVOD_OK;
    if (n > c->cap_pts) return fail(c, SCVOD_ERR_CAPACITY, "%d points > capacity %lld", n, (long long)c->cap_pts);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc_side = ensure_side(c, kSideIn | (h_labels ? kSideLabels : 0))) return rc_side;
    HIPCHK(c, hipMemcpyAsync(c->d_in, h_xyzi, sizeof(float) * 4 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    if (h_labels) HIPCHK(c, hipMemcpyAsync(c->d_labels, h_labels, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    float4* d_out = (float4*)c->A.cl_bbox;  // device-side output staging: 24 bytes per point, idle outside cluster typing
    int32_t off[2] = {0, n}, out_off[2] = {0, 0};
    int rc = run_voxelgrid(c, c->d_in, h_labels ? c->d_labels : nullptr, off, 1, leaf, max_intensity, d_out, n, out_off, c->stream);
    if (rc) return rc;
    *n_out = out_off[1];
    if (out_off[1] > out_capacity) return fail(c, SCVOD_ERR_CAPACITY, "output buffer too small (%d < %d)", out_capacity, out_off[1]);
    if (out_off[1]) HIPCHK(c, hipMemcpy(h_out_xyzi, d_out, sizeof(float) * 4 * (size_t)out_off[1], hipMemcpyDeviceToHost));
    return SCVOD_OK;
}

// grow-only device scratch owned by the ctx (no allocation on the steady-state path, nothing to leak on an error return)
static int nn_reserve(scvod_ctx* c, int slot, size_t bytes, void** out) {
    if (bytes > c->nn_cap[slot]) {
        if (c->nn_buf[slot]) hipFree(c->nn_buf[slot]);
        c->nn_buf[slot] = nullptr;
        c->nn_cap[slot] = 0;
        HIPCHK(c, hipMalloc(&c->nn_buf[slot], bytes + bytes / 4));
        c->nn_cap[slot] = bytes + bytes / 4;
    }
    *out = c->nn_buf[slot];
    return SCVOD_OK;
}

static int nn_run(scvod_ctx* c, const float* d_map, int32_t n_map, const float* d_q, int32_t n_query, float radius,
                  int32_t* d_idx, float* d_sq, uint8_t* d_w, const float origin[3], int bounded, hipStream_t st) {
    // grid: cell edge >= radius (so `within` is decided by the 27-cell probe); the origin only shifts the hash
    const float cell = radius > 0.2f ? radius : 0.2f;
    int32_t buckets = 1024;
    while (buckets < 2 * n_map && buckets < (1 << 26)) buckets <<= 1;
    const size_t nm = n_map ? n_map : 1, nq = n_query ? n_query : 1;
    const size_t work_ints = 3 * (size_t)buckets + nm + nq + 8 + (size_t)buckets / 1024 + 1;
    void* d_work = nullptr;
    int rc = nn_reserve(c, 5, work_ints * sizeof(int), &d_work);
    if (rc) return rc;
    launch_nn(d_map, n_map, d_q, n_query, radius, d_idx, d_sq, d_w, origin, cell, buckets, (int*)d_work, bounded, st);
    HIPCHK(c, hipGetLastError());
    return SCVOD_OK;
}

static int nn_search_host(scvod_ctx* c, const float* h_map_xyz, int32_t n_map, const float* h_query_xyz, int32_t n_query, float radius,
                          int32_t* h_nn_idx, float* h_nn_sqdist, uint8_t* h_within, int bounded) {
    if (!c || n_map < 0 || n_query < 0 || (n_map > 0 && !h_map_xyz) || (n_query > 0 && !h_query_xyz))
        return fail(c, SCVOD_ERR_INVALID, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nm = n_map ? n_map : 1, nq = n_query ? n_query : 1;
    float origin[3] = {0.f, 0.f, 0.f};  // min corner of the map (keeps the cell coordinates small)
    for (int i = 0; i < n_map; ++i)
        for (int k = 0; k < 3; ++k)
            if (i == 0 || h_map_xyz[3 * (size_t)i + k] < origin[k]) origin[k] = h_map_xyz[3 * (size_t)i + k];
    void *d_map, *d_q, *d_sq, *d_idx, *d_w;
    int rc;
    if ((rc = nn_reserve(c, 0, nm * 12, &d_map)) || (rc = nn_reserve(c, 1, nq * 12, &d_q)) || (rc = nn_reserve(c, 2, nq * 4, &d_sq)) ||
        (rc = nn_reserve(c, 3, nq * 4, &d_idx)) || (rc = nn_reserve(c, 4, nq, &d_w)))
        return rc;
    hipStream_t st = c->stream;
    if (n_map) HIPCHK(c, hipMemcpyAsync(d_map, h_map_xyz, (size_t)n_map * 12, hipMemcpyHostToDevice, st));
    if (n_query) HIPCHK(c, hipMemcpyAsync(d_q, h_query_xyz, (size_t)n_query * 12, hipMemcpyHostToDevice, st));
    if ((rc = nn_run(c, (const float*)d_map, n_map, (const float*)d_q, n_query, radius, (int32_t*)d_idx, (float*)d_sq, (uint8_t*)d_w,
                     origin, bounded, st)))
        return rc;
    HIPCHK(c, hipStreamSynchronize(st));
    if (n_query) {
        if (h_nn_idx) HIPCHK(c, hipMemcpy(h_nn_idx, d_idx, (size_t)n_query * 4, hipMemcpyDeviceToHost));
        if (h_nn_sqdist) HIPCHK(c, hipMemcpy(h_nn_sqdist, d_sq, (size_t)n_query * 4, hipMemcpyDeviceToHost));
        if (h_within) HIPCHK(c, hipMemcpy(h_within, d_w, (size_t)n_query, hipMemcpyDeviceToHost));
    }
    return SCVOD_OK;
}

int scvod_nn_search(scvod_ctx* c, const float* h_map_xyz, int32_t n_map, const float* h_query_xyz, int32_t n_query,
                    float radius, int32_t* h_nn_idx, float* h_nn_sqdist, uint8_t* h_within) {
    return nn_search_host(c, h_map_xyz, n_map, h_query_xyz, n_query, radius, h_nn_idx, h_nn_sqdist, h_within, 0);
}
int scvod_nn_radius_search(scvod_ctx* c, const float* h_map_xyz, int32_t n_map, const float* h_query_xyz, int32_t n_query,
                           float radius, int32_t* h_nn_idx, float* h_nn_sqdist) {
    if (!(radius > 0.f)) return fail(c, SCVOD_ERR_INVALID, "radius must be positive");
    std::vector<uint8_t> w(n_query > 0 ? n_query : 1);
    return nn_search_host(c, h_map_xyz, n_map, h_query_xyz, n_query, radius, h_nn_idx, h_nn_sqdist, w.data(), 1);
}

int scvod_nn_search_device(scvod_ctx* c, const float* d_map_xyz, int32_t n_map, const float* d_query_xyz, int32_t n_query,
                           float radius, int32_t* d_nn_idx, float* d_nn_sqdist, uint8_t* d_within, void* stream) {
    if (!c || n_map < 0 || n_query < 0 || (n_map > 0 && !d_map_xyz) || (n_query > 0 && (!d_query_xyz || !d_nn_idx || !d_nn_sqdist || !d_within)))
        return fail(c, SCVOD_ERR_INVALID, "bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const float origin[3] = {0.f, 0.f, 0.f};
    return nn_run(c, d_map_xyz, n_map, d_query_xyz, n_query, radius, d_nn_idx, d_nn_sqdist, d_within, origin, 0,
                  stream ? (hipStream_t)stream : c->stream);
}

}  // extern "C"
