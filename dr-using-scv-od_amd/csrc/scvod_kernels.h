// scvod_kernels.h -- launch interface between the C-ABI layer (scvod_capi.hip) and the
// gfx950 kernels (scvod_kernels.hip).  All pointers are device pointers.
#ifndef SCVOD_KERNELS_H_
#define SCVOD_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/scvod.h"
#include "scvod_math.h"

namespace scvod {

constexpr int kMaxPatches = SCVOD_MAX_PATCHES;
constexpr int kMaxBuckets = 1024;
constexpr int kVgLutBins = 16384;
constexpr int kIrrListCap = 256;  // points of a scan with an index triple outside the grid that k_emit lists (Arena::irr_list)
// marks per INPUT point for the static map, which streams the input in order: whether Patchwork kept a point at all follows
// from its patch id (pid) and that patch's population; k_tk_dyn marks the members of dynamic clusters; the two list marks are
// only written when a caller asks for a map without the ground or without the range/FOV rejects
constexpr uint8_t kMapDynamic = 1, kMapGround = 2, kMapRejected = 4, kMapCar = 8;  // (kMapCar: member of a `car` cluster, set by the clustering:
                                                                                    //  the only points a tracking result can take out of the map)

struct Xyz {
    float x, y, z;
};

struct DevParams {
    BinParams bin;
    CzmParams czm;
    KeepFast keep;       // shortcut of the range / FOV verdict (scvod_math.h::keep_of_point)
    BinFast binfast;     // guarded estimate of a transformed point's voxel index (scvod_math.h::voxel_idx_fast)
    int64_t key_off;     // added to voxel_idx before bucketing (R*S + S + 1)
    int32_t vb_shift;    // bucket = clamp((voxel_idx + key_off) >> vb_shift, 0, n_buckets-1)
    int32_t n_buckets;
    int32_t n_patches;
    float max_z, min_z, car_square;  // recognise thresholds (utility.h:294-298)
    int32_t to_be_class;             // utility.h:306
};

// per-patch record produced by the patch kernel, consumed by the emission kernels
struct PatchRec {
    int32_t n;       // points in the patch (pc2czm)
    int32_t n_g;     // size of the ground part after the last iteration
    int32_t status;  // 0 skipped, 1 kept, 2 rejected (tilt), 3 rejected (elevation+flatness)
    int32_t a_g;     // points of the ground part that pass the range/FOV filter of makeApriVec
    int32_t a_ng;    // same for the non-ground part
};

// Device arena of one batch.  Per-point arrays are indexed by scan_off[s] + local index;
// per-scan arrays by s * stride.
// board of k_cc_scan's shared exact re-clustering (scvod_k_cluster.inc, cc_help_loop): header [0] slots taken, [1] scans past their exact phase;
// per slot: [0,1] claim word (round << 32 | next chunk), [2] chunks done by helpers, [3] listed nodes, [4] chunks per round, [6..17] six table pointers of the rounds, [26..35] five more for the passes after them, [18..25] development clocks
constexpr int kCcHelpSlots = 256, kCcHelpHdr = 32, kCcHelpSlotWords = 64;  // (a slot = two 128-byte lines of its own)
constexpr size_t kCcHelpWords = kCcHelpHdr + (size_t)kCcHelpSlots * kCcHelpSlotWords;
constexpr int kCcExactBlocks = 256, kCcExactLeaders = 160;  // grid of k_cc_exact; blocks that lead a listed scan at most (the others help)
struct Arena {
    // inputs
    const float4* pts;
    const int32_t* scan_off;  // [B+1]
    int32_t n_scans;
    int32_t max_scan_pts;
    int64_t total_pts;
    // patchwork
    int16_t* pid;             // [N] patch id or -1
    uint64_t* keys;           // [N] (sortable z << 32 | local idx), patch-major per scan
    uint32_t* seg;            // [N] per patch: [ground part -> | <- non-ground part (stored back to front)], bit31 = passes bin filter
    Xyz* sorted_xyz;          // [N] per patch: points in (z, idx) order, packed 12-byte xyz
    uint32_t* sorted_idx;     // [N] same order: input index | (passes the range/FOV test) << 31
    uint32_t* zkey;           // [N] sortable z key per input point
    float* fit_thd;           // [B][kMaxPatches] th_dist_d_ of the last plane fit
    int4* order;              // [B * kMaxPatches] live patches by descending size class: {scan*1024+patch, n, scan base, patch offset}
    int32_t* order_hist;      // [64]
    int32_t* order_cursor;    // [64]
    int32_t* order_off;       // [65]  ([64] = number of live patches)
    int32_t* patch_count;     // [B][kMaxPatches]
    int32_t* patch_cursor;    // [B][kMaxPatches]
    int32_t* patch_off;       // [B][kMaxPatches+1]
    PatchRec* patch_rec;      // [B][kMaxPatches]
    scvod_patch_plane* planes;  // [B][kMaxPatches]
    int32_t* emit_off;        // [B][kMaxPatches][4]  ground / nonground / apri / rejected
    // outputs of the Patchwork + binning stage
    uint8_t* cls;             // [N] filled per scan on request (scvod_batch_fetch / per-scan API), see k_cls_from_lists
    int32_t* ground_idx;      // [N]
    int32_t* nonground_idx;   // [N]
    scvod_apri* apri;         // [N]
    int32_t* apri_src;        // [N]
    int32_t* apri_key;        // [N] PointAPRI::voxel_idx, compact copy for the voxel stage
    float* apri_int;          // [N] PointAPRI::intensity, compact copy for the voxel stage
    int32_t* apri_idx3;       // [N] PointAPRI::{range,sector,azimuth}_idx packed 11+11+10 bits (clustering)
    int32_t* rejected_src;    // [N]
    int32_t* counts;          // [B][8]
    int32_t* scan_irr;        // [B] != 0: the scan holds an index triple outside the grid (set by the binning kernels; only orders the clustering)
    int32_t* irr_list;        // [B][kIrrListCap + 1] [0] = how many points of the scan have an index triple outside the grid (-1: not listed by the
                              //   kernel that binned the batch), then their apri indices: k_emit -> scvod_lastname.hip
    int32_t* cc_perm;         // [B] order in which k_cc_scan takes the scans: the irregular ones (the long-running workgroups) first
    // voxel stage
    int32_t* vb_count;        // [B][kMaxBuckets]
    int32_t* vb_off;          // [B][kMaxBuckets+1]
    int32_t* vb_nvox;         // [B][kMaxBuckets]
    int32_t* vox_off;         // [B][kMaxBuckets+1]
    int4* vorder;             // [B * kMaxBuckets] non-empty buckets by descending size class (same item layout)
    int32_t* vorder_hist;     // [64]
    int32_t* vorder_cursor;   // [64]
    int32_t* vorder_off;      // [65]
    uint64_t* vkeys;          // [N] (biased voxel key << 32 | apri idx), bucket-major per scan; or 32-bit keys (vx_k32)
    int32_t vx_k32;           // 1: the voxel stage of this batch sorts 32-bit keys: (key - bucket's first key) << vx_idx_bits | apri idx
                              //    (range/FOV-filtered keys only: every key lies inside its bucket's range; vb_shift + vx_idx_bits <= 32)
    int32_t vx_idx_bits;      // bits of an apri index inside one scan of this batch
    int32_t* tmp_vox_key;     // [N] per-bucket voxel records before compaction
    int32_t* tmp_vox_begin;   // [N]
    float* tmp_vox_av;        // [N]
    float* tmp_vox_cov;       // [N]
    int32_t* vox_key;         // [N]
    int32_t* vox_pt_begin;    // [N + B]  (n_vox + 1 entries per scan, base scan_off[s] + s)
    int32_t* vox_pts;         // [N]
    float* vox_av;            // [N]
    float* vox_cov;           // [N]
    // clustering (connected components of occupied voxels)
    int32_t cc_exact_max;     // generic clustering variant: nodes of components with irregular runs that are re-clustered exactly
                              //   (default 4096; scvod_set_cluster_exact(ctx, 1 or 2) lifts it to "any")
    int32_t cc_plain_rule;    // 1: irregular runs that the cells around them settle are left as found (cc_run_is_plain); 0: every one is re-clustered
    int32_t* cc_stats;        // [8] per clustering call: [4] irregular runs settled by the rule, [5] the others; [0..3] scans that kept "everything found is joined" for a component, nodes of
                              //   those components (an upper bound from a sample when they are not even listed), 0, scans of the generic variant
                              //   whose z-planes are too large for the windowed search (forest in HBM)
    int32_t* cc_again;        // [1 + B] scans k_cc_scan hands over to k_cc_exact: [0] how many, then the scans; [0] cleared per launch
    int32_t* cc_help;         // [kCcHelpWords] board of the exact re-clustering rounds that leaders share with helper blocks (scvod_k_cluster.inc: cc_help_loop); cleared per launch
    int32_t cc_help_blocks;   // > 0: the blocks of k_cc_exact that lead no scan help (set per launch by launch_cluster)
    int32_t cc_help_blocks_wanted;  // what the ctx asks for (0 = leaders always work alone: tests / A-B runs)
    int32_t* cc_parent;       // [N] union-find forest over apri indices (scan-local)
    uint8_t* cc_touched;      // [N] per voxel slot: appeared in a neighbourhood
    int32_t* pt_voxel;        // [N] voxel slot of every apri point
    int32_t* pt_cluster;      // [N] canonical cluster name = smallest apri index of the component
    uint32_t* cl_bbox;        // [7N] clustering scratch: bounding-box records of the scans that do not fit the LDS
    int32_t* cl_count;        // [N] per cluster root: number of points
    uint8_t* pt_type;         // [N] per apri point: 0 erased, 1 other, 2 car
    // the cluster that still carries Frame::max_name as ssc.cpp:354 stores it (scvod_lastname.hip)
    int32_t* cc_last;         // [B][4] {canonical name or -1, lowest voxel slot whose first point belongs to it or -1,
                              //         status: 0 exact, 1 a replay did not fit the LDS, 2 too many index triples outside the grid, events replayed}
    int32_t* cc_redo;         // 4 x [B + 1] scans listed by the triage for the pass with the small / mid / large tables, [B] = how many
    int32_t* ln_state;        // [B][kLnStateWords] what the triage leaves a scan's follow-up pass: the set of classes, the irregular points, the class table
    int32_t* ln_prof;         // [B][8] phase clocks (10 ns ticks) and counts of the last pass over a scan: tools/lastname_lat.py
    int32_t* ln_prof2;        // [B][8] the largest class walked: nodes, Jacobi rounds, clocks of build / rounds / openers / partition / walk, events
    int32_t* ln_stats;        // [4] per clustering call: scans with status 1, with status 2, 0, 0
    // sequence differencing on the device (scvod_batch_track, scvod_track.hip)
    int4* vox_track;          // [N] per voxel: {key, label = cluster root of its points or -1, |occupy_voxels| of that
                              //     cluster, its type}: the table the probe of the PREVIOUS scan runs against; also the
                              //     boundary message between sequence shards (scvod_batch_export_table)
    int32_t* vox_rep;         // [N] per voxel: lowest voxel slot of the scan that carries the same label (-1: unlabelled): the
                              //     label's id in the sequential tracking chain (scvod_chain.hip)
    int32_t* tk_crep;         // [N] per scan: that id for each car cluster, in the order of tk_clusters
    int32_t* tk_prep;         // [N] per cluster region, parallel to tk_pairs: the id of the pair's label
    int8_t* cl_state;         // [N] per cluster root: Cluster::state (-1 untouched, 0 static, 1 dynamic)
    int32_t* tk_mbegin;       // [N] per car root: first slot of its members in tk_members (scan-local)
    int32_t* tk_cursor;       // [N] per car root: scatter cursor
    int32_t* tk_members;      // [N] per scan: apri indices of the car points, grouped by cluster
    int32_t* tk_hit;          // [N] per member slot: slot of the next table hit by the transformed point, or -1
    int32_t* tk_uniq;         // [N] per cluster region: sorted unique hit slots (sampleVec, ssc.cpp:1319-1321)
    int32_t* tk_nuniq;        // [N] per car root
    int2* tk_pairs;           // [N] per cluster region: (next label, unique voxels) = remap_name (ssc.cpp:1275,1304-1316)
    int32_t* tk_npairs;       // [N] per car root: remap_name.size()
    int32_t* tk_clusters;     // [N] per scan: roots of its car clusters, ascending
    int32_t* tk_scan;         // [B][4] per scan: car clusters, car points, dynamic clusters, dynamic points
    uint8_t* pt_dyn;          // [N] per apri point: SCVOD_DYN_*
    uint8_t* pt_mapcls;       // [N] per INPUT point, for the static map: kMap* bits
    // loader-side VoxelGrid
    int32_t* vg_par;          // [B][16] per scan: min_b[3], mul[3], overflow flag, kept points, distinct cells
    int32_t* vg_range;        // [1] largest cell index range of the batch
    int32_t* vg_outoff;       // [B+1] output offsets (uploaded by the host between the two phases)
    uint16_t* vb_lut;         // [B][kVgLutBins] monotone key-bin -> bucket table of the VoxelGrid run; nullptr in the hot path
    const int32_t* vb_lut_shift;  // device word: key >> *vb_lut_shift = bin (k_vg_lut derives it from the batch's index range)
    const uint32_t* vg_labels;    // VoxelGrid run only: per input point label (or nullptr) and the loader's intensity scale
    float vg_max_intensity;
};

struct TrackJob {          // scan-vs-next-scan probe
    const float4* pts;         // explicit cluster points, or nullptr when gathered from apri
    const int32_t* members;    // apri indices (batch mode) or nullptr
    const int32_t* pt_cluster_begin;  // [n_clusters+1] offsets into pts / members
    int32_t n_clusters;
    int32_t n_pts;
    // per cluster: which transform / which source scan / which next table
    const int32_t* cluster_pair;   // [n_clusters] pair index (0 for the single-pair API)
    const float* T;                // [n_pairs][12]
    const int32_t* pair_pt_begin;  // batch mode: [n_pairs+1] first point (offset into members) of every pair
    int32_t n_pairs, max_pair_pts;
    // next tables: explicit (single pair) or arena scans (batch)
    const int32_t* next_keys;      // explicit table or nullptr
    const int32_t* next_labels;    // explicit labels or nullptr (= all labelled)
    int32_t n_next_vox;
    // outputs
    int32_t* hit_slot;     // [n_pts]
    uint64_t* work;        // [n_pts] sort workspace
    int32_t* uniq_slots;   // [n_pts] per cluster region starts at pt_cluster_begin[c]
    int32_t* uniq_count;   // [n_clusters]
};

struct TrackBatch {            // scvod_batch_track: every scan of the batch against its successor
    const int32_t* next_scan;      // [B] successor of scan s: index in the batch, -1 = none, <= -2 = external table -2 - v
    const int4* const* ext_tables; // device array of external tables: record 0 = {n_voxels, 0, 0, 0}, then vox_track records
    int32_t n_ext;
    const float* T;                // [B][12] trans_next^-1 * trans_pre of (s, successor)
    float occupancy;               // ssc/occupancy_
};

typedef void (*TimerHook)(void* user, const char* name, int begin);

// Launches.  `th`/`tu` optional per-kernel timing hook (called before and after each launch).
// do_patchwork: 1 = Patchwork + fused binning, 0 = binning of the input cloud in input order,
// 2 = neither (apri / counts already in the arena: voxel stage only).
void launch_process(const DevParams& P, const Arena& A, hipStream_t st, int do_patchwork, int apply_filter,
                    int do_voxels, TimerHook th, void* tu);
void launch_apri_expand(const DevParams& P, const Arena& A, int s0, int n_scans, int max_pts, hipStream_t st);
struct VgJob {                // SSC::getCloud label filter + pcl::VoxelGrid (ssc.cpp:1063-1076, 1103-1106)
    const uint32_t* labels;   // per input point, or nullptr (no filter, no intensity scaling)
    float max_intensity;
    float inv_leaf[3];        // 1.f / leaf, fp32 like Eigen::Array4f::Ones() / leaf_size_
    float4* out;              // caller's output buffer
};
void launch_voxelgrid_keys(const Arena& A, const VgJob& J, hipStream_t st);
void launch_voxelgrid_lut(const Arena& A, hipStream_t st);
void launch_voxelgrid_gather(const DevParams& P, const Arena& A, const VgJob& J, long long out_capacity, hipStream_t st);
void launch_cls(const Arena& A, int s, size_t scan_base, int n_points, hipStream_t st);
void launch_cluster(const DevParams& P, const Arena& A, int from_apri, hipStream_t st, TimerHook th, void* tu);
void launch_lastname(const DevParams& P, const Arena& A, hipStream_t st, hipStream_t st2, hipStream_t st3, hipEvent_t ev_fork, hipEvent_t ev_join2,
                     hipEvent_t ev_join3, TimerHook th, void* tu);
void launch_track(const DevParams& P, const Arena& A, const TrackJob& J, int batch_mode, hipStream_t st,
                  TimerHook th, void* tu);
struct ChainJob;
void launch_track_batch(const DevParams& P, const Arena& A, const TrackBatch& J, int from_apri, int phases, hipStream_t st,
                        TimerHook th, void* tu, const ChainJob* chain = nullptr, hipEvent_t before_chain = nullptr);
void launch_track_dyn(const Arena& A, int from_apri, hipStream_t st);  // per-point bytes from the cluster states (again, after a resume)
void launch_export_table(const Arena& A, int s, int4* out, long long cap_records, hipStream_t st);
void launch_nn(const float* map_xyz, int32_t n_map, const float* q_xyz, int32_t n_q, float radius, int32_t* nn_idx,
               float* nn_sq, uint8_t* within, const float origin[3], float cell, int32_t buckets, int* work, int bounded,
               hipStream_t st);

}  // namespace scvod
#endif
