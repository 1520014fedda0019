/*
 * scvod.h -- C-ABI of the MI355X-native SCV-OD hot path (libscvod.so).
 *
 * This is the drop-in boundary for the hot path of Yixin-F/DR-Using-SCV-OD
 * (SURVEY.md section 8b).  The reference has no FFI of its own: the seam is plain
 * C++ member calls on `class SSC : public Utility` (include/ssc.h:7) and
 * `template<class PointT> class PatchWork` (include/patchwork.h:37-191).  Each
 * entry point below names the reference member it replaces (file:line into
 * /root/reference).  The C++ facade in dr-using-scv-od_amd/host/ keeps the
 * reference's class signatures and calls only the functions declared here.
 *
 * Conventions
 *   - every function returns int status: 0 = SCVOD_OK, <0 = error (no exceptions,
 *     no logging, no allocation visible to the caller except through the ctx);
 *     scvod_last_error(ctx) returns a human-readable message.
 *   - plain pointers and sizes only.  "h_" pointers are host memory owned by the
 *     caller, "d_" pointers are device (HBM) memory owned by the caller.
 *   - a ctx is single-owner and not thread-safe (the reference caller is
 *     single-threaded, src/main.cpp:9-12); one ctx per GPU / stream.
 *   - points are packed float4 {x, y, z, intensity} (pcl::PointXYZI without the
 *     padding, 16 B/point).
 *   - the library is GPU-only: if no gfx950 device / HIP runtime is usable,
 *     scvod_create fails with SCVOD_ERR_NO_DEVICE.  There is no CPU fallback.
 */
#ifndef SCVOD_H_
#define SCVOD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCVOD_OK 0
#define SCVOD_ERR_INVALID (-1)   /* bad argument / parameter combination           */
#define SCVOD_ERR_NO_DEVICE (-2) /* no usable HIP device (no CPU fallback exists)   */
#define SCVOD_ERR_HIP (-3)       /* HIP runtime error, see scvod_last_error         */
#define SCVOD_ERR_CAPACITY (-4)  /* batch larger than the ctx capacity              */
#define SCVOD_ERR_STATE (-5)     /* call order violated (e.g. fetch before run)     */

/* One scan holds at most 2^19 points (a 128-beam x 2048-column sweep is 262144): sort keys carry the point index in
 * 19 bits.  Larger scans are refused with SCVOD_ERR_CAPACITY. */
#define SCVOD_MAX_SCAN_POINTS 524288
/* A ctx holds at most 2^31 - 65 points in total (int32 scan offsets); in practice the ≈ 310 B/point arena of a 288 GB
 * device ends near 0.85 G points (scvod_arena_bytes). */

#define SCVOD_NUM_ZONES 4
#define SCVOD_MAX_PATCHES 1024 /* reference model has 504 (patchwork.h:48-49) */

/* per-point class written by Patchwork (patchwork.h:326-391) */
#define SCVOD_CLS_GROUND 0    /* emitted into cloud_out                         */
#define SCVOD_CLS_NONGROUND 1 /* emitted into cloud_nonground                   */
#define SCVOD_CLS_DROPPED 2   /* z < -1.8h, r outside (2.7, 80], patch size <= 10 */

/* ---- configuration --------------------------------------------------------------- */

/* The `ssc/` keys of config/<name>.yaml that the hot path reads (include/utility.h:283-313;
 * defaults are the nh.param<> defaults there).  Names drop the trailing underscore. */
typedef struct scvod_params {
    float sensor_height; /* 2.0  */
    float min_dis;       /* 0.0  */
    float max_dis;       /* 50.0 */
    float min_angle;     /* 0.0  */
    float max_angle;     /* 360  */
    float min_azimuth;   /* -30  */
    float max_azimuth;   /* 60   */
    float range_res;     /* 0.2  */
    float sector_res;    /* 1.2  */
    float azimuth_res;   /* 2.0  */
    float occupancy;     /* 0.6  */
    /* keys of the bounding-box refine / recognise step (ssc.cpp:437-467, 849-872), used only by
     * scvod_batch_cluster_types */
    float max_z;         /* 1.0  */
    float min_z;         /* -1.0 */
    float car_square;    /* 2.0  */
    int32_t toBeClass;   /* 1    */
    int32_t reserved;
} scvod_params;

/* Patchwork constants.  Hard-coded in the reference (patchwork.h:48-51, :115-129);
 * exposed here as a defaulted struct (scvod_pw_params_default). */
typedef struct scvod_pw_params {
    int32_t num_iter;                                /* 3  */
    int32_t num_lpr;                                 /* 20 */
    int32_t num_min_pts;                             /* 10 */
    int32_t num_rings_of_interest;                   /* 4 (= elevation_thr_.size(), patchwork.h:73) */
    int32_t num_sectors_each_zone[SCVOD_NUM_ZONES];  /* 16,32,54,32 */
    int32_t num_rings_each_zone[SCVOD_NUM_ZONES];    /* 2,4,4,4     */
    double th_seeds;                                 /* 0.3   */
    double th_dist;                                  /* 0.1   */
    double max_range;                                /* 80.0  */
    double min_range;                                /* 2.7   */
    double uprightness_thr;                          /* 0.707 */
    double adaptive_seed_selection_margin;           /* -1.1  */
    double elevation_thr[4];                         /* -1.2,-0.9984,-0.851,-0.605 */
    double flatness_thr[4];                          /* 0,0.000125,0.000185,0.000185 */
} scvod_pw_params;

void scvod_params_default(scvod_params* p);       /* utility.h:283-310 defaults */
void scvod_pw_params_default(scvod_pw_params* p); /* patchwork.h:48-51,115-129  */

/* Curved-voxel grid sizes exactly as SSC::SSC computes them (src/ssc.cpp:36-39):
 * (int)std::ceil((max - min) / res) in float. */
void scvod_grid_dims(const scvod_params* p, int32_t* range_num, int32_t* sector_num,
                     int32_t* azimuth_num, int32_t* bin_num);

/* ---- PODs -------------------------------------------------------------------------- */

/* struct PointAPRI, include/utility.h:96-106 (44 bytes, same field order) */
typedef struct scvod_apri {
    float x, y, z;
    float range;
    float angle;
    float azimuth;
    float intensity;
    int32_t range_idx;
    int32_t sector_idx;
    int32_t azimuth_idx;
    int32_t voxel_idx;
} scvod_apri;

/* Per-patch plane record written by the Patchwork kernel (state of normal_, pc_mean_,
 * singular_values_ after extract_piecewiseground, patchwork.h:339-343). */
typedef struct scvod_patch_plane {
    float normal[3];
    float mean[3];
    float sv[3];
    int32_t n_pts;      /* points binned into the patch (pc2czm)                   */
    int32_t n_ground;   /* |regionwise_ground_| after the last iteration            */
    int32_t status;     /* 0 skipped (size<=num_min_pts), 1 kept, 2 rejected: tilt,
                           3 rejected: elevation+flatness                           */
} scvod_patch_plane;

/* Result of one scan.  All pointers are host memory owned by the ctx, valid until the
 * next call that produces a scvod_scan_result on the same ctx. */
typedef struct scvod_scan_result {
    int32_t n_points;
    int32_t n_ground;     /* |cloud_out|        patchwork.h:362,377,381           */
    int32_t n_nonground;  /* |cloud_nonground|  patchwork.h:348-349,363,373-374   */
    int32_t n_dropped;
    int32_t n_apri;       /* |apri_vec| == |cloud_use|  ssc.cpp:174,193           */
    int32_t n_rejected;   /* pushes into cloud_eva_static, ssc.cpp:161-172        */
    int32_t n_voxels;     /* |hash_cloud|        ssc.cpp:253-280                  */
    int32_t n_patches;
    const uint8_t* cls;           /* [n_points] SCVOD_CLS_*                        */
    const int32_t* ground_idx;    /* [n_ground] input index of cloud_out[k]        */
    const int32_t* nonground_idx; /* [n_nonground] input index of cloud_nonground[k] */
    const scvod_patch_plane* planes; /* [n_patches] in (zone, ring, sector) order  */
    const scvod_apri* apri;       /* [n_apri] apri_vec                             */
    const int32_t* apri_src;      /* [n_apri] input index of apri_vec[k]/cloud_use[k] */
    const int32_t* rejected_src;  /* [n_rejected] input index, push order          */
    /* hash_cloud as CSR, voxels sorted by ascending voxel_idx key */
    const int32_t* vox_key;       /* [n_voxels] Voxel key (PointAPRI::voxel_idx)   */
    const int32_t* vox_pt_begin;  /* [n_voxels+1] offsets into vox_pts             */
    const int32_t* vox_pts;       /* [n_apri] Voxel::ptIdx, ascending per voxel    */
    const float* vox_av;          /* [n_voxels] Voxel::intensity_av  ssc.cpp:283   */
    const float* vox_cov;         /* [n_voxels] Voxel::intensity_cov ssc.cpp:284-287 */
} scvod_scan_result;

/* ---- context ------------------------------------------------------------------------- */

typedef struct scvod_ctx scvod_ctx;

/* max_points_total: capacity of the device arena in points summed over a batch;
 * max_scans: capacity in scans per batch.  pw may be NULL (defaults). */
int scvod_create(const scvod_params* params, const scvod_pw_params* pw, int device,
                 int64_t max_points_total, int32_t max_scans, scvod_ctx** out);
void scvod_destroy(scvod_ctx* ctx);
/* the parameters the ctx was created with */
int scvod_get_params(const scvod_ctx* ctx, scvod_params* out);
const char* scvod_last_error(const scvod_ctx* ctx);
/* bytes of HBM held by the ctx arena */
int64_t scvod_arena_bytes(const scvod_ctx* ctx);

/* ---- per-scan host entry points (what the SSC / PatchWork facade calls) ---------------- */

/* Replaces SSC::process up to and including makeHashCloud (src/ssc.cpp:224-241):
 * PatchWork::estimate_ground (patchwork.h:277-398) -> SSC::makeApriVec (ssc.cpp:155-195)
 * -> SSC::makeHashCloud (ssc.cpp:253-289).  h_xyzi: n x {x,y,z,intensity}. */
int scvod_process_scan(scvod_ctx* ctx, const float* h_xyzi, int32_t n, scvod_scan_result* out);

/* Replaces PatchWork::estimate_ground only (patchwork.h:105-109).  Fills the Patchwork
 * fields of `out`; the apri / voxel fields are zero. */
int scvod_patchwork(scvod_ctx* ctx, const float* h_xyzi, int32_t n, scvod_scan_result* out);

/* Replaces SSC::makeApriVec (ssc.cpp:155-195) on an arbitrary cloud (no Patchwork):
 * with apply_filter != 0 the range/FOV rejection of ssc.cpp:161-172 is applied,
 * with apply_filter == 0 every point is binned unclamped as in SSC::tracking
 * (ssc.cpp:1280-1286).  Followed by SSC::makeHashCloud when with_voxels != 0. */
int scvod_bin_scan(scvod_ctx* ctx, const float* h_xyzi, int32_t n, int32_t apply_filter,
                   int32_t with_voxels, scvod_scan_result* out);

/* Replaces SSC::makeHashCloud (ssc.cpp:253-289) on an apri_vec the caller already holds.  Fills the
 * voxel fields of `out` (and echoes the apri fields). */
int scvod_voxelize(scvod_ctx* ctx, const scvod_apri* h_apri, int32_t n, scvod_scan_result* out);

/* T = getTransformation(next)^-1 * getTransformation(pre), src/ssc.cpp:1255-1257
 * (pcl::getTransformation + Eigen::Affine3f inverse/product restated on the host).
 * pose = {x, y, z, roll, pitch, yaw}; T_out is row-major 3x4. */
void scvod_pose_delta(const float pose_pre[6], const float pose_next[6], float T_out[12]);

/* Bulk part of SSC::tracking (ssc.cpp:1274-1321): for every cluster c (points
 * h_xyzi[offsets[c] .. offsets[c+1]) ), transform by T (utility.h:394-406), re-bin
 * without range/FOV rejection (ssc.cpp:1280-1286), probe the next frame's voxel table
 * (sorted keys + labels) and keep hits whose label != -1 (ssc.cpp:1304-1305).
 *   h_hit_slot   [n_pts]        slot in the next table of the voxel hit by the point, or -1
 *   h_uniq_slots [n_pts]        per cluster: sorted unique hit slots (sampleVec, ssc.cpp:1319-1321)
 *   h_uniq_begin [n_clusters+1] offsets into h_uniq_slots
 * The label grouping / occupancy-ratio decisions (ssc.cpp:1323-1421) stay on the host
 * because they mutate the next frame sequentially. */
int scvod_track_probe(scvod_ctx* ctx, const float* h_xyzi, const int32_t* h_offsets,
                      int32_t n_clusters, const float T[12], const int32_t* h_next_keys,
                      const int32_t* h_next_labels, int32_t n_next_vox, int32_t* h_hit_slot,
                      int32_t* h_uniq_slots, int32_t* h_uniq_begin);

/* ---- loader step in front of the path ------------------------------------ */

/* SSC::getCloud's label filter + intensity scaling (src/ssc.cpp:1063-1076: points whose label & 0xFFFF is 0 or 1 are
 * skipped, intensity *= max_intensity) followed by pcl::VoxelGrid<pcl::PointXYZI>::filter with leaf (lx, ly, lz)
 * (src/ssc.cpp:1103-1106: 0.08 m; PCL 1.8.1 semantics incl. the "leaf size too small -> output = input" branch).
 * labels == NULL: VoxelGrid only (no filter, no scaling).  Output: one xyzi centroid per occupied cell in ascending
 * cell index, the cell's points summed in ascending input index (std::sort leaves that order unspecified in PCL).
 * d_* pointers are device memory; h_out_offsets[n_scans+1] receives the output offsets (in points).  The ctx's arena
 * is reused: results of an earlier scvod_batch_process are invalidated.  Synchronous. */
int scvod_batch_voxelgrid(scvod_ctx* ctx, const void* d_xyzi, const uint32_t* d_labels,
                          const int32_t* h_scan_offsets, int32_t n_scans, const float leaf[3],
                          float max_intensity, void* d_out_xyzi, int64_t out_capacity,
                          int32_t* h_out_offsets, void* stream);

/* The same for one scan in host memory. */
int scvod_voxelgrid(scvod_ctx* ctx, const float* h_xyzi, const uint32_t* h_labels, int32_t n,
                    const float leaf[3], float max_intensity, float* h_out_xyzi,
                    int32_t out_capacity, int32_t* n_out);

/* ---- device-resident batch entry points (sequence shards; used by bench.py) ------------- */

/* Run Patchwork -> binning -> voxel descriptors over n_scans scans already resident in
 * HBM.  d_xyzi: all scans concatenated; h_scan_offsets[n_scans+1] point offsets.
 * stream: hipStream_t (NULL = the ctx's own stream).  Asynchronous w.r.t. the host
 * unless sync != 0.  d_xyzi must stay valid and unchanged until the next batch call: apri_vec is kept
 * on the device in compact form (source index, voxel key, intensity) and the PointAPRI records,
 * the per-point class array and the tracking probe read the points through it on request. */
int scvod_batch_process(scvod_ctx* ctx, const void* d_xyzi, const int32_t* h_scan_offsets,
                        int32_t n_scans, void* stream, int32_t sync);

/* Per-scan counters of the last batch: out[n_scans][8] =
 * {n_points, n_ground, n_nonground, n_dropped, n_apri, n_rejected, n_voxels, 0}. */
int scvod_batch_counts(scvod_ctx* ctx, int32_t* h_out);

/* Download the full result of scan `s` of the last batch. */
int scvod_batch_fetch(scvod_ctx* ctx, int32_t s, scvod_scan_result* out);

/* Scan-vs-next-scan differencing of the whole batch on the device: SSC::tracking (src/ssc.cpp:1250-1426) for every scan
 * against its successor, the way SSC::segDF drives it (src/ssc.cpp:1449-1451).  Needs scvod_batch_cluster and
 * scvod_batch_cluster_types of the same batch: the clusters walked are those whose type is `car` (ssc.cpp:1262), the
 * successor's Voxel::label is the cluster of the voxel's points, -1 where refineClusterByBoundingBox erased it
 * (ssc.cpp:461-466), Cluster::occupy_voxels.size() the number of voxels carrying a label (ssc.cpp:382-392).
 *   h_T          [n_scans][12]  trans_next^-1 * trans_pre of (s, successor) (scvod_pose_delta); unused rows ignored
 *   h_next_scan  [n_scans] or NULL: successor of scan s -- an index of the batch, -1 = none (last scan of a sequence),
 *                -2 - e = the external table e.  NULL: s + 1, none for the last scan.
 *   h_ext_tables [n_ext] device pointers to tables written by scvod_batch_export_table on ANOTHER shard (the first scan of
 *                the next block of the sequence): how a block's last scan is tracked across a shard boundary.
 * Per cluster: the transformed points are re-binned without range/FOV rejection (ssc.cpp:1280-1286) and looked up in the
 * successor's table; hits are grouped by label and de-duplicated (remap_name, ssc.cpp:1304-1321); Cluster::state follows
 * ssc.cpp:1323-1397, first against the successor in its freshly segmented state (all pairs in parallel), then -- mode
 * SCVOD_TRACK_CHAIN, the default -- with the re-labelling / cloud appending of ssc.cpp:1354-1372, 1378-1384, 1399-1419
 * replayed in sequence order on the device (scvod_set_track_mode below).  Per apri point a SCVOD_DYN_* byte.
 * No host synchronisation inside the call; h_T / h_next_scan / h_ext_tables are copied (and re-uploaded only when they
 * differ from the previous call). */
#define SCVOD_DYN_STATIC 0      /* member of a cluster that is not dynamic                                        */
#define SCVOD_DYN_DYNAMIC 1     /* member of a `car` cluster with state == 1                                      */
#define SCVOD_DYN_UNCLUSTERED 2 /* its cluster was erased by the bounding-box refine (listed static, ssc.cpp:450-454) */
int scvod_batch_track(scvod_ctx* ctx, const float* h_T, const int32_t* h_next_scan, const void* const* h_ext_tables,
                      int32_t n_ext, void* stream, int32_t sync);

/* What scvod_batch_track decides (default SCVOD_TRACK_CHAIN).
 *   SCVOD_TRACK_CHAIN        the reference's SEQUENTIAL chain (SSC::segDF, src/ssc.cpp:1449-1451): tracking(i, i + 1) appends
 *                            the transformed cloud of a static car cluster to its successor cluster (ssc.cpp:1378-1384),
 *                            splits hit voxels off a non-car cluster (ssc.cpp:1351-1372) or fuses the car clusters it
 *                            hits (ssc.cpp:1396-1419) BEFORE tracking(i + 1, i + 2) walks that frame.  Cluster states, the
 *                            per-point bytes and the dynamic counters of every scan with a successor in the batch are
 *                            those of that loop, cluster_set walked in ascending canonical name (created clusters after
 *                            the original ones, in creation order).  On the device the chain of each sequence is cut into
 *                            segments of `segment_steps` steps walked concurrently, each warmed up `warmup_steps` steps
 *                            earlier; a segment whose warm-up did not reproduce the state its predecessor really ended
 *                            in is walked again from that state, so the result never depends on the two lengths
 *                            (segment_steps 0 = the shortest segment whose walkers still fit the device one per CU (<= 256), the default;
 *                            warmup_steps -1 = keep; by default the warm-up is chosen per stream: 10 steps, two more for the next batch whenever more than a few segments had to be walked again, up to 16).  A scan tracked against an EXTERNAL table ends its
 *                            chain: across shard boundaries the decision is first-order -- keep a sequence on one shard.
 *   SCVOD_TRACK_FIRST_ORDER  every cluster against its successor's fresh segmentation (all pairs independent).
 * n_unique / pair_* of scvod_track_result always describe a cluster's OWN points against the fresh successor. */
#define SCVOD_TRACK_CHAIN 1
#define SCVOD_TRACK_FIRST_ORDER 0
#define SCVOD_TRACK_CHAIN_GENERIC 3 /* testing: the chain with every step through the kernel's generic (HBM-resident) step */
int scvod_set_track_mode(scvod_ctx* ctx, int32_t mode, int32_t segment_steps, int32_t warmup_steps);
/* Chain mode needs CHAINS: every scan is the successor of at most one scan of the batch and the table holds no cycle
 * (SCVOD_ERR_INVALID otherwise).  Many scans against one reference scan is a first-order question: SCVOD_TRACK_FIRST_ORDER.
 * The chain's workspace is a separate allocation made by the first scvod_batch_track that needs it (walkers x ~70 bytes x the pool
 * capacity: 16 GB for a seq-05 job, NOT part of scvod_arena_bytes); scvod_chain_workspace_bytes reports it. */
int64_t scvod_chain_workspace_bytes(scvod_ctx* ctx);
/* Points of appended clouds one segment's state can hold (0 = default: 8 x the largest scan of the batch).  The reference
 * appends a static car cluster's whole cloud to its successor at every step (ssc.cpp:1381), so a cloud grows for as long
 * as the object is tracked; a state that outgrows the capacity is reported (SCVOD_ERR_CAPACITY) by scvod_batch_fetch_track /
 * scvod_batch_track_stats, never truncated silently.  Workspace = segments x ~70 bytes x this number. */
int scvod_set_chain_capacity(scvod_ctx* ctx, int64_t pool_points);
/* ONE sequence over several shards (SSC::segDF's loop #1, ssc.cpp:1435-1445, is per scan; loop #2, :1449-1451, is a chain).
 * A shard processes a block of scans plus a HALO of earlier ones (warm-up steps x tracking stride) and one successor per chain
 * behind the block.  scvod_set_track_owned(first): scans below `first` are that halo -- walked over as a warm-up, never decided
 * (their per-point bytes stay those of the first-order pass and belong to the shard before).  After scvod_batch_track:
 *   scvod_batch_track_chains      h_first_scan[k] = first scan of chain k (which interleaved sub-sequence it is), returns the count
 *   scvod_chain_state_bytes       size of a boundary-state record of this job
 *   scvod_chain_export_state      which = 1: the state chain k ENDED in (a device buffer: send it to the shard that owns the next
 *                                 block); which = 0: the state the chain assumed at its first own step (its warm-up's snapshot).
 *                                 Record = int32 {entries, carried points, parts, valid}, entries, parts, carried points; valid = 1,
 *                                 0 (the chain has no such state) or 2 (the state needs more than cap_bytes: nothing but the header is written)
 *   scvod_batch_track_resume      h_d_states[k] = device pointer of the record the previous shard sent for chain k (NULL: none):
 *                                 compared with that snapshot (entries and parts bit for bit, the carried points per entry as a
 *                                 multiset: their order depends on where a walk started and nothing reads it); a chain whose warm-up did not reproduce it is walked
 *                                 again from the received state (and verified / walked on segment by segment, like inside one
 *                                 shard), then the per-point bytes are rebuilt.  scvod_batch_track_stats counts the checks and walks.
 *   scvod_batch_track_compare     the comparison alone: *h_differ = chains whose warm-up did NOT reproduce the received record
 *                                 (they would be walked again); nothing is changed.  Lets the shards of a job exchange their
 *                                 end states all at once and fall back to one-after-the-other only behind the first shard
 *                                 that reports a difference (pyshim/shard.py resolve_chain_boundaries).  Synchronises.
 * The result is the single-shard chain's, whatever the halo length.  A shard resumes at most once per scvod_batch_track. */
int scvod_set_track_owned(scvod_ctx* ctx, int32_t first_owned_scan);
/* the same per scan (a shard that holds blocks of several sequences): h_is_halo[s] != 0 marks scan s as halo; a chain's halo
 * scans must be its first ones */
int scvod_set_track_halo(scvod_ctx* ctx, const uint8_t* h_is_halo, int32_t n_scans);
int scvod_batch_track_chains(scvod_ctx* ctx, int32_t* h_first_scan, int32_t cap);
int64_t scvod_chain_state_bytes(scvod_ctx* ctx);
int scvod_chain_export_state(scvod_ctx* ctx, int32_t chain, int32_t which, void* d_dst, int64_t cap_bytes, void* stream);
int scvod_batch_track_resume(scvod_ctx* ctx, const void* const* h_d_states, int32_t n_states, void* stream, int32_t sync);
int scvod_batch_track_compare(scvod_ctx* ctx, const void* const* h_d_states, int32_t n_states, int32_t* h_differ, void* stream);
/* The comparison without a word read on the host: the chains whose warm-up did NOT reproduce the received record are ADDED to
 * *d_differ, a device word the caller cleared (and may all-reduce over RCCL afterwards); a record that did not fit the buffer
 * it was exported into (scvod_chain_export_state with a fixed-size exchange buffer: header word 3 = 2) counts as a difference.
 * Asynchronous on `stream`; nothing is changed.  h_d_states is copied before the call returns. */
int scvod_batch_track_compare_device(scvod_ctx* ctx, const void* const* h_d_states, int32_t n_states, int32_t* d_differ, void* stream);
/* h_out8 = {mode the last scvod_batch_track ran, segments (workgroups), segments verified against their predecessor's
 * end state, segments walked again after that check failed, error bits, segment_steps, warmup_steps, scans of the batch whose
 * Frame::max_name the clustering could not determine (scvod_batch_cluster_last_name status 1 / 2: the chain handed out a fresh
 * cluster number there where the reference re-uses the last one)}.  Synchronises.
 * Returns SCVOD_ERR_CAPACITY when a chain state did not fit the walkers' workspace (the result is then invalid). */
int scvod_batch_track_stats(scvod_ctx* ctx, int32_t* h_out8);

/* Tracking result of scan `s` of the last scvod_batch_track.  Pointers are host memory owned by the ctx, valid until the
 * next fetch.  Clusters are the `car` clusters of the scan in ascending canonical name (smallest apri index). */
typedef struct scvod_track_result {
    int32_t n_apri;
    int32_t n_clusters;
    int32_t n_car_points;
    int32_t n_dynamic_clusters;
    int32_t n_dynamic_points;
    int32_t reserved;
    const int32_t* cluster_root;  /* [n_clusters] canonical cluster name                                  */
    const int32_t* cluster_size;  /* [n_clusters] |occupy_pts|                                            */
    const int32_t* cluster_state; /* [n_clusters] Cluster::state: -1 untouched, 0 static, 1 dynamic       */
    const int32_t* n_unique;      /* [n_clusters] unique labelled voxels of the successor under the cluster */
    const int32_t* pair_begin;    /* [n_clusters+1] offsets into pair_label / pair_count                   */
    const int32_t* pair_label;    /* remap_name keys (canonical names in the successor), ascending         */
    const int32_t* pair_count;    /* remap_name[label].size() after sampleVec                              */
    const uint8_t* pt_dyn;        /* [n_apri] SCVOD_DYN_*                                                  */
} scvod_track_result;
int scvod_batch_fetch_track(scvod_ctx* ctx, int32_t s, scvod_track_result* out);

/* Publishes the successor tables of the batch: Voxel::label, cluster sizes and types of every scan (the table its
 * predecessor is tracked against; the clustering kernel writes them together with the clusters, this call checks that the
 * clustering and the box rules of the batch are current).  A sharded sequence calls this, exports the tables of its blocks'
 * first scans, exchanges them, and then calls scvod_batch_track.  Asynchronous on `stream`. */
int scvod_batch_track_tables(scvod_ctx* ctx, void* stream);

/* Boundary message of a sequence shard: writes 1 + n_voxels records of 16 bytes into d_out (device memory, capacity
 * cap_records): record 0 = {records that follow, n_voxels, 0, 0}, then per voxel of scan `s` in ascending key
 * {key, label, |occupy_voxels| of the label's cluster, its type (0 erased, 1 other, 2 car)}.  Asynchronous on `stream`.
 * The shard that owns the PREVIOUS scan of the sequence passes the buffer (after a device-to-device / RCCL transfer) as
 * an external table to scvod_batch_track. */
int scvod_batch_export_table(scvod_ctx* ctx, int32_t s, void* d_out, int64_t cap_records, void* stream);

/* Curved-voxel clustering of the last batch (SSC::clusterAndCreateFrame, src/ssc.cpp:299-352):
 * connected components of apri points under the reference's 3x3x3 occupied-voxel neighbourhood (grid
 * clipped, no sector wrap-around, findVoxelNeighbors ssc.cpp:395-411).  The partition equals the
 * reference's; cluster NAMES are canonical (smallest apri index of the cluster) instead of the
 * order-dependent 5, 6, 7... of the reference.  Results stay on the device until fetched. */
int scvod_batch_cluster(scvod_ctx* ctx, void* stream, int32_t sync);
/* The visiting order of clusterAndCreateFrame (ssc.cpp:322-340) only matters around index triples OUTSIDE the grid (a
 * return at polar angle exactly 0 has sector index -1, ...).  Scans whose tables fit the LDS (every 64-beam scan) are
 * clustered with the exact visiting-order model always.  Larger scans (128 beams on a fine grid) first ask a local rule
 * per irregular run -- do the cells around its triple and around its key's own cell settle that every find sticks? (the
 * statement: csrc/scvod_k_cluster.inc cc_run_is_plain, pinned against the reference loop by
 * tests/test_irregular_runs_rule.py) -- and model the visiting order exactly for the components of the runs it does not
 * settle (about one 128-beam scan in thirty has such a run): such a scan is handed to a second kernel (k_cc_exact) whose
 * workgroups cluster it again from scratch and SHARE the passes over an affected component -- the listed voxels of every node, the
 * Jacobi rounds of the labelling times, the joins -- with the blocks of that kernel that lead no scan (round 6).
 * on = 1 (the default since round 6): whatever the component's size (4-6 ms per 1000 128-beam scans; nothing for scans that fit the
 * LDS).  on = 0 (the default of rounds 3-5): while those components have <= 4096 nodes together; beyond that the scan keeps
 * "everything found is joined" for them (the reference's partition then refines the device's) and is counted.
 * on = 2: exact without the rule (every component with an irregular run is clustered again: what the rule is
 * tested against).  on = 3: like 1, every scan's workgroup on its own (the round-5 form of on = 1: tens of milliseconds for a
 * component of tens of thousands of nodes; A/B runs and the test of the shared passes).
 * scvod_batch_cluster_stats: h_out4 = {scans of the last clustering that kept the approximation, nodes
 * of the components concerned (upper bound), 1 when the bound is lifted (on = 1, 2, 3), scans beyond the LDS whose z-planes were
 * too large for the windowed search and were joined on a forest in HBM instead (slower, same result)};
 * scvod_batch_cluster_rule_stats: h_out2 = {irregular runs of those larger scans the rule settled, runs whose component
 * was clustered again}.  Both synchronise. */
int scvod_set_cluster_exact(scvod_ctx* ctx, int32_t on);
/* scvod_batch_cluster_help_stats: h_out2 = {workgroups of the last clustering that published their passes, chunks (1024 nodes of
 * one pass) that helper blocks took}.  Synchronises. */
int scvod_batch_cluster_help_stats(scvod_ctx* ctx, int32_t* h_out2);
int scvod_batch_cluster_stats(scvod_ctx* ctx, int32_t* h_out4);
int scvod_batch_cluster_rule_stats(scvod_ctx* ctx, int32_t* h_out2);
/* Frame::max_name.  clusterAndCreateFrame ends with `frame_ssc.max_name = cluster_name ++;` (ssc.cpp:354): the frame keeps
 * the LAST USED running number K, and the first cluster SSC::tracking splits off or fuses in that frame is called K again
 * (ssc.cpp:1357, :1401) -- when a cluster K is still alive the insert (:1372, :1419) is a no-op and the new cluster is lost.
 * scvod_batch_cluster therefore also determines, per scan, which cluster (canonical name) still carries K when the visiting
 * loop ends (csrc/scvod_lastname.hip), and the tracking chain hands that name out first.  literal = 0 switches both off:
 * every new cluster gets a fresh number (rounds 1-3 of this library).  Default 1.
 * scvod_batch_cluster_last_name: h_out4[s] = {canonical name of the cluster carrying K or -1 (K was merged away, or every cluster
 * that could carry it was erased by the bounding-box refine: such a cluster is no cluster any more and the pass does not walk it),
 * lowest voxel slot whose first point belongs to it or -1, status, events replayed}; status 0 = exact; 1 = the classes
 * that had to be replayed hold more than 32 767 voxels together (node numbers are 16-bit in the replay's lists; up to 8192 the
 * tables live in LDS, beyond that in arena scratch: the facades of a 128-beam scan) or do not fit the scan's scratch: reported as "none"
 * (the fourth word then holds the number of voxels the set has, when the triage pass already knew: 33-54 k on the 128-beam bench job); 2 = more than 256 points with
 * an index triple outside the grid: reported as "none".  h_stats4 (optional) = {scans with status 1, with status 2, 0, 0}. */
int scvod_set_max_name_literal(scvod_ctx* ctx, int32_t literal);
int scvod_batch_cluster_last_name(scvod_ctx* ctx, int32_t* h_out4, int32_t cap_scans, int32_t* h_stats4);
/* copies the cluster name of every apri point of scan s into h_pt_cluster[cap]; returns the count (>= 0)
 * or a negative status */
int scvod_batch_fetch_clusters(scvod_ctx* ctx, int32_t s, int32_t* h_pt_cluster, int32_t cap);
/* Bounding-box refine + the bounding-box part of recognize for the clusters of scvod_batch_cluster
 * (SSC::refineClusterByBoundingBox ssc.cpp:437-467, SSC::recognize ssc.cpp:849-872):
 * per apri point of scan s, h_type[i] = -1 when its cluster is erased (min z > 0, fewer than toBeClass
 * points, z extent < 0.2 m), `car_label` when bbox area <= car_square && min z < min_z && max z < max_z,
 * otherwise `other_label` (the reference separates building / tree with PCL region growing, which stays
 * on the host).  No intensity merge (ssc.cpp:571-635) is applied.  Returns the count or a negative status. */
int scvod_batch_cluster_types(scvod_ctx* ctx, void* stream, int32_t sync);
int scvod_batch_fetch_cluster_types(scvod_ctx* ctx, int32_t s, int32_t car_label, int32_t other_label, int32_t* h_type,
                                    int32_t cap);
/* one-shot host version on an apri_vec the caller holds (voxelises it first) */
int scvod_cluster(scvod_ctx* ctx, const scvod_apri* h_apri, int32_t n, int32_t* h_pt_cluster);

/* Streaming ingest of a sequence held in HOST memory (the reference reads one .bin per scan, SSC::getCloud
 * src/ssc.cpp:1040-1125): chunks of `chunk_scans` scans travel host -> device on a copy stream into one of two device
 * buffers while the previous chunk runs scvod_batch_process on the ctx's stream; after the launches of a chunk are
 * enqueued `fn(user, ctx, first_scan, n_scans, stream)` is called to enqueue the consumers of that chunk (clustering,
 * tracking, map accumulation) on `stream` -- the arena holds one chunk at a time; fn may be NULL.  h_xyzi should be
 * pinned; SCVOD_INGEST_REGISTER pins it for the duration of the call.  Synchronous: returns when every chunk is done. */
#define SCVOD_INGEST_REGISTER 1
typedef int (*scvod_chunk_fn)(void* user, scvod_ctx* ctx, int32_t first_scan, int32_t n_scans, void* stream);
int scvod_sequence_ingest(scvod_ctx* ctx, const float* h_xyzi, const int32_t* h_scan_offsets, int32_t n_scans,
                          int32_t chunk_scans, int32_t flags, scvod_chunk_fn fn, void* user);

/* hipEvent timing of the kernels of the last batch call, in launch order:
 * names[i] (static strings), ms[i].  Returns the number of entries (<= cap). */
int scvod_batch_timings(scvod_ctx* ctx, const char** names, float* ms, int32_t cap);
/* enable (1) / disable (0) per-kernel hipEvent timing for subsequent batch calls */
int scvod_set_timing(scvod_ctx* ctx, int32_t enabled);

/* ---- correspondence search (north_star "GICP correspondence search"; the reference's
 * real analogue is the kd-tree look-up of src/evaluate.cpp:79-145) ----------------------- */

/* For every query point: index of the nearest map point (ties: lowest index) and the
 * squared distance (fp32, ((dx*dx + dy*dy) + dz*dz)); h_within[q] = 1 if any map point
 * lies within `radius` (squared distance < radius^2: pcl radiusSearch non-empty, evaluate.cpp:95,104).  h_xyz arrays
 * are n x 3 floats. */
int scvod_nn_search(scvod_ctx* ctx, const float* h_map_xyz, int32_t n_map,
                    const float* h_query_xyz, int32_t n_query, float radius,
                    int32_t* h_nn_idx, float* h_nn_sqdist, uint8_t* h_within);

/* pcl::KdTreeFLANN::radiusSearch as src/evaluate.cpp:95,104 uses it (is anything inside the radius?): per query the
 * nearest map point with squared distance < radius^2 (FLANN keeps dist < r^2), or index -1 / distance +inf.  Unlike
 * scvod_nn_search it never looks beyond the radius, so queries far from the map cost one 27-cell probe. */
int scvod_nn_radius_search(scvod_ctx* ctx, const float* h_map_xyz, int32_t n_map, const float* h_query_xyz,
                           int32_t n_query, float radius, int32_t* h_nn_idx, float* h_nn_sqdist);

/* The same search on arrays already resident in HBM (packed xyz, 12 B per point); asynchronous on `stream`
 * (NULL = the ctx's stream).  For sequence-scale evaluation without host round trips. */
int scvod_nn_search_device(scvod_ctx* ctx, const float* d_map_xyz, int32_t n_map,
                           const float* d_query_xyz, int32_t n_query, float radius,
                           int32_t* d_nn_idx, float* d_nn_sqdist, uint8_t* d_within, void* stream);

/* ---- world-frame static map of a sequence, mergeable across shards -------------------------------------------------
 * Reference analogue: `*instance_map += *rgb_ptr` over the clusters that are not dynamic (SSC::saveSegCloud mode 3,
 * src/ssc.cpp:477-554) plus the ground clouds and range/FOV rejects of the evaluation block (ssc.cpp:1460-1480), each scan
 * moved to the world by pcl::getTransformation(pose) (ssc.cpp:1455-1458).  Kept as a set of occupied cells of edge `leaf`
 * with ONE representative point per cell chosen by an order-independent rule (smallest packed in-cell offset), so that
 * shards can accumulate independently and merge their record lists (the payload of the RCCL all_gather) into
 * bit-identical maps. */
typedef struct scvod_map scvod_map;
#define SCVOD_MAP_NO_GROUND 1       /* leave cloud_out (ground) out                                  */
#define SCVOD_MAP_NO_REJECTED 2     /* leave cloud_eva_static (range/FOV rejects) out                 */
#define SCVOD_MAP_IGNORE_DYNAMIC 4  /* raw map: keep the points scvod_batch_track marked dynamic too  */
/* The map in two parts, so that most of it is accumulated WHILE the batch is tracked (a second stream): tracking can only
 * remove members of `car` clusters (src/ssc.cpp:1262), everything else is final once scvod_batch_cluster_types ran.
 * The cell rule is order-independent, so part UNTRACKED + part TRACKED == one call without a part flag, bit for bit. */
#define SCVOD_MAP_PART_UNTRACKED 8  /* every kept point that is not a member of a car cluster (needs no tracking result) */
#define SCVOD_MAP_PART_TRACKED 16   /* the car-cluster members scvod_batch_track left static                            */
int scvod_map_create(int device, int64_t capacity_cells, float leaf, scvod_map** out);
void scvod_map_destroy(scvod_map* map);
const char* scvod_map_last_error(const scvod_map* map);
int64_t scvod_map_capacity(const scvod_map* map);
int scvod_map_clear(scvod_map* map, void* stream);
/* pcl::getTransformation(x, y, z, roll, pitch, yaw) as a row-major 3x4 matrix */
void scvod_pose_matrix(const float pose[6], float T_out[12]);
/* adds the static points of every scan of ctx's last batch: h_poses [n_scans][6].  Asynchronous on `stream`. */
int scvod_batch_map_accumulate(scvod_ctx* ctx, scvod_map* map, const float* h_poses, int32_t flags, void* stream);
/* the same for scans [first, first + count) of the batch only (a shard's own block, without its halo); h_poses still [n_scans][6] */
int scvod_batch_map_accumulate_range(scvod_ctx* ctx, scvod_map* map, const float* h_poses, int32_t flags, int32_t first, int32_t count, void* stream);
/* occupied cells as 16-byte records {uint64 cell key, uint64 packed point} into device memory (NULL: count only);
 * *n_out = number of cells.  Synchronises `stream`.  Record order is unspecified (sort by key for a canonical order). */
int scvod_map_export(scvod_map* map, void* d_records, int64_t cap_records, int64_t* n_out, void* stream);
/* The same records grouped by OWNER for a reduce-scatter of the map over n_parts <= 64 shards (owner = a hash of the cell
 * key modulo n_parts, identical on every shard): h_counts[n_parts] receives the group sizes, group p starts at the sum of
 * the sizes before it.  Every shard sends group p to shard p (one all-to-all over xGMI: every link carries 1/n of the
 * map instead of everything converging on one root) and merges what it receives into the map of its own part.
 * Synchronises `stream`. */
int scvod_map_export_parts(scvod_map* map, int32_t n_parts, void* d_records, int64_t cap_records, int64_t* h_counts, void* stream);
/* The same grouping into FIXED-SIZE slots, without any host synchronisation (the form a timed multi-GPU step uses): group p
 * is written to d_records[p * cap_per_part .. (p + 1) * cap_per_part), the rest of a slot is padding (key ~0, skipped by
 * scvod_map_merge); d_counts (device, n_parts x int64, optional) receives the group sizes.  The all-to-all then moves
 * n_parts equal slots (all_to_all_single on device tensors).  A group that does not fit its slot is counted and reported
 * by the next scvod_map_export* call as SCVOD_ERR_CAPACITY.  Asynchronous on `stream`. */
int scvod_map_export_parts_padded(scvod_map* map, int32_t n_parts, void* d_records, int64_t cap_per_part, void* d_counts, void* stream);
/* inserts records exported by another shard (a record with key ~0 is padding and skipped).  Asynchronous. */
int scvod_map_merge(scvod_map* map, const void* d_records, int64_t n, void* stream);
/* the map as points: d_xyzi [cap][4] floats (cell origin + stored offset, intensity), optionally the records beside them */
int scvod_map_points(scvod_map* map, void* d_xyzi, void* d_records, int64_t cap, int64_t* n_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCVOD_H_ */
