/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 * C interface of the CPU restatement of the SCV-OD hot path, loaded by tests/ through
 * ctypes (oracle/liboracle.so).  It reuses the POD types of the public C-ABI header so
 * the same ctypes structures describe both sides; it shares no code with the product. */
#ifndef SCVOD_ORACLE_H_
#define SCVOD_ORACLE_H_

#include "../include/scvod.h"

#ifdef __cplusplus
extern "C" {
#endif

void oracle_params_default(scvod_params* p);
void oracle_pw_params_default(scvod_pw_params* p);
void oracle_grid_dims(const scvod_params* p, int32_t* range_num, int32_t* sector_num, int32_t* azimuth_num,
                      int32_t* bin_num);

/* PatchWork::estimate_ground, include/patchwork.h:277-398 */
int oracle_patchwork(const scvod_params* params, const scvod_pw_params* pw, const float* xyzi, int32_t n,
                     int32_t sort_mode, uint8_t* cls, int32_t* ground_idx, int32_t* n_ground, int32_t* nonground_idx,
                     int32_t* n_nonground, scvod_patch_plane* planes, int32_t* n_patches);
/* number of estimate_plane_ calls that met an empty ground set since the last reset (must stay 0, see patchwork_oracle.cpp) */
long long oracle_patchwork_empty_sets(int reset);
void oracle_patch_ids(const scvod_params* params, const float* xyzi, int32_t n, int32_t* pid);
void oracle_svd3(const float cov_rowmajor[9], float sv[3], float U_rowmajor[9]);

/* SSC::makeApriVec, src/ssc.cpp:155-195.  apply_filter == 0: SSC::tracking's unfiltered
 * re-binning (ssc.cpp:1280-1286).  src_idx[k] = input index of apri[k]; rejected_idx =
 * points pushed to cloud_eva_static. */
int oracle_bin(const scvod_params* params, const float* xyzi, int32_t n, int32_t apply_filter, scvod_apri* apri,
               int32_t* src_idx, int32_t* n_kept, int32_t* rejected_idx, int32_t* n_rejected);

/* SSC::makeHashCloud, src/ssc.cpp:253-289.  Output sorted by ascending key. */
int oracle_voxelize(const scvod_params* params, const scvod_apri* apri, int32_t n, int32_t* vox_key,
                    int32_t* vox_pt_begin, int32_t* vox_pts, float* vox_av, float* vox_cov, int32_t* vox_idx3,
                    float* vox_center, int32_t* n_vox);

/* getTransformation(next)^-1 * getTransformation(pre), src/ssc.cpp:1255-1257 */
void oracle_pose_delta(const float pose_pre[6], const float pose_next[6], float T_out[12]);

/* bulk part of SSC::tracking, src/ssc.cpp:1274-1321 (+ utility.h:394-406) */
int oracle_track_probe(const scvod_params* params, const float* xyzi, const int32_t* offsets, int32_t n_clusters,
                       const float T[12], const int32_t* next_keys, const int32_t* next_labels, int32_t n_next_vox,
                       int32_t* hit_slot, int32_t* uniq_slots, int32_t* uniq_begin);

/* SSC::clusterAndCreateFrame (src/ssc.cpp:299-393): per apri point cluster name, and per
 * voxel (sorted-key order) label.  Returns number of clusters. */
int oracle_cluster(const scvod_params* params, const scvod_apri* apri, int32_t n, int32_t* pt_cluster,
                   int32_t* max_name);

/* the cluster that still carries the last running number K = Frame::max_name as ssc.cpp:354 stores it (smallest point
 * index, -1 if none); info[6]: see ssc_oracle.cpp */
int oracle_cluster_last_name(const scvod_params* params, const scvod_apri* apri, int32_t n, int64_t* info);

/* refineClusterByBoundingBox + bounding-box part of recognize (ssc.cpp:437-467, 723-751, 849-872) */
int oracle_cluster_types(const scvod_params* params, const scvod_apri* apri, int32_t n, const int32_t* pt_cluster,
                         int32_t car_label, int32_t other_label, int32_t* pt_type);

/* SSC::tracking incl. the host bookkeeping (src/ssc.cpp:1250-1426) on two frames built from apri vectors with a toy
 * segmentation (oracle/tracking_oracle.cpp); states = {name, state, |occupy_voxels|} per tracked cluster of frame a. */
int oracle_toy_tracking(const scvod_params* params, const scvod_apri* apri_a, int32_t n_a, const scvod_apri* apri_b, int32_t n_b,
                        const float pose_a[6], const float pose_b[6], int32_t car, int32_t tree, int32_t* states, int32_t* n_states,
                        int32_t* next_labels, int32_t* n_next_vox, int32_t* dynamic_num, int32_t* n_next_clusters);

/* one pair, successor freshly segmented (first-order decision of ssc.cpp:1323-1397); see tracking_oracle.cpp */
int oracle_track_decide(const scvod_params* params, const scvod_apri* apri_a, int32_t n_a, const int32_t* cl_a, const int32_t* ty_a,
                        const scvod_apri* apri_b, int32_t n_b, const int32_t* cl_b, const int32_t* ty_b, const float T[12], int32_t car,
                        int32_t* out_clusters, int32_t* n_clusters, int32_t* pair_begin, int32_t* pairs);
/* SSC::segDF's tracking loop over a segmented sequence: chain = 1 the reference's sequential re-labelling, 0 first-order */
int oracle_sequence_tracking(const scvod_params* params, const scvod_apri* apri, const int32_t* offs, int32_t n_scans,
                             const int32_t* pt_cluster, const int32_t* pt_type, const float* poses, int32_t car, int32_t chain,
                             uint8_t* pt_dyn, int32_t* dynamic_clusters);

/* the same loop with the reference's literal max_name (first new cluster of a frame re-uses running number K) */
int oracle_sequence_tracking_literal(const scvod_params* params, const scvod_apri* apri, const int32_t* offs, int32_t n_scans,
                                     const int32_t* pt_cluster, const int32_t* pt_type, const int32_t* collide, const float* poses,
                                     int32_t car, int32_t chain, uint8_t* pt_dyn, int32_t* dynamic_clusters, int64_t* literal_stats);

/* brute-force nearest neighbour / radius test (src/evaluate.cpp:79-145 analogue) */
int oracle_nn_search(const float* map_xyz, int32_t n_map, const float* query_xyz, int32_t n_query, float radius,
                     int32_t* nn_idx, float* nn_sqdist, uint8_t* within);

/* label filter + intensity scaling of SSC::getCloud (src/ssc.cpp:1063-1076) and pcl::VoxelGrid 0.08 m
 * (src/ssc.cpp:1103-1106, PCL 1.8.1 restated).  labels == NULL: VoxelGrid only.  Returns 1 when PCL's "leaf size too
 * small" branch copied the input. */
int oracle_voxelgrid(const float* xyzi, const uint32_t* labels, int32_t n, float max_intensity, const float leaf[3],
                     int32_t sort_mode, float* out_xyzi, int32_t* n_out);

/* one poses.txt line -> velodyne-frame pose, SSC::getPose KITTI branch (src/ssc.cpp:960-989, utility.h:488-505) */
int oracle_kitti_pose(const float tr[16], const float cam12[12], float pose6[6], float velo_to_cam[16]);

/* index flips between the two readings of the unqualified atan2 in utility.h:382-391 (float overload vs double + rounding) */
int oracle_atan2_overload_flips(const scvod_params* params, const float* xyzi, int32_t n, int64_t* counts);

/* libm probes for tests/test_math_spec.py */
float oracle_libm_atan2f(float y, float x);
double oracle_libm_atan2(double y, double x);

/* Timed CPU baseline: runs Patchwork + binning + voxelisation over n_scans scans
 * (concatenated xyzi, offsets[n_scans+1]) single-threaded; returns seconds per stage in
 * stage_s[3] = {patchwork, bin, voxelize}. */
int oracle_time_process(const scvod_params* params, const float* xyzi, const int32_t* offsets, int32_t n_scans,
                        double stage_s[3], int64_t* checksum);

/* Timed CPU baseline of the whole path (Patchwork, binning, voxel descriptors, clustering, box rules, the sequential
 * tracking chain), single thread; stage_s[6]; in_label (optional) per input point: 0 static, 1 dynamic, 2 in no cluster,
 * 3 dropped by Patchwork. */
int oracle_time_sequence(const scvod_params* params, const float* xyzi, const int32_t* offsets, int32_t n_scans, const float* poses,
                         int32_t car, int32_t other, double stage_s[6], uint8_t* in_label, int64_t* checksum);

#ifdef __cplusplus
}
#endif
#endif
