// ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing in the product may include, link or call this.
//
// Sequential restatement of SSC::tracking (/root/reference/src/ssc.cpp:1250-1426) including the label
// bookkeeping that mutates the next frame, on std::unordered_map containers like the reference, so the
// iteration order of cluster_set is the library's own.  Used to pin the host-side bookkeeping of the
// product's C++ facade (dr-using-scv-od_amd/host/ssc.cpp) end to end: tests build two frames from the
// oracle's own binning / voxel table plus the toy segmentation below (a stand-in for SSC::segment /
// recognize, which are out of scope), run this, and compare cluster states and next-frame labels.
// PARITY UNPINNED against the real binary.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <vector>

#include "oracle.h"

namespace {

struct P3 {
    float x, y, z, intensity;
};
struct VoxelT {
    int range_idx, sector_idx, azimuth_idx;
    int label = -1;
    std::vector<int> ptIdx;
};
struct ClusterT {
    int track_id = -1, name = -1, type = -1, state = -1;
    int order = 0;  // walk position for the `ordered` mode: the name for a frame's own clusters, 2^30 + creation number for the ones a call made
    std::vector<int> occupy_pts, occupy_voxels;
    std::vector<P3> cloud;
};
struct FrameT {
    int max_name = 0;
    // LITERAL max_name (ssc.cpp:354 `frame_ssc.max_name = cluster_name ++;` stores the LAST USED running number K, not
    // K + 1): collide_name = the name, in the names this frame was built with, of the cluster that still carries running
    // number K when clusterAndCreateFrame ends, or -1 when K was merged away (mergeClusters renames oc -> nc).
    int collide_name = -1;
    bool literal = false, first_name_used = false;
    int created = 0;
    std::vector<P3> cloud_use;
    std::unordered_map<int, VoxelT> hash_cloud;
    std::unordered_map<int, ClusterT> cluster_set;
};

// same construction order as the facade: voxels inserted in ascending key order
void build_frame(const scvod_params& P, const scvod_apri* apri, int n, FrameT& f) {
    std::vector<int32_t> key(n ? n : 1), beg(n + 1), pts(n ? n : 1), idx3(3 * (n ? n : 1));
    std::vector<float> av(n ? n : 1), cov(n ? n : 1);
    int32_t nv = 0;
    oracle_voxelize(&P, apri, n, key.data(), beg.data(), pts.data(), av.data(), cov.data(), idx3.data(), nullptr, &nv);
    f.cloud_use.resize(n);
    for (int i = 0; i < n; ++i) f.cloud_use[i] = P3{apri[i].x, apri[i].y, apri[i].z, apri[i].intensity};
    f.hash_cloud.reserve(nv);
    for (int v = 0; v < nv; ++v) {
        VoxelT vx;
        vx.range_idx = idx3[3 * v];
        vx.sector_idx = idx3[3 * v + 1];
        vx.azimuth_idx = idx3[3 * v + 2];
        vx.ptIdx.assign(pts.begin() + beg[v], pts.begin() + beg[v + 1]);
        f.hash_cloud.insert(std::make_pair(key[v], vx));
    }
}

// toy stand-in for SSC::segment + recognize (identical rule in host/facade_check.cpp)
void toy_segment(FrameT& f, int car, int tree) {
    f.cluster_set.clear();
    std::vector<int> vkeys;
    for (auto& kv : f.hash_cloud) vkeys.push_back(kv.first);
    std::sort(vkeys.begin(), vkeys.end());
    for (int key : vkeys) {
        VoxelT& v = f.hash_cloud[key];
        int name = 5 + (v.range_idx / 6) * 64 + (v.sector_idx / 12);
        v.label = name;
        ClusterT& c = f.cluster_set[name];
        c.name = c.order = name;
        c.occupy_voxels.push_back(key);
        c.occupy_pts.insert(c.occupy_pts.end(), v.ptIdx.begin(), v.ptIdx.end());
    }
    f.max_name = 5 + 64 * 64;
    for (auto& kv : f.cluster_set) {
        ClusterT& c = kv.second;
        std::sort(c.occupy_voxels.begin(), c.occupy_voxels.end());
        std::sort(c.occupy_pts.begin(), c.occupy_pts.end());
        for (int p : c.occupy_pts) c.cloud.push_back(f.cloud_use[p]);
        c.type = (c.occupy_pts.size() < 400) ? car : tree;
    }
}

template <typename T>
void sampleVec(std::vector<T>& v) {  // utility.h:452-456
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
}
template <typename T>
void reduceVec(std::vector<T>& central, const std::vector<T>& reduce) {  // utility.h:445-450
    for (auto it = reduce.begin(); it != reduce.end(); it++) central.erase(std::remove(central.begin(), central.end(), *it), central.end());
}

float rad2deg_f(float r) { return (float)r * 180.0 / M_PI; }
float pointDistance2d(const P3& p) { return (float)sqrt(p.x * p.x + p.y * p.y); }
float getPolarAngle(const P3& p) {
    if (p.x == 0 && p.y == 0) return 0.f;
    if (p.y >= 0) return rad2deg_f((float)atan2f(p.y, p.x));
    double s = (float)atan2f(p.y, p.x) + 2 * M_PI;
    return (float)((float)s * 180.0 / M_PI);
}
float getAzimuth(const P3& p) { return rad2deg_f((float)atan2f(p.z, (float)pointDistance2d(p))); }

// `cluster_new.name = frame_next_.max_name ++` (ssc.cpp:1357, :1401).  Literal mode: the first name a frame hands out is K
// itself.  If cluster K is alive in cluster_set, `cluster_set.insert` at :1372 / :1419 is a no-op: the voxels were already
// re-labelled K (:1366, :1417), the source clusters were reduced (:1364, :1370) or erased (:1411), and cluster_new -- the
// split-off part or the fused car cluster -- is dropped: its points sit in no cluster any more and nothing walks it in the
// next call, while cluster K keeps its own occupy_voxels / type / cloud and now answers for more voxels.  If K is not in
// cluster_set (merged away during clustering, erased by refineClusterByBoundingBox, or erased by this very fuse) the insert
// succeeds and the only difference to a fresh number is the name.
int next_name(FrameT& f) {
    if (f.literal && !f.first_name_used) {
        f.first_name_used = true;
        if (f.collide_name != -1) return f.collide_name;
    }
    f.first_name_used = true;
    return f.max_name++;
}
long long g_literal[4];  // {splits whose insert was a no-op, fuses whose insert was a no-op, car clusters lost that way, their points}
void insert_new(FrameT& f, ClusterT& cluster_new, bool fuse) {
    cluster_new.order = (1 << 30) + f.created++;
    if (f.cluster_set.insert(std::make_pair(cluster_new.name, cluster_new)).second) return;
    g_literal[fuse ? 1 : 0]++;
    if (fuse) g_literal[3] += (long long)cluster_new.occupy_pts.size();
}

// ordered: walk frame_pre_.cluster_set in ascending cluster name instead of the container's order (the reference's order
// is the one of ITS names and ITS libstdc++, neither reproducible: DESIGN.md section 2); stats (optional): counters of
// the branches taken, see oracle_sequence_tracking_stats.
int tracking(const scvod_params& P, FrameT& frame_pre_, FrameT& frame_next_, const float pose_pre[6], const float pose_next[6], int car,
             int& name, bool ordered = false, long long* stats = nullptr) {
    int32_t R, S, A, bins;
    oracle_grid_dims(&P, &R, &S, &A, &bins);
    float T[12];
    oracle_pose_delta(pose_pre, pose_next, T);
    int dynamic_num = 0;
    std::vector<std::pair<const int, ClusterT>*> walk;
    for (auto& c : frame_pre_.cluster_set) walk.push_back(&c);
    if (ordered) std::sort(walk.begin(), walk.end(), [](auto* a, auto* b) { return a->second.order < b->second.order; });
    for (auto* cp : walk) {
        auto& c = *cp;
        if (c.second.type != car) continue;
        if (stats) {
            stats[0]++;
            stats[1] += (long long)c.second.occupy_pts.size();
            stats[2] += (long long)c.second.cloud.size();
            if (getenv("ORACLE_TRACK_AGES")) {  // experiment: the caller stored the frame index in `intensity`
                float now = -1e30f;
                for (auto& p : c.second.cloud) now = std::max(now, p.intensity);
                for (auto& p : c.second.cloud) {
                    int age = (int)(now - p.intensity);
                    if (age > stats[10]) stats[10] = age;
                    stats[11 + std::min(age, 52)]++;
                }
            }
        }
        if (c.second.track_id == -1) {
            c.second.track_id = name;
            name++;
        }
        std::vector<P3> cluster(c.second.cloud.size());
        for (size_t i = 0; i < cluster.size(); ++i) {  // transformCloud, utility.h:400-405
            const P3& in = c.second.cloud[i];
            cluster[i].x = T[0] * in.x + T[1] * in.y + T[2] * in.z + T[3];
            cluster[i].y = T[4] * in.x + T[5] * in.y + T[6] * in.z + T[7];
            cluster[i].z = T[8] * in.x + T[9] * in.y + T[10] * in.z + T[11];
            cluster[i].intensity = in.intensity;
        }
        std::unordered_map<int, std::vector<int>> remap_name;
        for (size_t k = 0; k < cluster.size(); k++) {
            P3 pt = cluster[k];
            float dis = pointDistance2d(pt);
            float angle = getPolarAngle(pt);
            float azimuth = getAzimuth(pt);
            int range_idx = std::ceil((dis - P.min_dis) / P.range_res) - 1;
            int sector_idx = std::ceil((angle - P.min_angle) / P.sector_res) - 1;
            int azimuth_idx = std::ceil((azimuth - P.min_azimuth) / P.azimuth_res) - 1;
            int voxel_idx = azimuth_idx * R * S + range_idx * S + sector_idx;
            auto it_find = frame_next_.hash_cloud.find(voxel_idx);
            if (it_find != frame_next_.hash_cloud.end() && it_find->second.label != -1) {
                auto l_find = remap_name.find(it_find->second.label);
                if (l_find == remap_name.end()) {
                    std::vector<int> vec;
                    vec.emplace_back(it_find->first);
                    remap_name.insert(std::make_pair(it_find->second.label, vec));
                } else {
                    l_find->second.emplace_back(it_find->first);
                }
            }
        }
        for (auto& re : remap_name) sampleVec(re.second);
        if (remap_name.size() == 0) {
            c.second.state = 1;
            dynamic_num++;
            if (stats) stats[3]++;
        } else if (remap_name.size() == 1) {
            auto it = remap_name.begin();
            float ratio = (float)it->second.size() / (float)frame_next_.cluster_set[it->first].occupy_voxels.size();
            if (ratio < P.occupancy) {
                if (frame_next_.cluster_set[it->first].type == car) {
                    c.second.state = 1;
                    dynamic_num++;
                    if (stats) stats[4]++;
                } else {
                    if (stats) stats[5]++;
                    c.second.state = 0;
                    c.second.type = frame_next_.cluster_set[it->first].type;
                    ClusterT cluster_new;
                    cluster_new.track_id = c.second.track_id;
                    cluster_new.name = next_name(frame_next_);
                    cluster_new.type = frame_next_.cluster_set[it->first].type;
                    cluster_new.occupy_voxels = it->second;
                    reduceVec(frame_next_.cluster_set[it->first].occupy_voxels, cluster_new.occupy_voxels);
                    for (auto& v : it->second) {
                        frame_next_.hash_cloud[v].label = cluster_new.name;
                        cluster_new.occupy_pts.insert(cluster_new.occupy_pts.end(), frame_next_.hash_cloud[v].ptIdx.begin(),
                                                      frame_next_.hash_cloud[v].ptIdx.end());
                    }
                    for (int p : cluster_new.occupy_pts) cluster_new.cloud.push_back(frame_next_.cloud_use[p]);
                    reduceVec(frame_next_.cluster_set[it->first].occupy_pts, cluster_new.occupy_pts);
                    insert_new(frame_next_, cluster_new, false);
                }
            } else {
                if (frame_next_.cluster_set[it->first].type == car) {
                    if (stats) stats[6]++;
                    c.second.state = 0;
                    frame_next_.cluster_set[it->first].track_id = c.second.track_id;
                    auto& dst = frame_next_.cluster_set[it->first].cloud;
                    dst.insert(dst.end(), cluster.begin(), cluster.end());
                } else if (stats) {
                    stats[7]++;
                }
            }
        } else {
            if (stats) stats[8]++;
            c.second.state = 0;
            ClusterT cluster_new;
            cluster_new.track_id = c.second.track_id;
            cluster_new.name = next_name(frame_next_);
            cluster_new.type = car;
            for (auto& re : remap_name) {
                if (frame_next_.cluster_set[re.first].type == car &&
                    ((float)re.second.size() / (float)frame_next_.cluster_set[re.first].occupy_voxels.size()) >= P.occupancy) {
                    auto& src = frame_next_.cluster_set[re.first];
                    cluster_new.occupy_pts.insert(cluster_new.occupy_pts.end(), src.occupy_pts.begin(), src.occupy_pts.end());
                    cluster_new.occupy_voxels.insert(cluster_new.occupy_voxels.end(), src.occupy_voxels.begin(), src.occupy_voxels.end());
                    if (stats) stats[9]++;
                    if (frame_next_.literal && re.first != cluster_new.name && frame_next_.cluster_set.count(cluster_new.name)) g_literal[2]++;
                    frame_next_.cluster_set.erase(re.first);
                }
            }
            for (int p : cluster_new.occupy_pts) cluster_new.cloud.push_back(frame_next_.cloud_use[p]);
            for (auto& v : cluster_new.occupy_voxels) frame_next_.hash_cloud[v].label = cluster_new.name;
            insert_new(frame_next_, cluster_new, true);
        }
    }
    return dynamic_num;
}


// frame with an explicit segmentation: pt_cluster = name per apri point, pt_type = Cluster::type or -1 for points whose
// cluster refineClusterByBoundingBox erased (ssc.cpp:437-467: erased clusters leave cluster_set, their voxels get label -1).
// Voxel::label = cluster of the voxel's points (clusterAndCreateFrame, ssc.cpp:388-392); occupy_voxels = sampleVec of the
// members' voxel_idx (ssc.cpp:382-384).  A voxel whose points sit in different clusters (possible only through index
// aliasing of the -1 bins) takes the label of its first point here; the reference's own choice depends on unordered_map
// iteration order.
void build_frame_seg(const scvod_params& P, const scvod_apri* apri, int n, const int32_t* pt_cluster, const int32_t* pt_type, FrameT& f) {
    build_frame(P, apri, n, f);
    f.cluster_set.clear();
    int max_name = 4;
    for (auto& kv : f.hash_cloud) {
        const int first = kv.second.ptIdx.front();
        kv.second.label = pt_type[first] == -1 ? -1 : pt_cluster[first];
    }
    for (int i = 0; i < n; ++i) {
        if (pt_cluster[i] > max_name) max_name = pt_cluster[i];  // (fresh numbers must stay clear of an erased cluster's name too)
        if (pt_type[i] == -1) continue;
        ClusterT& c = f.cluster_set[pt_cluster[i]];
        if (c.name == -1) {
            c.name = c.order = pt_cluster[i];
            c.type = pt_type[i];
            if (c.name > max_name) max_name = c.name;
        }
        c.occupy_pts.push_back(i);
        c.occupy_voxels.push_back(apri[i].voxel_idx);
        c.cloud.push_back(f.cloud_use[i]);
    }
    for (auto& kv : f.cluster_set) sampleVec(kv.second.occupy_voxels);
    f.max_name = max_name + 1;
}

void dyn_of_frame(const FrameT& f, int n, const int32_t* pt_type, uint8_t* dyn) {
    for (int i = 0; i < n; ++i) dyn[i] = pt_type[i] == -1 ? 2 : 0;  // 2: in no cluster
    for (auto& kv : f.cluster_set)
        if (kv.second.state == 1)  // saveSegCloud, ssc.cpp:479-481: dynamic_pt = occupy_pts of the clusters with state == 1
            for (int p : kv.second.occupy_pts) dyn[p] = 1;
}

// ssc.cpp:1274-1397 for ONE cluster against an untouched successor: remap_name (ordered map: only the iteration order of
// the output differs from the reference's unordered_map) and the state the reference would assign; nothing is mutated.
int decide_independent(const scvod_params& P, int R, int S, const ClusterT& c, FrameT& fb, const float T[12], int car,
                       std::map<int, std::vector<int>>& remap_name) {
    for (const P3& in : c.cloud) {
        P3 pt;
        pt.x = T[0] * in.x + T[1] * in.y + T[2] * in.z + T[3];  // transformCloud, utility.h:400-405
        pt.y = T[4] * in.x + T[5] * in.y + T[6] * in.z + T[7];
        pt.z = T[8] * in.x + T[9] * in.y + T[10] * in.z + T[11];
        float dis = pointDistance2d(pt);
        float angle = getPolarAngle(pt);
        float azimuth = getAzimuth(pt);
        int range_idx = std::ceil((dis - P.min_dis) / P.range_res) - 1;
        int sector_idx = std::ceil((angle - P.min_angle) / P.sector_res) - 1;
        int azimuth_idx = std::ceil((azimuth - P.min_azimuth) / P.azimuth_res) - 1;
        int voxel_idx = azimuth_idx * R * S + range_idx * S + sector_idx;
        auto it_find = fb.hash_cloud.find(voxel_idx);
        if (it_find != fb.hash_cloud.end() && it_find->second.label != -1) remap_name[it_find->second.label].emplace_back(it_find->first);
    }
    for (auto& re : remap_name) sampleVec(re.second);
    int state = -1;
    if (remap_name.size() == 0) {
        state = 1;
    } else if (remap_name.size() == 1) {
        auto it = remap_name.begin();
        float ratio = (float)it->second.size() / (float)fb.cluster_set[it->first].occupy_voxels.size();
        if (ratio < P.occupancy)
            state = (fb.cluster_set[it->first].type == car) ? 1 : 0;
        else if (fb.cluster_set[it->first].type == car)
            state = 0;
    } else {
        state = 0;
    }
    return state;
}

long long* g_track_stats = nullptr;
}  // namespace

extern "C" {

// Branch counters of the next oracle_sequence_tracking calls (chain 1 / 3), or NULL to stop counting.  stats[10]:
// {car clusters walked, their own points, points of their accumulated clouds, dynamic: no label hit, dynamic: one car
// label below `occupancy`, split off a non-car cluster, cloud appended to a car cluster, one non-car label at or above
// `occupancy` (state untouched), several labels (fuse), clusters fused}.
void oracle_track_stats(long long* stats) { g_track_stats = stats; }

// Builds frames a and b from their apri vectors (toy segmentation), runs tracking(a, b) and reports:
//   states: for every cluster of frame a with state != -1, sorted by name: {name, state, |occupy_voxels|}
//   next_labels: label of every voxel of frame b in ascending key order, after the call
int oracle_toy_tracking(const scvod_params* params, const scvod_apri* apri_a, int32_t n_a, const scvod_apri* apri_b, int32_t n_b,
                        const float pose_a[6], const float pose_b[6], int32_t car, int32_t tree, int32_t* states, int32_t* n_states,
                        int32_t* next_labels, int32_t* n_next_vox, int32_t* dynamic_num, int32_t* n_next_clusters) {
    FrameT fa, fb;
    build_frame(*params, apri_a, n_a, fa);
    build_frame(*params, apri_b, n_b, fb);
    toy_segment(fa, car, tree);
    toy_segment(fb, car, tree);
    int name = 0;
    int dyn = tracking(*params, fa, fb, pose_a, pose_b, car, name);
    std::vector<int> names;
    for (auto& kv : fa.cluster_set) names.push_back(kv.first);
    std::sort(names.begin(), names.end());
    int k = 0;
    for (int nm : names) {
        ClusterT& c = fa.cluster_set[nm];
        if (c.state != -1) {
            states[3 * k] = nm;
            states[3 * k + 1] = c.state;
            states[3 * k + 2] = (int)c.occupy_voxels.size();
            ++k;
        }
    }
    *n_states = k;
    std::vector<int> keys;
    for (auto& kv : fb.hash_cloud) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    for (size_t v = 0; v < keys.size(); ++v) next_labels[v] = fb.hash_cloud[keys[v]].label;
    *n_next_vox = (int)keys.size();
    *dynamic_num = dyn;
    *n_next_clusters = (int)fb.cluster_set.size();
    return 0;
}


// SSC::tracking of ONE pair with the successor in its freshly segmented state (first-order decision: what the device's
// scvod_batch_track computes).  Per `car` cluster of frame a in ascending name: out_clusters[k] = {name, state, unique
// labelled voxels hit, remap_name.size()}; pairs: for cluster k, pair_begin[k] .. pair_begin[k+1]: {label, |hits|} in
// ascending label.  No re-labelling is applied to b (the decisions of ssc.cpp:1323-1397 are read off, the bookkeeping of
// :1354-1372 / :1399-1419 is skipped).
int oracle_track_decide(const scvod_params* params, const scvod_apri* apri_a, int32_t n_a, const int32_t* cl_a, const int32_t* ty_a,
                        const scvod_apri* apri_b, int32_t n_b, const int32_t* cl_b, const int32_t* ty_b, const float T[12], int32_t car,
                        int32_t* out_clusters, int32_t* n_clusters, int32_t* pair_begin, int32_t* pairs) {
    const scvod_params& P = *params;
    FrameT fa, fb;
    build_frame_seg(P, apri_a, n_a, cl_a, ty_a, fa);
    build_frame_seg(P, apri_b, n_b, cl_b, ty_b, fb);
    int32_t R, S, A, bins;
    oracle_grid_dims(&P, &R, &S, &A, &bins);
    std::vector<int> names;
    for (auto& kv : fa.cluster_set)
        if (kv.second.type == car) names.push_back(kv.first);
    std::sort(names.begin(), names.end());
    int k = 0, np = 0;
    for (int nm : names) {
        ClusterT& c = fa.cluster_set[nm];
        std::map<int, std::vector<int>> remap_name;
        const int state = decide_independent(P, R, S, c, fb, T, car, remap_name);
        int uniq = 0;
        for (auto& re : remap_name) uniq += (int)re.second.size();
        out_clusters[4 * k] = nm;
        out_clusters[4 * k + 1] = state;
        out_clusters[4 * k + 2] = uniq;
        out_clusters[4 * k + 3] = (int)remap_name.size();
        pair_begin[k] = np;
        for (auto& re : remap_name) {
            pairs[2 * np] = re.first;
            pairs[2 * np + 1] = (int)re.second.size();
            ++np;
        }
        ++k;
    }
    pair_begin[k] = np;
    *n_clusters = k;
    return 0;
}

// SSC::segDF's tracking loop (ssc.cpp:1449-1451) over a sequence of segmented frames.  chain == 1: the reference's
// semantics -- tracking(i, i + 1) re-labels / splits / fuses the clusters of frame i + 1 before tracking(i + 1, i + 2) walks
// them, cluster_set iterated in the container's order.  chain == 0: every pair against a freshly segmented successor
// (the clusters of one pair still see each other's re-labelling).  chain == 2: every cluster on its own against the
// untouched successor (first-order; the device path).  pt_dyn per apri point: 1 = member of a cluster with state == 1 after the loop
// (saveSegCloud's dynamic_pt, ssc.cpp:479-481), 2 = in no cluster, 0 otherwise.
static int sequence_tracking(const scvod_params* params, const scvod_apri* apri, const int32_t* offs, int32_t n_scans,
                             const int32_t* pt_cluster, const int32_t* pt_type, const int32_t* collide, const float* poses, int32_t car,
                             int32_t chain, uint8_t* pt_dyn, int32_t* dynamic_clusters) {
    const scvod_params& P = *params;
    std::vector<FrameT> frames(n_scans);
    for (int s = 0; s < n_scans; ++s) {
        build_frame_seg(P, apri + offs[s], offs[s + 1] - offs[s], pt_cluster + offs[s], pt_type + offs[s], frames[s]);
        if (collide) {
            frames[s].literal = true;
            frames[s].collide_name = collide[s];
        }
    }
    int name = 0, dyn = 0;
    for (int i = 0; i + 1 < n_scans; ++i) {
        if (chain == 2) {  // every cluster on its own against the untouched successor (what the device computes)
            int32_t R, S, A, bins;
            oracle_grid_dims(&P, &R, &S, &A, &bins);
            float T[12];
            oracle_pose_delta(poses + 6 * i, poses + 6 * (i + 1), T);
            for (auto& kv : frames[i].cluster_set) {
                if (kv.second.type != car) continue;
                std::map<int, std::vector<int>> remap_name;
                kv.second.state = decide_independent(P, R, S, kv.second, frames[i + 1], T, car, remap_name);
                dyn += kv.second.state == 1;
            }
        } else if (chain) {  // 1: container order; 3: ascending cluster name (the order the device path defines)
            dyn += tracking(P, frames[i], frames[i + 1], poses + 6 * i, poses + 6 * (i + 1), car, name, chain == 3, g_track_stats);
        } else {
            FrameT fresh;
            build_frame_seg(P, apri + offs[i + 1], offs[i + 2] - offs[i + 1], pt_cluster + offs[i + 1], pt_type + offs[i + 1], fresh);
            dyn += tracking(P, frames[i], fresh, poses + 6 * i, poses + 6 * (i + 1), car, name);
        }
    }
    for (int s = 0; s < n_scans; ++s) dyn_of_frame(frames[s], offs[s + 1] - offs[s], pt_type + offs[s], pt_dyn + offs[s]);
    if (dynamic_clusters) *dynamic_clusters = dyn;
    return 0;
}

int oracle_sequence_tracking(const scvod_params* params, const scvod_apri* apri, const int32_t* offs, int32_t n_scans,
                             const int32_t* pt_cluster, const int32_t* pt_type, const float* poses, int32_t car, int32_t chain,
                             uint8_t* pt_dyn, int32_t* dynamic_clusters) {
    return sequence_tracking(params, apri, offs, n_scans, pt_cluster, pt_type, nullptr, poses, car, chain, pt_dyn, dynamic_clusters);
}

// The same loop with the reference's LITERAL `max_name` (ssc.cpp:354: the last USED running number K, so the first
// `max_name ++` of every frame hands out K again, ssc.cpp:1357 / :1401).  collide[s] = name (in pt_cluster's names) of the
// cluster of scan s that carries running number K when clusterAndCreateFrame ends, -1 if none does
// (oracle_cluster_last_name).  literal_stats[4] (optional): {split-offs dropped by the no-op insert, fuses dropped, car
// clusters that left cluster_set that way, their points}.
int oracle_sequence_tracking_literal(const scvod_params* params, const scvod_apri* apri, const int32_t* offs, int32_t n_scans,
                                     const int32_t* pt_cluster, const int32_t* pt_type, const int32_t* collide, const float* poses,
                                     int32_t car, int32_t chain, uint8_t* pt_dyn, int32_t* dynamic_clusters, int64_t* literal_stats) {
    for (auto& v : g_literal) v = 0;
    const int rc = sequence_tracking(params, apri, offs, n_scans, pt_cluster, pt_type, collide, poses, car, chain, pt_dyn, dynamic_clusters);
    if (literal_stats)
        for (int k = 0; k < 4; ++k) literal_stats[k] = g_literal[k];
    return rc;
}


// Timed CPU baseline of the WHOLE path bench.py times on the GPU, single-threaded, scan after scan the way SSC::segDF runs
// (ssc.cpp:1434-1451): Patchwork -> makeApriVec -> makeHashCloud -> clusterAndCreateFrame -> bounding-box refine + type
// rules -> SSC::tracking of every consecutive pair (the reference's sequential chain).  stage_s[6] = seconds in
// {patchwork, bin, voxelize, cluster, types, tracking}.  Optionally returns, per INPUT point, a byte: 0 static (ground,
// range/FOV reject or member of a cluster that is not dynamic), 1 dynamic, 2 in no cluster, 3 dropped by Patchwork.
int oracle_time_sequence(const scvod_params* params, const float* xyzi, const int32_t* offsets, int32_t n_scans, const float* poses,
                         int32_t car, int32_t other, double stage_s[6], uint8_t* in_label, int64_t* checksum) {
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    for (int k = 0; k < 6; ++k) stage_s[k] = 0.0;
    const scvod_params& P = *params;
    std::vector<FrameT> frames(n_scans);
    std::vector<std::vector<int32_t>> src(n_scans), types(n_scans);
    int64_t cs = 0;
    for (int s = 0; s < n_scans; ++s) {
        const float* pts = xyzi + 4 * (size_t)offsets[s];
        const int n = offsets[s + 1] - offsets[s];
        std::vector<uint8_t> cls(n ? n : 1);
        std::vector<int32_t> g(n ? n : 1), ng(n ? n : 1);
        std::vector<scvod_patch_plane> planes(SCVOD_MAX_PATCHES);
        int32_t n_g = 0, n_ng = 0, n_p = 0;
        auto t0 = clk::now();
        oracle_patchwork(params, nullptr, pts, n, 0, cls.data(), g.data(), &n_g, ng.data(), &n_ng, planes.data(), &n_p);
        auto t1 = clk::now();
        std::vector<float> ngc(4 * (size_t)(n_ng ? n_ng : 1));
        for (int k = 0; k < n_ng; ++k) std::memcpy(&ngc[4 * (size_t)k], pts + 4 * (size_t)ng[k], 16);
        std::vector<scvod_apri> apri(n_ng ? n_ng : 1);
        std::vector<int32_t> asrc(n_ng ? n_ng : 1), rej(n_ng ? n_ng : 1);
        int32_t n_a = 0, n_r = 0;
        oracle_bin(params, ngc.data(), n_ng, 1, apri.data(), asrc.data(), &n_a, rej.data(), &n_r);
        auto t2 = clk::now();
        FrameT& f = frames[s];
        build_frame(P, apri.data(), n_a, f);  // makeHashCloud
        auto t3 = clk::now();
        std::vector<int32_t> cl(n_a ? n_a : 1), ty(n_a ? n_a : 1);
        int32_t mx = 0;
        oracle_cluster(params, apri.data(), n_a, cl.data(), &mx);
        {   // canonical cluster names (smallest apri index of the cluster) instead of the running numbers 5, 6, 7 ...: the
            // names only order the walk of SSC::tracking, and this is the order the product defines
            std::unordered_map<int, int> first;
            for (int i = 0; i < n_a; ++i) first.emplace(cl[i], i);
            // Frame::max_name as ssc.cpp:354 stores it: the LAST USED running number (mx); the cluster that still carries it
            f.literal = true;
            f.collide_name = first.count(mx) ? first[mx] : -1;
            for (int i = 0; i < n_a; ++i) cl[i] = first[cl[i]];
        }
        auto t4 = clk::now();
        oracle_cluster_types(params, apri.data(), n_a, cl.data(), car, other, ty.data());
        // Frame as recognize leaves it: labels, occupy lists, types (the voxel table of build_frame is reused)
        f.cluster_set.clear();
        int max_name = 4;
        for (auto& kv : f.hash_cloud) {
            const int first = kv.second.ptIdx.front();
            kv.second.label = ty[first] == -1 ? -1 : cl[first];
        }
        for (int i = 0; i < n_a; ++i) {
            if (ty[i] == -1) continue;
            ClusterT& c = f.cluster_set[cl[i]];
            if (c.name == -1) {
                c.name = c.order = cl[i];
                c.type = ty[i];
                if (c.name > max_name) max_name = c.name;
            }
            c.occupy_pts.push_back(i);
            c.occupy_voxels.push_back(apri[i].voxel_idx);
            c.cloud.push_back(f.cloud_use[i]);
        }
        for (auto& kv : f.cluster_set) sampleVec(kv.second.occupy_voxels);
        f.max_name = n_a + 5;  // fresh numbers: clear of every canonical name (erased clusters included)
        auto t5 = clk::now();
        stage_s[0] += secs(t0, t1);
        stage_s[1] += secs(t1, t2);
        stage_s[2] += secs(t2, t3);
        stage_s[3] += secs(t3, t4);
        stage_s[4] += secs(t4, t5);
        cs += n_g + 3 * (int64_t)n_ng + 7 * (int64_t)n_a + 11 * (int64_t)f.hash_cloud.size() + 13 * (int64_t)f.cluster_set.size();
        if (in_label) {
            uint8_t* L = in_label + offsets[s];
            for (int i = 0; i < n; ++i) L[i] = cls[i] == SCVOD_CLS_DROPPED ? 3 : 0;
            src[s].resize(n_a);
            for (int i = 0; i < n_a; ++i) src[s][i] = ng[asrc[i]];
            types[s] = ty;
        }
    }
    auto t6 = clk::now();
    int name = 0, dyn = 0;
    for (int i = 0; i + 1 < n_scans; ++i) dyn += tracking(P, frames[i], frames[i + 1], poses + 6 * i, poses + 6 * (i + 1), car, name, true);
    stage_s[5] = secs(t6, clk::now());
    cs += 17 * (int64_t)dyn;
    if (in_label) {
        for (int s = 0; s < n_scans; ++s) {
            const int n_a = (int)src[s].size();
            std::vector<uint8_t> d(n_a ? n_a : 1);
            dyn_of_frame(frames[s], n_a, types[s].data(), d.data());
            for (int i = 0; i < n_a; ++i) in_label[offsets[s] + src[s][i]] = d[i];
        }
    }
    if (checksum) *checksum = cs;
    return 0;
}

}  // extern "C"
