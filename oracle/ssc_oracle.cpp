// ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing in the product may include, link or call this.
//
// CPU restatement of the SSC side of the hot path of /root/reference:
//   SSC::SSC grid sizes          src/ssc.cpp:36-39
//   SSC::makeApriVec             src/ssc.cpp:155-195   (+ include/utility.h:346-392)
//   SSC::makeHashCloud           src/ssc.cpp:253-289   (+ include/utility.h:96-119)
//   SSC::tracking (bulk part)    src/ssc.cpp:1255-1257, 1274-1321 (+ utility.h:394-406)
//   SSC::clusterAndCreateFrame   src/ssc.cpp:299-419   (host glue needed to obtain labels)
//   kd-tree NN / radius look-up  src/evaluate.cpp:79-145 (brute force here)
// It calls glibc libm exactly where the reference does (atan2f via the float overloads that
// <math.h> injects into the global namespace -- pcl/pcl_macros.h includes <math.h>; sqrt).
// PARITY UNPINNED: no reference tests or golden vectors exist for this path and the
// reference cannot be built in this image.
#include "oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace {

struct Grid {
    int range_num, sector_num, azimuth_num, bin_num;
};

Grid grid_of(const scvod_params& p) {  // ssc.cpp:36-39
    Grid g;
    g.range_num = (int)std::ceil((p.max_dis - p.min_dis) / p.range_res);
    g.sector_num = (int)std::ceil((p.max_angle - p.min_angle) / p.sector_res);
    g.azimuth_num = (int)std::ceil((p.max_azimuth - p.min_azimuth) / p.azimuth_res);
    g.bin_num = g.range_num * g.sector_num * g.azimuth_num;
    return g;
}

// include/utility.h:346-354
template <typename T>
float rad2deg(const T& radians) {
    return (float)radians * 180.0 / M_PI;
}
template <typename T>
float deg2rad(const T& degrees) {
    return (float)degrees * M_PI / 180.0;
}
struct P3 {
    float x, y, z, intensity;
};
float pointDistance2d(const P3& p1) { return (float)sqrt((p1.x) * (p1.x) + (p1.y) * (p1.y)); }
float getPolarAngle(const P3& p) {
    if (p.x == 0 && p.y == 0) {
        return 0.f;
    } else if (p.y >= 0) {
        return (float)rad2deg((float)atan2f(p.y, p.x));
    } else {
        return (float)rad2deg((float)atan2f(p.y, p.x) + 2 * M_PI);
    }
}
float getAzimuth(const P3& p) { return (float)rad2deg((float)atan2f(p.z, (float)pointDistance2d(p))); }

void bin_point(const scvod_params& P, const Grid& g, const P3& pt, float dis, float angle, float azimuth,
               scvod_apri& apri) {
    apri.x = pt.x;
    apri.y = pt.y;
    apri.z = pt.z;
    apri.range = dis;
    apri.angle = angle;
    apri.azimuth = azimuth;
    apri.intensity = pt.intensity;
    apri.range_idx = std::ceil((dis - P.min_dis) / P.range_res) - 1;
    apri.sector_idx = std::ceil((angle - P.min_angle) / P.sector_res) - 1;
    apri.azimuth_idx = std::ceil((azimuth - P.min_azimuth) / P.azimuth_res) - 1;
    apri.voxel_idx = apri.azimuth_idx * g.range_num * g.sector_num + apri.range_idx * g.sector_num + apri.sector_idx;
}

struct VoxelO {
    int range_idx, sector_idx, azimuth_idx;
    int label = -1;
    float center[4];
    std::vector<int> ptIdx;
    std::vector<float> intensity_record;
    float intensity_av = 0.f;
    float intensity_cov = 0.f;
};

void make_hash_cloud(const scvod_params& P, const scvod_apri* apriIn, int n,
                     std::unordered_map<int, VoxelO>& hash_cloud) {
    for (int i = 0; i < n; i++) {
        const scvod_apri& apri = apriIn[i];
        auto it_find = hash_cloud.find(apri.voxel_idx);
        if (it_find != hash_cloud.end()) {
            it_find->second.ptIdx.emplace_back(i);
            it_find->second.intensity_record.emplace_back(apri.intensity);
            it_find->second.intensity_av += apri.intensity;
        } else {
            VoxelO voxel;
            voxel.ptIdx.emplace_back(i);
            voxel.intensity_record.emplace_back(apri.intensity);
            voxel.intensity_av += apri.intensity;
            voxel.range_idx = apri.range_idx;
            voxel.sector_idx = apri.sector_idx;
            voxel.azimuth_idx = apri.azimuth_idx;
            float range_center = (apri.range_idx * 2 + 1) / 2 * P.range_res + P.min_dis;
            float sector_center = deg2rad((apri.sector_idx * 2 + 1) / 2 * P.sector_res) + P.min_angle;
            float azimuth_center = deg2rad((apri.azimuth_idx * 2 + 1) / 2 * P.azimuth_res) + deg2rad(P.min_azimuth);
            voxel.center[0] = range_center * std::cos(sector_center);
            voxel.center[1] = range_center * std::sin(sector_center);
            voxel.center[2] = range_center * std::tan(azimuth_center);
            voxel.center[3] = apri.voxel_idx;
            hash_cloud.insert(std::make_pair(apri.voxel_idx, voxel));
        }
    }
    for (auto& vox : hash_cloud) {
        vox.second.intensity_av /= vox.second.ptIdx.size();
        for (auto& in : vox.second.intensity_record) {
            vox.second.intensity_cov += std::pow((in - vox.second.intensity_av), 2);
        }
        vox.second.intensity_cov /= vox.second.ptIdx.size();
    }
}

// pcl::getTransformation (PCL 1.8 common/impl/eigen.hpp), Scalar = float; row-major 3x4
void get_transformation(float x, float y, float z, float roll, float pitch, float yaw, float t[12]) {
    float A = std::cos(yaw), B = std::sin(yaw), C = std::cos(pitch), D = std::sin(pitch), E = std::cos(roll),
          F = std::sin(roll), DE = D * E,
          DF = D * F;
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
inline float sum3(float c0, float c1, float c2) { return c0 + (c1 + c2); }  // Eigen redux unroller, size 3

// Eigen::Affine3f::inverse() (Affine mode: general 3x3 inverse by cofactors)
void affine_inverse(const float m[12], float r[12]) {
    auto M = [&](int i, int j) { return m[4 * i + j]; };
    auto cof = [&](int i, int j) {
        int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return M(i1, j1) * M(i2, j2) - M(i1, j2) * M(i2, j1);
    };
    float c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
    float det = sum3(c0 * M(0, 0), c1 * M(1, 0), c2 * M(2, 0));
    float invdet = 1.f / det;
    float R[3][3];
    R[0][0] = c0 * invdet;
    R[0][1] = c1 * invdet;
    R[0][2] = c2 * invdet;
    R[1][0] = cof(0, 1) * invdet;
    R[1][1] = cof(1, 1) * invdet;
    R[2][2] = cof(2, 2) * invdet;
    R[1][2] = cof(2, 1) * invdet;
    R[2][1] = cof(1, 2) * invdet;
    R[2][0] = cof(0, 2) * invdet;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) r[4 * i + j] = R[i][j];
        r[4 * i + 3] = sum3((-R[i][0]) * m[3], (-R[i][1]) * m[7], (-R[i][2]) * m[11]);
    }
}
// Affine3f * Affine3f
void affine_mul(const float a[12], const float b[12], float r[12]) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) r[4 * i + j] = sum3(a[4 * i] * b[j], a[4 * i + 1] * b[4 + j], a[4 * i + 2] * b[8 + j]);
        r[4 * i + 3] = sum3(a[4 * i] * b[3], a[4 * i + 1] * b[7], a[4 * i + 2] * b[11]) + a[4 * i + 3];
    }
}

std::vector<int> find_voxel_neighbors(const Grid& g, int range_idx_, int sector_idx_, int azimuth_idx_, int size_) {
    std::vector<int> neighborIdxs;  // ssc.cpp:395-411
    if (range_idx_ > g.range_num * 0.6) size_ = 1;
    for (int x = range_idx_ - size_; x <= range_idx_ + size_; x++) {
        if (x > g.range_num - 1 || x < 0) continue;
        for (int y = sector_idx_ - size_; y <= sector_idx_ + size_; y++) {
            if (y > g.sector_num - 1 || y < 0) continue;
            for (int z = azimuth_idx_ - size_; z <= azimuth_idx_ + size_; z++) {
                if (z > g.azimuth_num - 1 || z < 0) continue;
                neighborIdxs.emplace_back(x * g.sector_num + y + z * g.range_num * g.sector_num);
            }
        }
    }
    return neighborIdxs;
}

}  // namespace

extern "C" {

void oracle_params_default(scvod_params* p) {  // utility.h:283-310
    std::memset(p, 0, sizeof(*p));
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

void oracle_pw_params_default(scvod_pw_params* p) {  // patchwork.h:48-51, 115-129
    std::memset(p, 0, sizeof(*p));
    p->num_iter = 3;
    p->num_lpr = 20;
    p->num_min_pts = 10;
    p->num_rings_of_interest = 4;
    const int s[4] = {16, 32, 54, 32}, r[4] = {2, 4, 4, 4};
    const double e[4] = {-1.2, -0.9984, -0.851, -0.605}, f[4] = {0.0, 0.000125, 0.000185, 0.000185};
    for (int i = 0; i < 4; ++i) {
        p->num_sectors_each_zone[i] = s[i];
        p->num_rings_each_zone[i] = r[i];
        p->elevation_thr[i] = e[i];
        p->flatness_thr[i] = f[i];
    }
    p->th_seeds = 0.3;
    p->th_dist = 0.1;
    p->max_range = 80.0;
    p->min_range = 2.7;
    p->uprightness_thr = 0.707;
    p->adaptive_seed_selection_margin = -1.1;
}

void oracle_grid_dims(const scvod_params* p, int32_t* range_num, int32_t* sector_num, int32_t* azimuth_num,
                      int32_t* bin_num) {
    Grid g = grid_of(*p);
    *range_num = g.range_num;
    *sector_num = g.sector_num;
    *azimuth_num = g.azimuth_num;
    *bin_num = g.bin_num;
}

int oracle_bin(const scvod_params* params, const float* xyzi, int32_t n, int32_t apply_filter, scvod_apri* apri_out,
               int32_t* src_idx, int32_t* n_kept, int32_t* rejected_idx, int32_t* n_rejected) {
    const scvod_params& P = *params;
    Grid g = grid_of(P);
    int kept = 0, rej = 0;
    for (int i = 0; i < n; i++) {
        P3 pt{xyzi[4 * i], xyzi[4 * i + 1], xyzi[4 * i + 2], xyzi[4 * i + 3]};
        float dis = pointDistance2d(pt);
        float angle = getPolarAngle(pt);
        float azimuth = getAzimuth(pt);
        if (apply_filter) {
            if (dis < P.min_dis || dis > P.max_dis || angle < P.min_angle || angle > P.max_angle ||
                azimuth < P.min_azimuth || azimuth > P.max_azimuth) {
                if (rejected_idx) rejected_idx[rej] = i;
                ++rej;
                continue;
            }
        }
        scvod_apri apri;
        bin_point(P, g, pt, dis, angle, azimuth, apri);
        if (apply_filter && apri.voxel_idx > g.bin_num) continue;  // ssc.cpp:189-192 (unreachable in practice)
        apri_out[kept] = apri;
        if (src_idx) src_idx[kept] = i;
        ++kept;
    }
    *n_kept = kept;
    if (n_rejected) *n_rejected = rej;
    return 0;
}

int oracle_voxelize(const scvod_params* params, const scvod_apri* apri, int32_t n, int32_t* vox_key,
                    int32_t* vox_pt_begin, int32_t* vox_pts, float* vox_av, float* vox_cov, int32_t* vox_idx3,
                    float* vox_center, int32_t* n_vox) {
    std::unordered_map<int, VoxelO> hash_cloud;
    make_hash_cloud(*params, apri, n, hash_cloud);
    std::vector<int> keys;
    keys.reserve(hash_cloud.size());
    for (auto& kv : hash_cloud) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    int off = 0;
    for (size_t v = 0; v < keys.size(); ++v) {
        const VoxelO& vx = hash_cloud[keys[v]];
        vox_key[v] = keys[v];
        vox_pt_begin[v] = off;
        for (int id : vx.ptIdx) vox_pts[off++] = id;
        vox_av[v] = vx.intensity_av;
        vox_cov[v] = vx.intensity_cov;
        if (vox_idx3) {
            vox_idx3[3 * v] = vx.range_idx;
            vox_idx3[3 * v + 1] = vx.sector_idx;
            vox_idx3[3 * v + 2] = vx.azimuth_idx;
        }
        if (vox_center) std::memcpy(vox_center + 4 * v, vx.center, 16);
    }
    vox_pt_begin[keys.size()] = off;
    *n_vox = (int32_t)keys.size();
    return 0;
}

void oracle_pose_delta(const float pose_pre[6], const float pose_next[6], float T_out[12]) {
    float tn[12], tp[12], tni[12];
    get_transformation(pose_next[0], pose_next[1], pose_next[2], pose_next[3], pose_next[4], pose_next[5], tn);
    get_transformation(pose_pre[0], pose_pre[1], pose_pre[2], pose_pre[3], pose_pre[4], pose_pre[5], tp);
    affine_inverse(tn, tni);
    affine_mul(tni, tp, T_out);
}

int oracle_track_probe(const scvod_params* params, const float* xyzi, const int32_t* offsets, int32_t n_clusters,
                       const float T[12], const int32_t* next_keys, const int32_t* next_labels, int32_t n_next_vox,
                       int32_t* hit_slot, int32_t* uniq_slots, int32_t* uniq_begin) {
    const scvod_params& P = *params;
    Grid g = grid_of(P);
    std::unordered_map<int, int> table;  // key -> slot
    for (int v = 0; v < n_next_vox; ++v) table.emplace(next_keys[v], v);
    int out = 0;
    for (int c = 0; c < n_clusters; ++c) {
        uniq_begin[c] = out;
        std::vector<int> hits;
        for (int k = offsets[c]; k < offsets[c + 1]; ++k) {
            const float* in = xyzi + 4 * k;
            P3 pt;  // Utility::transformCloud, utility.h:401-404
            pt.x = T[0] * in[0] + T[1] * in[1] + T[2] * in[2] + T[3];
            pt.y = T[4] * in[0] + T[5] * in[1] + T[6] * in[2] + T[7];
            pt.z = T[8] * in[0] + T[9] * in[1] + T[10] * in[2] + T[11];
            pt.intensity = in[3];
            float dis = pointDistance2d(pt);
            float angle = getPolarAngle(pt);
            float azimuth = getAzimuth(pt);
            int range_idx = std::ceil((dis - P.min_dis) / P.range_res) - 1;
            int sector_idx = std::ceil((angle - P.min_angle) / P.sector_res) - 1;
            int azimuth_idx = std::ceil((azimuth - P.min_azimuth) / P.azimuth_res) - 1;
            int voxel_idx = azimuth_idx * g.range_num * g.sector_num + range_idx * g.sector_num + sector_idx;
            auto it = table.find(voxel_idx);
            int slot = -1;
            if (it != table.end() && next_labels[it->second] != -1) slot = it->second;
            hit_slot[k] = slot;
            if (slot >= 0) hits.push_back(slot);
        }
        std::sort(hits.begin(), hits.end());  // sampleVec
        hits.erase(std::unique(hits.begin(), hits.end()), hits.end());
        for (int h : hits) uniq_slots[out++] = h;
    }
    uniq_begin[n_clusters] = out;
    return 0;
}

// info (optional), for the LITERAL max_name of ssc.cpp:354: {K = last running number handed out, index of the point that
// opened K, running number the opener's cluster carries at the end, openers whose clusters ended up in that cluster,
// mergeClusters calls that renamed K away, mergeClusters calls in total}
static int cluster_and_create_frame(const scvod_params* params, const scvod_apri* apri_vec_, int32_t n, int32_t* pt_cluster,
                                    int32_t* max_name, int64_t* info) {
    Grid g = grid_of(*params);
    std::vector<int> opened(n + 6, 1);  // per running number: how many openers' clusters it holds
    int last_opener = -1;
    int64_t merges = 0, k_renamed = 0;
    std::unordered_map<int, VoxelO> hash_cloud_;
    make_hash_cloud(*params, apri_vec_, n, hash_cloud_);
    int cluster_name = 4;
    std::vector<int> clusterIdxs(n, -1);
    for (int i = 0; i < n; i++) {
        const scvod_apri& apri = apri_vec_[i];
        std::vector<int> neighbors;
        auto it_find1 = hash_cloud_.find(apri.voxel_idx);
        if (it_find1 != hash_cloud_.end()) {
            std::vector<int> neighbor = find_voxel_neighbors(g, apri.range_idx, apri.sector_idx, apri.azimuth_idx, 1);
            for (size_t k = 0; k < neighbor.size(); k++) {
                auto it_find2 = hash_cloud_.find(neighbor[k]);
                if (it_find2 != hash_cloud_.end())
                    neighbors.insert(neighbors.end(), it_find2->second.ptIdx.begin(), it_find2->second.ptIdx.end());
            }
        }
        if (neighbors.size() > 0) {
            for (size_t nn = 0; nn < neighbors.size(); nn++) {
                int oc = clusterIdxs[i];
                int nc = clusterIdxs[neighbors[nn]];
                if (oc != -1 && nc != -1) {
                    if (oc != nc) {
                        for (int q = 0; q < n; q++)  // mergeClusters(clusterIdxs, oc, nc)
                            if (clusterIdxs[q] == oc) clusterIdxs[q] = nc;
                        opened[nc] += opened[oc];
                        merges++;
                        if (oc == cluster_name) k_renamed++;  // (cluster_name is the last number so far: only the final K matters, see below)
                    }
                } else {
                    if (nc != -1) {
                        clusterIdxs[i] = nc;
                    } else if (oc != -1) {
                        clusterIdxs[neighbors[nn]] = oc;
                    }
                }
            }
        }
        if (clusterIdxs[i] == -1) {
            cluster_name++;
            clusterIdxs[i] = cluster_name;
            for (size_t m = 0; m < neighbors.size(); m++) clusterIdxs[neighbors[m]] = cluster_name;
            last_opener = i;
            k_renamed = 0;
        }
    }
    if (max_name) *max_name = cluster_name;
    if (info) {
        info[0] = cluster_name;
        info[1] = last_opener;
        info[2] = last_opener >= 0 ? clusterIdxs[last_opener] : -1;
        info[3] = last_opener >= 0 ? opened[clusterIdxs[last_opener]] : 0;
        info[4] = k_renamed;
        info[5] = merges;
    }
    std::vector<int> names(clusterIdxs);
    std::sort(names.begin(), names.end());
    names.erase(std::unique(names.begin(), names.end()), names.end());
    std::memcpy(pt_cluster, clusterIdxs.data(), n * sizeof(int));
    return (int)names.size();
}

// `atan2` in utility.h:382,385,391 is unqualified inside a function template.  With libstdc++'s <cmath> / <math.h>
// in scope the float overload is chosen (= atan2f, what the restatement uses); a build that only saw the C prototype would
// promote to double and round the result, `float(atan2(double, double))`.  Counts how many points of a cloud change an INDEX
// between the two readings: counts[4] = {points inside the range / FOV filter under the float reading, of those: sector_idx
// differs, azimuth_idx differs, kept-or-rejected verdict differs}.
int oracle_atan2_overload_flips(const scvod_params* params, const float* xyzi, int32_t n, int64_t* counts) {
    const scvod_params& P = *params;
    for (int k = 0; k < 4; ++k) counts[k] = 0;
    auto deg = [](float rad) { return (float)((float)rad * 180.0 / M_PI); };
    for (int i = 0; i < n; ++i) {
        const float x = xyzi[4 * i], y = xyzi[4 * i + 1], z = xyzi[4 * i + 2];
        const float dis = (float)sqrt(x * x + y * y);
        float ang[2], azi[2];
        for (int v = 0; v < 2; ++v) {
            const float a = v == 0 ? atan2f(y, x) : (float)atan2((double)y, (double)x);
            if (x == 0 && y == 0)
                ang[v] = 0.f;
            else if (y >= 0)
                ang[v] = deg(a);
            else
                ang[v] = (float)((float)((float)a + 2 * M_PI) * 180.0 / M_PI);
            azi[v] = deg(v == 0 ? atan2f(z, dis) : (float)atan2((double)z, (double)dis));
        }
        auto kept = [&](int v) {
            return !(dis < P.min_dis || dis > P.max_dis || ang[v] < P.min_angle || ang[v] > P.max_angle || azi[v] < P.min_azimuth || azi[v] > P.max_azimuth);
        };
        const bool k0 = kept(0), k1 = kept(1);
        if (k0 != k1) counts[3]++;
        if (!k0) continue;
        counts[0]++;
        const int s0 = (int)(std::ceil((ang[0] - P.min_angle) / P.sector_res) - 1), s1 = (int)(std::ceil((ang[1] - P.min_angle) / P.sector_res) - 1);
        const int a0 = (int)(std::ceil((azi[0] - P.min_azimuth) / P.azimuth_res) - 1), a1 = (int)(std::ceil((azi[1] - P.min_azimuth) / P.azimuth_res) - 1);
        if (s0 != s1) counts[1]++;
        if (a0 != a1) counts[2]++;
    }
    return 0;
}

int oracle_cluster(const scvod_params* params, const scvod_apri* apri_vec_, int32_t n, int32_t* pt_cluster, int32_t* max_name) {
    return cluster_and_create_frame(params, apri_vec_, n, pt_cluster, max_name, nullptr);
}

// ssc.cpp:354 `frame_ssc.max_name = cluster_name ++;`: max_name is the LAST USED running number K.  Returns the smallest
// point index of the cluster that carries K when clusterAndCreateFrame ends (its canonical name in the product's naming), or
// -1 when no point carries K any more; info[6] as documented at cluster_and_create_frame.
int oracle_cluster_last_name(const scvod_params* params, const scvod_apri* apri_vec_, int32_t n, int64_t* info) {
    std::vector<int32_t> cl(n ? n : 1);
    int32_t K = 4;
    cluster_and_create_frame(params, apri_vec_, n, cl.data(), &K, info);
    for (int i = 0; i < n; ++i)
        if (cl[i] == K) return i;
    return -1;
}

// refineClusterByBoundingBox (ssc.cpp:437-467) + the bounding-box branch of recognize (ssc.cpp:849-872, features of
// getDescriptorByEigenValue ssc.cpp:723-751) for given per-point cluster names; building vs tree (PCL region growing)
// is not separated: both map to other_label.
int oracle_cluster_types(const scvod_params* params, const scvod_apri* apri, int32_t n, const int32_t* pt_cluster,
                         int32_t car_label, int32_t other_label, int32_t* pt_type) {
    struct Box {
        float mn[3], mx[3];
        int count = 0;
    };
    std::unordered_map<int, Box> boxes;  // pcl::getMinMax3D per cluster cloud
    for (int i = 0; i < n; ++i) {
        Box& b = boxes[pt_cluster[i]];
        const float p[3] = {apri[i].x, apri[i].y, apri[i].z};
        for (int k = 0; k < 3; ++k) {
            if (b.count == 0 || p[k] < b.mn[k]) b.mn[k] = p[k];
            if (b.count == 0 || p[k] > b.mx[k]) b.mx[k] = p[k];
        }
        b.count++;
    }
    std::unordered_map<int, int> type;
    for (auto& kv : boxes) {
        const Box& b = kv.second;
        float diff_z = b.mx[2] - b.mn[2];
        if (b.mn[2] > 0.f || (b.count < params->toBeClass) || diff_z < 0.2) {
            type[kv.first] = -1;  // erased, voxels relabelled -1
            continue;
        }
        double diff_x = b.mx[0] - b.mn[0], diff_y = b.mx[1] - b.mn[1];
        double square = diff_x * diff_y;                 // f_11(0,7)
        double f6 = b.mx[2], f9 = b.mn[2];               // f_11(0,6) = point_max.z, f_11(0,9) = point_min.z
        if (square > params->car_square)
            type[kv.first] = other_label;
        else if (f9 < params->min_z && square < params->car_square && f6 < params->max_z)
            type[kv.first] = car_label;
        else
            type[kv.first] = other_label;
    }
    for (int i = 0; i < n; ++i) pt_type[i] = type[pt_cluster[i]];
    return 0;
}

int oracle_nn_search(const float* map_xyz, int32_t n_map, const float* query_xyz, int32_t n_query, float radius,
                     int32_t* nn_idx, float* nn_sqdist, uint8_t* within) {
    const float r2 = radius * radius;
    for (int q = 0; q < n_query; ++q) {
        float best = 0.f;
        int bi = -1;
        for (int m = 0; m < n_map; ++m) {
            float dx = map_xyz[3 * m] - query_xyz[3 * q], dy = map_xyz[3 * m + 1] - query_xyz[3 * q + 1],
                  dz = map_xyz[3 * m + 2] - query_xyz[3 * q + 2];
            float d = (dx * dx + dy * dy) + dz * dz;
            if (bi < 0 || d < best) {
                best = d;
                bi = m;
            }
        }
        nn_idx[q] = bi;
        nn_sqdist[q] = best;
        within[q] = (bi >= 0 && best < r2) ? 1 : 0;  // FLANN's RadiusResultSet keeps dist < r^2
    }
    return 0;
}

float oracle_libm_atan2f(float y, float x) { return atan2f(y, x); }
double oracle_libm_atan2(double y, double x) { return atan2(y, x); }

int oracle_time_process(const scvod_params* params, const float* xyzi, const int32_t* offsets, int32_t n_scans,
                        double stage_s[3], int64_t* checksum) {
    using clk = std::chrono::steady_clock;
    stage_s[0] = stage_s[1] = stage_s[2] = 0.0;
    int64_t cs = 0;
    for (int s = 0; s < n_scans; ++s) {
        const float* pts = xyzi + 4 * (size_t)offsets[s];
        int n = offsets[s + 1] - offsets[s];
        std::vector<uint8_t> cls(n);
        std::vector<int32_t> g(n), ng(n);
        std::vector<scvod_patch_plane> planes(SCVOD_MAX_PATCHES);
        int32_t n_g = 0, n_ng = 0, n_p = 0;
        auto t0 = clk::now();
        oracle_patchwork(params, nullptr, pts, n, 0, cls.data(), g.data(), &n_g, ng.data(), &n_ng, planes.data(), &n_p);
        auto t1 = clk::now();
        std::vector<float> ngc(4 * (size_t)n_ng);
        for (int k = 0; k < n_ng; ++k) std::memcpy(&ngc[4 * (size_t)k], pts + 4 * (size_t)ng[k], 16);
        std::vector<scvod_apri> apri(n_ng);
        int32_t n_a = 0;
        oracle_bin(params, ngc.data(), n_ng, 1, apri.data(), nullptr, &n_a, nullptr, nullptr);
        auto t2 = clk::now();
        std::unordered_map<int, VoxelO> hash_cloud;
        make_hash_cloud(*params, apri.data(), n_a, hash_cloud);
        auto t3 = clk::now();
        stage_s[0] += std::chrono::duration<double>(t1 - t0).count();
        stage_s[1] += std::chrono::duration<double>(t2 - t1).count();
        stage_s[2] += std::chrono::duration<double>(t3 - t2).count();
        cs += n_g + 3 * (int64_t)n_ng + 7 * (int64_t)n_a + 11 * (int64_t)hash_cloud.size();
    }
    if (checksum) *checksum = cs;
    return 0;
}

}  // extern "C"
