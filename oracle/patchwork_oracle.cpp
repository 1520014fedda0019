// ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing in the product (libscvod, the host facade,
// bench.py's measured path) may include, link or call this file.  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
//
// CPU restatement of PatchWork<PointT>::estimate_ground and its helpers
// (/root/reference/include/patchwork.h:193-504), written from the reference's behaviour,
// sequential like the reference, calling glibc libm where the reference does.
// PARITY UNPINNED: the reference has no tests / golden vectors for this path and cannot be
// built here (ROS, PCL, Eigen, OpenCV, Boost absent: SURVEY.md 8c), so this restatement is
// pinned only by the known-answer cases in tests/ derived from the formulas.
//
// Third-party arithmetic restated (not under /root/reference):
//   pcl::computeMeanAndCovarianceMatrix  (PCL 1.8.1 common/impl/centroid.hpp, dense cloud,
//        float accumulators)                                   -> mean_and_covariance()
//   Eigen::JacobiSVD<MatrixXf>(cov, ComputeFullU) (Eigen 3.3.4) -> jacobi_svd3()
//   Eigen GEMV `points * normal_` -> fl(fl(x*n0 + y*n1) + z*n2), no FMA
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct Pt {
    float x, y, z, intensity;
    int32_t idx;  // index in the input cloud (bookkeeping only)
};

// ---------------------------------------------------------------------------------------
// Eigen 3.3 JacobiSVD for a 3x3 float matrix, ComputeFullU (matrix-style restatement).
// M is column-major like Eigen::MatrixXf.
struct Mat3 {
    float a[9];
    float& operator()(int r, int c) { return a[c * 3 + r]; }
    float operator()(int r, int c) const { return a[c * 3 + r]; }
};
struct Rot {
    float c, s;
};

static void rotate_rows(Mat3& m, int p, int q, Rot j) {  // applyOnTheLeft(p,q,j)
    if (j.c == 1.f && j.s == 0.f) return;
    for (int i = 0; i < 3; ++i) {
        float xi = m(p, i), yi = m(q, i);
        m(p, i) = j.c * xi + j.s * yi;
        m(q, i) = -j.s * xi + j.c * yi;
    }
}
static void rotate_cols(Mat3& m, int p, int q, Rot j) {  // applyOnTheRight(p,q,j): uses j^T
    Rot t{j.c, -j.s};
    if (t.c == 1.f && t.s == 0.f) return;
    for (int i = 0; i < 3; ++i) {
        float xi = m(i, p), yi = m(i, q);
        m(i, p) = t.c * xi + t.s * yi;
        m(i, q) = -t.s * xi + t.c * yi;
    }
}
static Rot make_jacobi(float x, float y, float z) {
    Rot r;
    float deno = 2.f * std::fabs(y);
    if (deno < std::numeric_limits<float>::min()) {
        r.c = 1.f;
        r.s = 0.f;
        return r;
    }
    float tau = (x - z) / deno;
    float w = std::sqrt(tau * tau + 1.f);
    float t = (tau > 0.f) ? 1.f / (tau + w) : 1.f / (tau - w);
    float sign_t = t > 0.f ? 1.f : -1.f;
    float n = 1.f / std::sqrt(t * t + 1.f);
    r.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
    r.c = n;
    return r;
}

static void jacobi_svd3(const float cov_colmajor[9], float sv[3], float U_colmajor[9]) {
    const float precision = 2.f * std::numeric_limits<float>::epsilon();
    const float consider_as_zero = std::numeric_limits<float>::min();
    Mat3 W, U;
    float scale = 0.f;
    for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(cov_colmajor[i]));
    if (scale == 0.f) scale = 1.f;
    for (int i = 0; i < 9; ++i) W.a[i] = cov_colmajor[i] / scale;
    for (int i = 0; i < 9; ++i) U.a[i] = 0.f;
    U(0, 0) = U(1, 1) = U(2, 2) = 1.f;
    float max_diag = std::max(std::fabs(W(0, 0)), std::max(std::fabs(W(1, 1)), std::fabs(W(2, 2))));
    bool finished = false;
    int sweeps = 0;
    while (!finished && sweeps++ < 64) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                float threshold = std::max(consider_as_zero, precision * max_diag);
                if (std::fabs(W(p, q)) > threshold || std::fabs(W(q, p)) > threshold) {
                    finished = false;
                    // internal::real_2x2_jacobi_svd
                    float m[2][2] = {{W(p, p), W(p, q)}, {W(q, p), W(q, q)}};
                    Rot rot1;
                    float t = m[0][0] + m[1][1];
                    float d = m[1][0] - m[0][1];
                    if (std::fabs(d) < std::numeric_limits<float>::min()) {
                        rot1.s = 0.f;
                        rot1.c = 1.f;
                    } else {
                        float u = t / d;
                        float tmp = std::sqrt(1.f + u * u);
                        rot1.s = 1.f / tmp;
                        rot1.c = u / tmp;
                    }
                    if (!(rot1.c == 1.f && rot1.s == 0.f)) {
                        for (int i = 0; i < 2; ++i) {
                            float xi = m[0][i], yi = m[1][i];
                            m[0][i] = rot1.c * xi + rot1.s * yi;
                            m[1][i] = -rot1.s * xi + rot1.c * yi;
                        }
                    }
                    Rot j_right = make_jacobi(m[0][0], m[0][1], m[1][1]);
                    Rot jrt{j_right.c, -j_right.s};
                    Rot j_left{rot1.c * jrt.c - rot1.s * jrt.s, rot1.c * jrt.s + rot1.s * jrt.c};
                    rotate_rows(W, p, q, j_left);
                    rotate_cols(U, p, q, Rot{j_left.c, -j_left.s});  // j_left.transpose()
                    rotate_cols(W, p, q, j_right);
                    max_diag = std::max(max_diag, std::max(std::fabs(W(p, p)), std::fabs(W(q, q))));
                }
            }
    }
    for (int i = 0; i < 3; ++i) {
        float a = W(i, i);
        sv[i] = std::fabs(a);
        if (a < 0.f)
            for (int r = 0; r < 3; ++r) U(r, i) = -U(r, i);
    }
    for (int i = 0; i < 3; ++i) sv[i] *= scale;
    for (int i = 0; i < 3; ++i) {
        int pos = 0;
        float mx = sv[i];
        for (int j = 1; j < 3 - i; ++j)
            if (sv[i + j] > mx) {
                mx = sv[i + j];
                pos = j;
            }
        if (mx == 0.f) break;
        if (pos) {
            pos += i;
            std::swap(sv[i], sv[pos]);
            for (int r = 0; r < 3; ++r) std::swap(U(r, i), U(r, pos));
        }
    }
    std::memcpy(U_colmajor, U.a, sizeof(U.a));
}

// how often an estimate_plane_ call met an EMPTY ground set: the one situation in which the reference's plane state leaks
// from the previous patch (and, through the static object, from the previous scan).  The product does not model the leak;
// tests/test_oracle_known_answers.py shows the count stays 0 (and DESIGN.md why it must for th_seeds >= 0, th_dist >= 0.01).
static long long g_empty_ground_sets = 0;

// ---------------------------------------------------------------------------------------
struct PatchworkState {  // members of class PatchWork that survive between calls
    float d_ = 0.f;
    float normal_[3] = {0.f, 0.f, 1.f};
    float singular_values_[3] = {0.f, 0.f, 0.f};
    float th_dist_d_ = 0.f;
    float cov_[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // column-major
    float pc_mean_[4] = {0.f, 0.f, 0.f, 1.f};
};

class PatchworkOracle {
  public:
    PatchworkOracle(const scvod_pw_params& pw, double sensor_height) : pw_(pw), sensor_height_(sensor_height) {
        // patchwork.h:83-94
        min_range_z2_ = (7 * pw_.min_range + pw_.max_range) / 8.0;
        min_range_z3_ = (3 * pw_.min_range + pw_.max_range) / 4.0;
        min_range_z4_ = (pw_.min_range + pw_.max_range) / 2.0;
        min_ranges_[0] = pw_.min_range;
        min_ranges_[1] = min_range_z2_;
        min_ranges_[2] = min_range_z3_;
        min_ranges_[3] = min_range_z4_;
        ring_sizes_[0] = (min_range_z2_ - pw_.min_range) / pw_.num_rings_each_zone[0];
        ring_sizes_[1] = (min_range_z3_ - min_range_z2_) / pw_.num_rings_each_zone[1];
        ring_sizes_[2] = (min_range_z4_ - min_range_z3_) / pw_.num_rings_each_zone[2];
        ring_sizes_[3] = (pw_.max_range - min_range_z4_) / pw_.num_rings_each_zone[3];
        for (int k = 0; k < 4; ++k) sector_sizes_[k] = 2 * M_PI / pw_.num_sectors_each_zone[k];
        int base = 0;
        for (int k = 0; k < 4; ++k) {
            patch_base_[k] = base;
            base += pw_.num_rings_each_zone[k] * pw_.num_sectors_each_zone[k];
        }
        num_patches_ = base;
    }

    int num_patches() const { return num_patches_; }

    // patch id of one point as pc2czm + the z cut assign it (-1: not binned); for the spec-vs-libm test
    int patch_of(float xf, float yf, float zf) {
        if (zf < -1.8 * sensor_height_) return -1;
        double r = xy2radius(xf, yf);
        if (!((r <= pw_.max_range) && (r > pw_.min_range))) return -1;
        double theta = xy2theta(xf, yf);
        int k = (r < min_range_z2_) ? 0 : (r < min_range_z3_) ? 1 : (r < min_range_z4_) ? 2 : 3;
        int ring_idx = std::min(static_cast<int>(((r - min_ranges_[k]) / ring_sizes_[k])), pw_.num_rings_each_zone[k] - 1);
        int sector_idx = std::min(static_cast<int>((theta / sector_sizes_[k])), pw_.num_sectors_each_zone[k] - 1);
        if (sector_idx < 0) sector_idx = 0;
        return patch_base_[k] + ring_idx * pw_.num_sectors_each_zone[k] + sector_idx;
    }

    // sort_mode 0: std::sort with `a.z < b.z` exactly as patchwork.h:295 (tie order is whatever
    //              libstdc++'s introsort yields);
    // sort_mode 1: ties broken by input index (the canonical order the GPU path implements).
    void estimate_ground(const float* xyzi, int n, int sort_mode, std::vector<int32_t>& ground_idx,
                         std::vector<int32_t>& nonground_idx, std::vector<uint8_t>& cls,
                         std::vector<scvod_patch_plane>& planes) {
        std::vector<Pt> cloud(n);
        for (int i = 0; i < n; ++i) cloud[i] = Pt{xyzi[4 * i], xyzi[4 * i + 1], xyzi[4 * i + 2], xyzi[4 * i + 3], i};
        // 2. sort on z (patchwork.h:295)
        if (sort_mode == 0)
            std::sort(cloud.begin(), cloud.end(), [](const Pt& a, const Pt& b) { return a.z < b.z; });
        else
            std::sort(cloud.begin(), cloud.end(), [](const Pt& a, const Pt& b) {
                if (a.z < b.z) return true;
                if (b.z < a.z) return false;
                // -0.0 sorts before +0.0 in the canonical order (float_sort_key)
                bool an = std::signbit(a.z), bn = std::signbit(b.z);
                if (an != bn) return an;
                return a.idx < b.idx;
            });
        // 3. error point removal (patchwork.h:302-310)
        size_t first = 0;
        for (size_t i = 0; i < cloud.size(); ++i) {
            if (cloud[i].z < -1.8 * sensor_height_)
                ++first;
            else
                break;
        }
        cloud.erase(cloud.begin(), cloud.begin() + first);
        // 4. pc2czm (patchwork.h:431-459)
        std::vector<std::vector<Pt>> patches(num_patches_);
        for (const Pt& pt : cloud) {
            int ring_idx, sector_idx;
            double r = xy2radius(pt.x, pt.y);
            if ((r <= pw_.max_range) && (r > pw_.min_range)) {
                double theta = xy2theta(pt.x, pt.y);
                int k;
                if (r < min_range_z2_)
                    k = 0;
                else if (r < min_range_z3_)
                    k = 1;
                else if (r < min_range_z4_)
                    k = 2;
                else
                    k = 3;
                ring_idx = std::min(static_cast<int>(((r - min_ranges_[k]) / ring_sizes_[k])), pw_.num_rings_each_zone[k] - 1);
                sector_idx = std::min(static_cast<int>((theta / sector_sizes_[k])), pw_.num_sectors_each_zone[k] - 1);
                if (sector_idx < 0) sector_idx = 0;  // y == -0.0f, x < 0: theta = -pi; the reference indexes out of bounds here
                patches[patch_base_[k] + ring_idx * pw_.num_sectors_each_zone[k] + sector_idx].emplace_back(pt);
            }
        }
        ground_idx.clear();
        nonground_idx.clear();
        cls.assign(n, SCVOD_CLS_DROPPED);
        planes.assign(num_patches_, scvod_patch_plane{});
        std::vector<Pt> regionwise_ground, regionwise_nonground;
        int concentric_idx = 0;
        for (int k = 0; k < 4; ++k) {
            for (int ring_idx = 0; ring_idx < pw_.num_rings_each_zone[k]; ++ring_idx) {
                for (int sector_idx = 0; sector_idx < pw_.num_sectors_each_zone[k]; ++sector_idx) {
                    int pid = patch_base_[k] + ring_idx * pw_.num_sectors_each_zone[k] + sector_idx;
                    const std::vector<Pt>& patch = patches[pid];
                    scvod_patch_plane& rec = planes[pid];
                    rec.n_pts = (int32_t)patch.size();
                    if ((int)patch.size() > pw_.num_min_pts) {
                        extract_piecewiseground(k, patch, regionwise_ground, regionwise_nonground);
                        const double ground_z_vec = std::abs(st_.normal_[2]);
                        const double ground_z_elevation = st_.pc_mean_[2];
                        const float sv_min = std::min(st_.singular_values_[0], std::min(st_.singular_values_[1], st_.singular_values_[2]));
                        const double surface_variable =
                            sv_min / (st_.singular_values_[0] + st_.singular_values_[1] + st_.singular_values_[2]);
                        for (int c = 0; c < 3; ++c) {
                            rec.normal[c] = st_.normal_[c];
                            rec.mean[c] = st_.pc_mean_[c];
                            rec.sv[c] = st_.singular_values_[c];
                        }
                        rec.n_ground = (int32_t)regionwise_ground.size();
                        bool ground_to_nonground;
                        if (ground_z_vec < pw_.uprightness_thr) {
                            ground_to_nonground = true;
                            rec.status = 2;
                        } else if (concentric_idx < pw_.num_rings_of_interest) {
                            if (ground_z_elevation > pw_.elevation_thr[ring_idx + 2 * k]) {
                                if (pw_.flatness_thr[ring_idx + 2 * k] > surface_variable) {
                                    ground_to_nonground = false;
                                    rec.status = 1;
                                } else {
                                    ground_to_nonground = true;
                                    rec.status = 3;
                                }
                            } else {
                                ground_to_nonground = false;
                                rec.status = 1;
                            }
                        } else {
                            ground_to_nonground = false;
                            rec.status = 1;
                        }
                        if (ground_to_nonground) {
                            for (const Pt& p : regionwise_ground) emit(nonground_idx, cls, p, SCVOD_CLS_NONGROUND);
                        } else {
                            for (const Pt& p : regionwise_ground) emit(ground_idx, cls, p, SCVOD_CLS_GROUND);
                        }
                        for (const Pt& p : regionwise_nonground) emit(nonground_idx, cls, p, SCVOD_CLS_NONGROUND);
                    }
                }
                ++concentric_idx;
            }
        }
    }

  private:
    static void emit(std::vector<int32_t>& dst, std::vector<uint8_t>& cls, const Pt& p, uint8_t c) {
        dst.push_back(p.idx);
        cls[p.idx] = c;
    }
    double xy2theta(const double& x, const double& y) {  // patchwork.h:416-423
        if (y >= 0) return atan2(y, x);
        return 2 * M_PI + atan2(y, x);
    }
    double xy2radius(const double& x, const double& y) { return sqrt(pow(x, 2) + pow(y, 2)); }

    // pcl::computeMeanAndCovarianceMatrix, dense cloud, Scalar = float (PCL 1.8.1)
    void mean_and_covariance(const std::vector<Pt>& cloud) {
        float accu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        size_t point_count = cloud.size();
        for (size_t i = 0; i < point_count; ++i) {
            accu[0] += cloud[i].x * cloud[i].x;
            accu[1] += cloud[i].x * cloud[i].y;
            accu[2] += cloud[i].x * cloud[i].z;
            accu[3] += cloud[i].y * cloud[i].y;
            accu[4] += cloud[i].y * cloud[i].z;
            accu[5] += cloud[i].z * cloud[i].z;
            accu[6] += cloud[i].x;
            accu[7] += cloud[i].y;
            accu[8] += cloud[i].z;
        }
        if (point_count != 0) {
            for (int i = 0; i < 9; ++i) accu[i] /= static_cast<float>(point_count);
            st_.pc_mean_[0] = accu[6];
            st_.pc_mean_[1] = accu[7];
            st_.pc_mean_[2] = accu[8];
            st_.pc_mean_[3] = 1;
            float* c = st_.cov_;  // coeffRef(k) on a column-major 3x3
            c[0] = accu[0] - accu[6] * accu[6];
            c[1] = accu[1] - accu[6] * accu[7];
            c[2] = accu[2] - accu[6] * accu[8];
            c[4] = accu[3] - accu[7] * accu[7];
            c[5] = accu[4] - accu[7] * accu[8];
            c[8] = accu[5] - accu[8] * accu[8];
            c[3] = c[1];
            c[6] = c[2];
            c[7] = c[5];
        } else {
            ++g_empty_ground_sets;  // n == 0: cov_ / pc_mean_ keep their previous values (PCL leaves its outputs untouched)
        }
    }

    void estimate_plane_(const std::vector<Pt>& ground) {  // patchwork.h:216-232
        mean_and_covariance(ground);
        float U[9];
        jacobi_svd3(st_.cov_, st_.singular_values_, U);
        for (int r = 0; r < 3; ++r) st_.normal_[r] = U[2 * 3 + r];  // matrixU().col(2)
        // d_ = -(normal_.transpose() * seeds_mean)(0,0)
        float dot = st_.normal_[0] * st_.pc_mean_[0];
        dot = dot + st_.normal_[1] * st_.pc_mean_[1];
        dot = dot + st_.normal_[2] * st_.pc_mean_[2];
        st_.d_ = -dot;
        st_.th_dist_d_ = (float)(pw_.th_dist - (double)st_.d_);
    }

    void extract_initial_seeds_(int zone_idx, const std::vector<Pt>& p_sorted, std::vector<Pt>& init_seeds) {
        init_seeds.clear();
        double sum = 0;
        int cnt = 0;
        int init_idx = 0;
        if (zone_idx == 0) {
            for (size_t i = 0; i < p_sorted.size(); i++) {
                if (p_sorted[i].z < pw_.adaptive_seed_selection_margin * sensor_height_)
                    ++init_idx;
                else
                    break;
            }
        }
        for (size_t i = init_idx; i < p_sorted.size() && cnt < pw_.num_lpr; i++) {
            sum += p_sorted[i].z;
            cnt++;
        }
        double lpr_height = cnt != 0 ? sum / cnt : 0;
        for (size_t i = 0; i < p_sorted.size(); i++)
            if (p_sorted[i].z < lpr_height + pw_.th_seeds) init_seeds.push_back(p_sorted[i]);
    }

    void extract_piecewiseground(int zone_idx, const std::vector<Pt>& src, std::vector<Pt>& dst, std::vector<Pt>& non_ground_dst) {
        std::vector<Pt> ground_pc;
        dst.clear();
        non_ground_dst.clear();
        extract_initial_seeds_(zone_idx, src, ground_pc);
        for (int i = 0; i < pw_.num_iter; i++) {
            estimate_plane_(ground_pc);
            ground_pc.clear();
            for (size_t r = 0; r < src.size(); r++) {
                // Eigen: result = points * normal_  -> fl(fl(x*n0 + y*n1) + z*n2)
                float res = src[r].x * st_.normal_[0];
                res = res + src[r].y * st_.normal_[1];
                res = res + src[r].z * st_.normal_[2];
                if (i < pw_.num_iter - 1) {
                    if (res < st_.th_dist_d_) ground_pc.push_back(src[r]);
                } else {
                    if (res < st_.th_dist_d_)
                        dst.push_back(src[r]);
                    else
                        non_ground_dst.push_back(src[r]);
                }
            }
        }
    }

    scvod_pw_params pw_;
    double sensor_height_;
    double min_range_z2_, min_range_z3_, min_range_z4_;
    double min_ranges_[4], ring_sizes_[4], sector_sizes_[4];
    int patch_base_[4];
    int num_patches_;
    PatchworkState st_;
};

}  // namespace

extern "C" {

int oracle_patchwork(const scvod_params* params, const scvod_pw_params* pw_in, const float* xyzi, int32_t n,
                     int32_t sort_mode, uint8_t* cls, int32_t* ground_idx, int32_t* n_ground, int32_t* nonground_idx,
                     int32_t* n_nonground, scvod_patch_plane* planes, int32_t* n_patches) {
    scvod_pw_params pw;
    if (pw_in)
        pw = *pw_in;
    else
        oracle_pw_params_default(&pw);
    // set_sensor(const double&) receives the float YAML value (ssc.cpp:93)
    PatchworkOracle po(pw, (double)params->sensor_height);
    std::vector<int32_t> g, ng;
    std::vector<uint8_t> c;
    std::vector<scvod_patch_plane> pl;
    po.estimate_ground(xyzi, n, sort_mode, g, ng, c, pl);
    if (cls) std::memcpy(cls, c.data(), c.size());
    if (ground_idx) std::memcpy(ground_idx, g.data(), g.size() * 4);
    if (nonground_idx) std::memcpy(nonground_idx, ng.data(), ng.size() * 4);
    if (n_ground) *n_ground = (int32_t)g.size();
    if (n_nonground) *n_nonground = (int32_t)ng.size();
    if (planes) std::memcpy(planes, pl.data(), pl.size() * sizeof(scvod_patch_plane));
    if (n_patches) *n_patches = po.num_patches();
    return 0;
}

void oracle_patch_ids(const scvod_params* params, const float* xyzi, int32_t n, int32_t* pid) {
    scvod_pw_params pw;
    oracle_pw_params_default(&pw);
    PatchworkOracle po(pw, (double)params->sensor_height);
    for (int i = 0; i < n; ++i) pid[i] = po.patch_of(xyzi[4 * i], xyzi[4 * i + 1], xyzi[4 * i + 2]);
}

// exposed for the SVD known-answer tests; cov row-major 3x3 (symmetric in practice)
void oracle_svd3(const float cov_rowmajor[9], float sv[3], float U_rowmajor[9]) {
    float cm[9], Ucm[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) cm[c * 3 + r] = cov_rowmajor[r * 3 + c];
    jacobi_svd3(cm, sv, Ucm);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) U_rowmajor[r * 3 + c] = Ucm[c * 3 + r];
}

long long oracle_patchwork_empty_sets(int reset) {
    const long long v = g_empty_ground_sets;
    if (reset) g_empty_ground_sets = 0;
    return v;
}

}  // extern "C"
