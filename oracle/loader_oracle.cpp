/* ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Never linked into libscvod.so.
 *
 * SURVEY 8(f)-3, the step in front of the hot path: SSC::getCloud's label filter and intensity scaling
 * (src/ssc.cpp:1063-1076) followed by pcl::VoxelGrid<pcl::PointXYZI> with a 0.08 m leaf (src/ssc.cpp:1103-1106).
 *
 * pcl::VoxelGrid is a third-party dependency that is not in /root/reference (ROS melodic ships PCL 1.8.1); its
 * published algorithm is restated here from filters/include/pcl/filters/impl/voxel_grid.hpp (applyFilter),
 * common/include/pcl/common/impl/common.hpp (getMinMax3D) and common/include/pcl/common/impl/accumulators.hpp
 * (CentroidPoint: AccumulatorXYZ, AccumulatorIntensity).  PARITY UNPINNED: there is no PCL here to run it against.
 *
 *   inverse_leaf = 1.f / leaf (fp32, per axis)
 *   min_p / max_p = component-wise min / max over the cloud (is_dense: no finite test)
 *   dx = int64((max - min) * inverse_leaf) + 1 per axis; dx*dy*dz > INT32_MAX -> output = input (warning)
 *   min_b = int(floor(min * inverse_leaf)), max_b likewise, div_b = max_b - min_b + 1, mul = (1, div0, div0*div1)
 *   idx(point) = sum_k int(floor(p_k * inverse_leaf_k) - float(min_b_k)) * mul_k
 *   std::sort by idx (operator< looks at idx only: the order INSIDE a voxel is unspecified), one output point per
 *   distinct idx in ascending idx order = CentroidPoint over the voxel's points in sorted order: fp32 running sums
 *   of x, y, z, intensity, each divided by float(count).
 * sort_mode 0 = std::sort as the reference (implementation-defined order inside a voxel), 1 = canonical
 * (idx, input index) -- what the GPU implements; the two differ only in the last bit of sums of >= 3 points. */
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace {
struct IndexIdx {
    unsigned int idx;
    unsigned int cloud_point_index;
    bool operator<(const IndexIdx& p) const { return idx < p.idx; }
};
}  // namespace

extern "C" int oracle_voxelgrid(const float* xyzi, const uint32_t* labels, int32_t n, float max_intensity, const float leaf[3],
                                int32_t sort_mode, float* out_xyzi, int32_t* n_out) {
    // ---- loader: label filter + intensity scaling (ssc.cpp:1063-1076); the distance test `dis >= min_dis ||
    // dis <= max_dis` (ssc.cpp:1095) is true for every finite point and is not modelled
    std::vector<float> cloud;
    cloud.reserve((size_t)n * 4);
    for (int k = 0; k < n; ++k) {
        float inten = xyzi[4 * k + 3];
        if (labels) {
            const uint32_t l = labels[k] & 0xFFFFu;
            if (l == 0 || l == 1) continue;  // unlabeled / outlier
            inten = inten * max_intensity;
        }
        cloud.push_back(xyzi[4 * k]);
        cloud.push_back(xyzi[4 * k + 1]);
        cloud.push_back(xyzi[4 * k + 2]);
        cloud.push_back(inten);
    }
    const int m = (int)(cloud.size() / 4);
    *n_out = 0;
    if (m == 0) return 0;
    float inv[3];
    for (int a = 0; a < 3; ++a) inv[a] = 1.0f / leaf[a];
    // getMinMax3D
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < m; ++i)
        for (int a = 0; a < 3; ++a) {
            const float v = cloud[4 * i + a];
            mn[a] = v < mn[a] ? v : mn[a];
            mx[a] = v > mx[a] ? v : mx[a];
        }
    const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv[0]) + 1;
    const int64_t dy = (int64_t)((mx[1] - mn[1]) * inv[1]) + 1;
    const int64_t dz = (int64_t)((mx[2] - mn[2]) * inv[2]) + 1;
    if (dx * dy * dz > (int64_t)std::numeric_limits<int32_t>::max()) {
        for (int i = 0; i < 4 * m; ++i) out_xyzi[i] = cloud[i];  // "Leaf size is too small": output = input
        *n_out = m;
        return 1;
    }
    int min_b[3], max_b[3], div_b[3];
    for (int a = 0; a < 3; ++a) {
        min_b[a] = (int)std::floor(mn[a] * inv[a]);
        max_b[a] = (int)std::floor(mx[a] * inv[a]);
        div_b[a] = max_b[a] - min_b[a] + 1;
    }
    const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
    std::vector<IndexIdx> iv;
    iv.reserve(m);
    for (int i = 0; i < m; ++i) {
        int idx = 0;
        for (int a = 0; a < 3; ++a) {
            const int ijk = (int)(std::floor(cloud[4 * i + a] * inv[a]) - (float)min_b[a]);
            idx += ijk * mul[a];
        }
        iv.push_back({(unsigned int)idx, (unsigned int)i});
    }
    if (sort_mode == 0)
        std::sort(iv.begin(), iv.end(), std::less<IndexIdx>());
    else
        std::sort(iv.begin(), iv.end(), [](const IndexIdx& a, const IndexIdx& b) {
            return a.idx != b.idx ? a.idx < b.idx : a.cloud_point_index < b.cloud_point_index;
        });
    int out = 0;
    size_t index = 0;
    while (index < iv.size()) {
        size_t i = index + 1;
        while (i < iv.size() && iv[i].idx == iv[index].idx) ++i;
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;  // CentroidPoint: Vector3f xyz, float intensity
        for (size_t li = index; li < i; ++li) {
            const float* p = &cloud[4 * (size_t)iv[li].cloud_point_index];
            sx += p[0];
            sy += p[1];
            sz += p[2];
            si += p[3];
        }
        const size_t cnt = i - index;
        out_xyzi[4 * out] = sx / cnt;
        out_xyzi[4 * out + 1] = sy / cnt;
        out_xyzi[4 * out + 2] = sz / cnt;
        out_xyzi[4 * out + 3] = si / cnt;
        ++out;
        index = i;
    }
    *n_out = out;
    return 0;
}

// SSC::getPose, KITTI branch (src/ssc.cpp:960-989) + Utility::rotationMatrixToEulerAngles (include/utility.h:488-505) for
// one line of poses.txt: velo_to_cam = tr.inverse() * cam * tr in float, pose = {translation, roll, pitch, yaw}.
// Eigen::Matrix4f::inverse() restated as cofactors from 2x2 sub-determinants over the determinant (Eigen's generic 4x4
// path; the SSE kernel x86 builds of Eigen 3.3 use rounds differently in the last bits): PARITY UNPINNED, the tests allow
// 1e-5 relative against a float64 evaluation.
extern "C" int oracle_kitti_pose(const float tr[16], const float cam12[12], float pose6[6], float velo_to_cam[16]) {
    auto at = [](const float* m, int r, int c) { return m[4 * r + c]; };
    const float* a = tr;
    // 2x2 sub-determinants of the two upper and the two lower rows
    const float s0 = at(a, 0, 0) * at(a, 1, 1) - at(a, 1, 0) * at(a, 0, 1), s1 = at(a, 0, 0) * at(a, 1, 2) - at(a, 1, 0) * at(a, 0, 2);
    const float s2 = at(a, 0, 0) * at(a, 1, 3) - at(a, 1, 0) * at(a, 0, 3), s3 = at(a, 0, 1) * at(a, 1, 2) - at(a, 1, 1) * at(a, 0, 2);
    const float s4 = at(a, 0, 1) * at(a, 1, 3) - at(a, 1, 1) * at(a, 0, 3), s5 = at(a, 0, 2) * at(a, 1, 3) - at(a, 1, 2) * at(a, 0, 3);
    const float c5 = at(a, 2, 2) * at(a, 3, 3) - at(a, 3, 2) * at(a, 2, 3), c4 = at(a, 2, 1) * at(a, 3, 3) - at(a, 3, 1) * at(a, 2, 3);
    const float c3 = at(a, 2, 1) * at(a, 3, 2) - at(a, 3, 1) * at(a, 2, 2), c2 = at(a, 2, 0) * at(a, 3, 3) - at(a, 3, 0) * at(a, 2, 3);
    const float c1 = at(a, 2, 0) * at(a, 3, 2) - at(a, 3, 0) * at(a, 2, 2), c0 = at(a, 2, 0) * at(a, 3, 1) - at(a, 3, 0) * at(a, 2, 1);
    const float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    if (det == 0.f) return -1;
    const float id = 1.0f / det;
    float inv[16];
    inv[0] = (at(a, 1, 1) * c5 - at(a, 1, 2) * c4 + at(a, 1, 3) * c3) * id;
    inv[1] = (-at(a, 0, 1) * c5 + at(a, 0, 2) * c4 - at(a, 0, 3) * c3) * id;
    inv[2] = (at(a, 3, 1) * s5 - at(a, 3, 2) * s4 + at(a, 3, 3) * s3) * id;
    inv[3] = (-at(a, 2, 1) * s5 + at(a, 2, 2) * s4 - at(a, 2, 3) * s3) * id;
    inv[4] = (-at(a, 1, 0) * c5 + at(a, 1, 2) * c2 - at(a, 1, 3) * c1) * id;
    inv[5] = (at(a, 0, 0) * c5 - at(a, 0, 2) * c2 + at(a, 0, 3) * c1) * id;
    inv[6] = (-at(a, 3, 0) * s5 + at(a, 3, 2) * s2 - at(a, 3, 3) * s1) * id;
    inv[7] = (at(a, 2, 0) * s5 - at(a, 2, 2) * s2 + at(a, 2, 3) * s1) * id;
    inv[8] = (at(a, 1, 0) * c4 - at(a, 1, 1) * c2 + at(a, 1, 3) * c0) * id;
    inv[9] = (-at(a, 0, 0) * c4 + at(a, 0, 1) * c2 - at(a, 0, 3) * c0) * id;
    inv[10] = (at(a, 3, 0) * s4 - at(a, 3, 1) * s2 + at(a, 3, 3) * s0) * id;
    inv[11] = (-at(a, 2, 0) * s4 + at(a, 2, 1) * s2 - at(a, 2, 3) * s0) * id;
    inv[12] = (-at(a, 1, 0) * c3 + at(a, 1, 1) * c1 - at(a, 1, 2) * c0) * id;
    inv[13] = (at(a, 0, 0) * c3 - at(a, 0, 1) * c1 + at(a, 0, 2) * c0) * id;
    inv[14] = (-at(a, 3, 0) * s3 + at(a, 3, 1) * s1 - at(a, 3, 2) * s0) * id;
    inv[15] = (at(a, 2, 0) * s3 - at(a, 2, 1) * s1 + at(a, 2, 2) * s0) * id;
    float cam[16];
    for (int i = 0; i < 12; ++i) cam[i] = cam12[i];
    cam[12] = cam[13] = cam[14] = 0.f;
    cam[15] = 1.f;
    auto mul = [&](const float* x, const float* y, float* z) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float acc = 0.f;
                for (int k = 0; k < 4; ++k) acc += x[4 * i + k] * y[4 * k + j];
                z[4 * i + j] = acc;
            }
    };
    float t1[16];
    mul(inv, cam, t1);
    mul(t1, tr, velo_to_cam);
    const float* R = velo_to_cam;
    pose6[0] = at(R, 0, 3);
    pose6[1] = at(R, 1, 3);
    pose6[2] = at(R, 2, 3);
    const float sy = std::sqrt(at(R, 0, 0) * at(R, 0, 0) + at(R, 1, 0) * at(R, 1, 0));
    if (!(sy < 1e-6)) {
        pose6[3] = atan2f(at(R, 2, 1), at(R, 2, 2));
        pose6[4] = atan2f(-at(R, 2, 0), sy);
        pose6[5] = atan2f(at(R, 1, 0), at(R, 0, 0));
    } else {
        pose6[3] = atan2f(-at(R, 1, 2), at(R, 1, 1));
        pose6[4] = atan2f(-at(R, 2, 0), sy);
        pose6[5] = 0;
    }
    return 0;
}
