// utility.h -- host-side mirror of the reference's `class Utility` configuration surface
// (/root/reference/include/utility.h:187-327): same member names, same two-level keys with trailing
// underscore, same nh.param<> defaults; values come from the YAML file directly (yaml_lite.h) instead
// of the ROS parameter server, so the same config/*.yaml files work with or without ROS.
#ifndef SCVOD_HOST_UTILITY_H_
#define SCVOD_HOST_UTILITY_H_
#include <cmath>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/scvod.h"
#include "pcl_shim.h"
#include "yaml_lite.h"

// include/utility.h:77-93 (PointXYZIRPYT)
struct Pose {
    float x = 0, y = 0, z = 0, intensity = 0, roll = 0, pitch = 0, yaw = 0;
    double time = 0;
};
typedef scvod_apri PointAPRI;  // include/utility.h:96-106, identical layout

struct Voxel {  // include/utility.h:109-119
    int range_idx, sector_idx, azimuth_idx;
    int label = -1;
    pcl::PointXYZI center;
    std::vector<int> ptIdx;
    std::vector<float> intensity_record;
    float intensity_av = 0.f;
    float intensity_cov = 0.f;
};

struct Cluster {  // include/utility.h:142-162 (fields the hot path touches)
    int track_id = -1, name = -1, type = -1, state = -1;
    int color[3] = {0, 0, 0};
    std::vector<int> occupy_pts, occupy_voxels;
    pcl::PointCloud<pcl::PointXYZI>::Ptr cloud{new pcl::PointCloud<pcl::PointXYZI>()};
};

struct Frame {  // include/utility.h:165-185
    int id = 0, max_name = 0;
    pcl::PointCloud<pcl::PointXYZI>::Ptr cloud_use{new pcl::PointCloud<pcl::PointXYZI>()};
    std::unordered_map<int, Voxel> hash_cloud;
    std::unordered_map<int, Cluster> cluster_set;
};

class Utility {
  public:
    std::string out_path;
    int kNumOmpCores = 6;
    bool save = true, mapping_init = false, is_pcd = false;
    int skip = 2;
    std::string data_path, label_path, pose_path;
    int init = 5, start = 5, end = 50;
    float sensor_height = 2.0f, min_dis = 0.0f, max_dis = 50.0f, min_angle = 0.0f, max_angle = 360.0f,
          min_azimuth = -30.0f, max_azimuth = 60.0f, range_res = 0.2f, sector_res = 1.2f, azimuth_res = 2.0f,
          refine_height = -1.0f, max_z = 1.0f, min_z = -1.0f, car_angle = 120.0f, car_height = 2.0f, car_square = 2.0f;
    float max_intensity = 200.0f, correct_ratio = 0.5f, correct_radius = 0.5f;
    int search_num = 10, iteration = 3, toBeClass = 1, search_c = 2;
    float intensity_diff = 50, intensity_cov = 20, occupancy = 0.6f;
    int building = 0, tree = 1, car = 2;
    std::vector<float> tr_v;
    float tr[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};  // velodyne -> camera extrinsic (utility.h:244,316)

    // include/utility.h:488-505 (float arithmetic; the unqualified sqrt / atan2 on floats are the float overloads)
    static void rotationMatrixToEulerAngles(const float R[3][3], float rpy[3]) {
        const float sy = std::sqrt(R[0][0] * R[0][0] + R[1][0] * R[1][0]);
        const bool singular = sy < 1e-6;
        if (!singular) {
            rpy[0] = std::atan2(R[2][1], R[2][2]);
            rpy[1] = std::atan2(-R[2][0], sy);
            rpy[2] = std::atan2(R[1][0], R[0][0]);
        } else {
            rpy[0] = std::atan2(-R[1][2], R[1][1]);
            rpy[1] = std::atan2(-R[2][0], sy);
            rpy[2] = 0;
        }
    }
    // A^-1 for the 4x4 of `tr.inverse()` (ssc.cpp:967): adjugate over determinant in float.  Eigen 3.3 picks an SSE
    // kernel for Matrix4f on x86; its rounding differs in the last bits (parity unpinned, tolerance 1e-5 relative).
    static bool inverse4(const float m[4][4], float inv[4][4]) {
        const float* a = &m[0][0];
        float o[16];
        o[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
        o[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
        o[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
        o[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
        o[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
        o[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
        o[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
        o[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
        o[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
        o[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
        o[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
        o[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
        o[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
        o[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
        o[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
        o[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
        const float det = a[0] * o[0] + a[1] * o[4] + a[2] * o[8] + a[3] * o[12];
        if (det == 0.f) return false;
        const float id = 1.0f / det;
        for (int i = 0; i < 16; ++i) (&inv[0][0])[i] = o[i] * id;
        return true;
    }
    static void mul4(const float a[4][4], const float b[4][4], float c[4][4]) {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float acc = a[i][0] * b[0][j];
                for (int k = 1; k < 4; ++k) acc += a[i][k] * b[k][j];
                c[i][j] = acc;
            }
    }
    // one line of a KITTI poses.txt (3x4 camera pose, row-major) -> the velodyne-frame Pose of ssc.cpp:962-989:
    // velo_to_cam = tr^-1 * cam * tr, translation from its last column, roll / pitch / yaw from its rotation
    bool kittiPose(const float pose_v[12], Pose& pose, float velo_to_cam[4][4]) const {
        float cam[4][4] = {{pose_v[0], pose_v[1], pose_v[2], pose_v[3]},
                           {pose_v[4], pose_v[5], pose_v[6], pose_v[7]},
                           {pose_v[8], pose_v[9], pose_v[10], pose_v[11]},
                           {0.f, 0.f, 0.f, 1.f}};
        float ti[4][4], t1[4][4];
        if (!inverse4(tr, ti)) return false;
        mul4(ti, cam, t1);
        mul4(t1, tr, velo_to_cam);
        pose.x = velo_to_cam[0][3];
        pose.y = velo_to_cam[1][3];
        pose.z = velo_to_cam[2][3];
        float R[3][3], rpy[3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[i][j] = velo_to_cam[i][j];
        rotationMatrixToEulerAngles(R, rpy);
        pose.roll = rpy[0];
        pose.pitch = rpy[1];
        pose.yaw = rpy[2];
        return true;
    }

    virtual ~Utility() {}
    Utility() {}
    // The keys and defaults of Utility::Utility() (utility.h:260-326)
    bool loadYaml(const std::string& path) {
        scvod_host::YamlLite y;
        if (!y.load(path)) return false;
        y.param<std::string>("common/out_path_", out_path, " ");
        y.param<int>("common/kNumOmpCores_", kNumOmpCores, 6);
        y.param<bool>("common/save_", save, true);
        y.param<bool>("common/mapping_init_", mapping_init, false);
        y.param<bool>("common/is_pcd_", is_pcd, false);
        y.param<int>("common/skip_", skip, 2);
        y.param<std::string>("session/data_path_", data_path, " ");
        y.param<std::string>("session/label_path_", label_path, " ");
        y.param<std::string>("session/pose_path_", pose_path, " ");
        y.param<int>("session/init_", init, 5);
        y.param<int>("session/start_", start, 5);
        y.param<int>("session/end_", end, 50);
        y.param<float>("ssc/sensor_height_", sensor_height, 2.0f);
        y.param<float>("ssc/min_dis_", min_dis, 0.0f);
        y.param<float>("ssc/max_dis_", max_dis, 50.0f);
        y.param<float>("ssc/min_angle_", min_angle, 0.0f);
        y.param<float>("ssc/max_angle_", max_angle, 360.0f);
        y.param<float>("ssc/min_azimuth_", min_azimuth, -30.0f);
        y.param<float>("ssc/max_azimuth_", max_azimuth, 60.0f);
        y.param<float>("ssc/range_res_", range_res, 0.2f);
        y.param<float>("ssc/sector_res_", sector_res, 1.2f);
        y.param<float>("ssc/azimuth_res_", azimuth_res, 2.0f);
        y.param<float>("ssc/refine_height_", refine_height, -1.0f);
        y.param<float>("ssc/max_z_", max_z, 1.0f);
        y.param<float>("ssc/min_z_", min_z, -1.0f);
        y.param<float>("ssc/car_angle_", car_angle, 120.0f);
        y.param<float>("ssc/car_height_", car_height, 2.0f);
        y.param<float>("ssc/car_square_", car_square, 2.0f);
        y.param<float>("ssc/max_intensity_", max_intensity, 200.0f);
        y.param<float>("ssc/correct_radius_", correct_radius, 0.5f);
        y.param<float>("ssc/correct_ratio_", correct_ratio, 0.5f);
        y.param<int>("ssc/search_num_", search_num, 10);
        y.param<int>("ssc/iteration_", iteration, 3);
        y.param<int>("ssc/toBeClass_", toBeClass, 1);
        y.param<int>("ssc/search_c_", search_c, 2);
        y.param<float>("ssc/intensity_diff_", intensity_diff, 50.f);
        y.param<float>("ssc/intensity_cov_", intensity_cov, 20.f);
        y.param<float>("ssc/occupancy_", occupancy, 0.6f);
        y.param<int>("ssc/building_", building, 0);
        y.param<int>("ssc/tree_", tree, 1);
        y.param<int>("ssc/car_", car, 2);
        tr_v = y.floats("ssc/tr_");
        if (tr_v.size() == 16)  // Eigen::Map<RowMajor 4x4> (utility.h:316)
            for (int i = 0; i < 16; ++i) tr[i / 4][i % 4] = tr_v[i];
        return true;
    }
    scvod_params toScvodParams() const {
        scvod_params p;
        scvod_params_default(&p);
        p.sensor_height = sensor_height;
        p.min_dis = min_dis;
        p.max_dis = max_dis;
        p.min_angle = min_angle;
        p.max_angle = max_angle;
        p.min_azimuth = min_azimuth;
        p.max_azimuth = max_azimuth;
        p.range_res = range_res;
        p.sector_res = sector_res;
        p.azimuth_res = azimuth_res;
        p.occupancy = occupancy;
        p.max_z = max_z;
        p.min_z = min_z;
        p.car_square = car_square;
        p.toBeClass = toBeClass;
        return p;
    }
};
#endif
