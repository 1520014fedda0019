// utility.h -- host-side mirror of the reference's `class Utility` configuration surface
// (/root/reference/include/utility.h:187-327): same member names, same two-level keys with trailing
// underscore, same nh.param<> defaults; values come from the YAML file directly (yaml_lite.h) instead
// of the ROS parameter server, so the same config/*.yaml files work with or without ROS.
#ifndef SCVOD_HOST_UTILITY_H_
#define SCVOD_HOST_UTILITY_H_
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <locale>
#include <sstream>
#include <stdexcept>
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
    // max_name is the LAST USED cluster name when clusterAndCreateFrame ends (ssc.cpp:354 `max_name = cluster_name ++`), and
    // SSC::tracking hands it out again (ssc.cpp:1357, 1401).  The facade's names are canonical (smallest point + 5), not the
    // reference's running numbers, so the numbers after that first one start at name_floor, above every name of the frame.
    int name_floor = 0;
    int takeName() {  // `frame_next_.max_name ++`
        const int nm = max_name;
        max_name = (max_name + 1 > name_floor) ? max_name + 1 : name_floor;
        return nm;
    }
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

    // Utility::saveCloud / loadCloud (utility.h:408-428): <path_><id><name_> as an ASCII .pcd the way pcl::io::savePCDFile writes
    // it (PCDWriter::writeASCII of PCL 1.8: the eleven header lines, eight significant digits in the classic locale, "nan"
    // for NaN); the reader takes ASCII and binary files with x y z and, when present, intensity fields.
    static std::string pcdFloat(float v) {
        if (v != v) return "nan";
        std::ostringstream o;
        o.imbue(std::locale::classic());
        o.precision(8);
        o << v;
        return o.str();
    }
    static bool writePcdAscii(const std::string& file, const pcl::PointCloud<pcl::PointXYZI>& cloud) {
        std::ofstream f(file.c_str());
        if (!f) return false;
        const size_t n = cloud.points.size();
        f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
          << "WIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA ascii\n";
        for (const auto& p : cloud.points) f << pcdFloat(p.x) << ' ' << pcdFloat(p.y) << ' ' << pcdFloat(p.z) << ' ' << pcdFloat(p.intensity) << '\n';
        return (bool)f;
    }
    static bool readPcd(const std::string& file, pcl::PointCloud<pcl::PointXYZI>& cloud) {
        std::ifstream f(file.c_str(), std::ios::binary);
        if (!f) return false;
        std::vector<std::string> fields;
        std::vector<int> sizes, counts;
        std::vector<char> types;
        size_t points = 0;
        std::string line, data;
        while (std::getline(f, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty() || line[0] == '#') continue;
            std::istringstream ls(line);
            std::string key;
            ls >> key;
            if (key == "FIELDS") {
                for (std::string t; ls >> t;) fields.push_back(t);
            } else if (key == "SIZE") {
                for (int t; ls >> t;) sizes.push_back(t);
            } else if (key == "TYPE") {
                for (char t; ls >> t;) types.push_back(t);
            } else if (key == "COUNT") {
                for (int t; ls >> t;) counts.push_back(t);
            } else if (key == "POINTS") {
                ls >> points;
            } else if (key == "DATA") {
                ls >> data;
                break;
            }
        }
        const size_t nf = fields.size();
        if (!nf || sizes.size() != nf || types.size() != nf || (data != "ascii" && data != "binary")) return false;
        if (counts.size() != nf) counts.assign(nf, 1);
        int at[4] = {-1, -1, -1, -1};  // x y z intensity -> field
        for (size_t k = 0; k < nf; ++k) {
            if (fields[k] == "x") at[0] = (int)k;
            if (fields[k] == "y") at[1] = (int)k;
            if (fields[k] == "z") at[2] = (int)k;
            if (fields[k] == "intensity") at[3] = (int)k;
        }
        if (at[0] < 0 || at[1] < 0 || at[2] < 0) return false;
        for (int k = 0; k < 4; ++k)
            if (at[k] >= 0 && (types[at[k]] != 'F' || sizes[at[k]] != 4 || counts[at[k]] != 1)) return false;
        cloud.points.assign(points, pcl::PointXYZI());
        if (data == "ascii") {
            std::vector<size_t> first(nf);  // index of a field's first token on a line
            size_t tokens = 0;
            for (size_t k = 0; k < nf; ++k) {
                first[k] = tokens;
                tokens += (size_t)counts[k];
            }
            std::vector<std::string> tok(tokens);
            for (size_t i = 0; i < points; ++i) {
                if (!std::getline(f, line)) return false;
                std::istringstream ls(line);
                for (size_t t = 0; t < tokens; ++t)
                    if (!(ls >> tok[t])) return false;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < 4; ++k)
                    if (at[k] >= 0) v[k] = std::strtof(tok[first[at[k]]].c_str(), nullptr);
                cloud.points[i].x = v[0];
                cloud.points[i].y = v[1];
                cloud.points[i].z = v[2];
                cloud.points[i].intensity = v[3];
            }
        } else {
            std::vector<size_t> off(nf);
            size_t stride = 0;
            for (size_t k = 0; k < nf; ++k) {
                off[k] = stride;
                stride += (size_t)sizes[k] * (size_t)counts[k];
            }
            std::vector<char> rec(stride);
            for (size_t i = 0; i < points; ++i) {
                if (!f.read(rec.data(), (std::streamsize)stride)) return false;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < 4; ++k)
                    if (at[k] >= 0) std::memcpy(&v[k], rec.data() + off[at[k]], 4);
                cloud.points[i].x = v[0];
                cloud.points[i].y = v[1];
                cloud.points[i].z = v[2];
                cloud.points[i].intensity = v[3];
            }
        }
        cloud.width = (unsigned)points;
        cloud.height = 1;
        return true;
    }
    template <typename CloudT>
    void saveCloud(const CloudT& cloud_, const std::string& path_, const int& id, const std::string& name_) {  // root path
        const std::string save_path = path_ + std::to_string(id) + name_;
        cloud_->height = 1;
        cloud_->width = (unsigned)cloud_->points.size();
        if (save) {
            if (cloud_->points.size() == 0 || !writePcdAscii(save_path, *cloud_)) std::fprintf(stderr, "%s save error\n", (std::to_string(id) + name_).c_str());
        }
    }
    template <typename CloudT>
    void loadCloud(CloudT& cloud_, const std::string& path_) {
        if (!readPcd(path_, *cloud_)) throw std::runtime_error("pose file " + path_ + " load error");
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
