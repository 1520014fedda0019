// utility.h -- host-side mirror of the reference's `class Utility` configuration surface
// (/root/reference/include/utility.h:187-327): same member names, same two-level keys with trailing
// underscore, same nh.param<> defaults; values come from the YAML file directly (yaml_lite.h) instead
// of the ROS parameter server, so the same config/*.yaml files work with or without ROS.
#ifndef SCVOD_HOST_UTILITY_H_
#define SCVOD_HOST_UTILITY_H_
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
