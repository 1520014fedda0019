// ssc.h -- `class SSC : public Utility` with the reference's member names and the signatures of the
// hot-path methods (/root/reference/include/ssc.h:7-105).  Only the methods SURVEY.md section 8 puts
// on the hot path are implemented here (each is a thin call through include/scvod.h); segment(),
// recognize(), I/O and the writers stay the reference's own host code (INTEGRATION.md shows the patch).
#ifndef SCVOD_HOST_SSC_H_
#define SCVOD_HOST_SSC_H_
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "patchwork.h"
#include "utility.h"

class SSC : public Utility {
  public:
    static int id;

    int range_num = 0, sector_num = 0, azimuth_num = 0, bin_num = 0;  // ssc.h:11-14

    std::vector<pcl::PointCloud<pcl::PointXYZI>::Ptr> g_cloud_vec;     // ssc.h:26
    std::vector<PointAPRI> apri_vec;                                   // ssc.h:28
    std::unordered_map<int, Voxel> hash_cloud;                         // ssc.h:29
    Frame frame_ssc;                                                   // ssc.h:30
    std::shared_ptr<PatchWork<pcl::PointXYZI>> PatchworkGroundSeg;     // ssc.h:32
    pcl::PointCloud<pcl::PointXYZI>::Ptr cloud_use;                    // ssc.h:33
    pcl::PointCloud<pcl::PointXYZI>::Ptr cloud_eva_static;             // ssc.h:45
    int name = 0;                                                      // ssc.h:50
    std::vector<Frame> frame_set;                                      // ssc.h:51
    std::vector<pcl::PointCloud<pcl::PointXYZI>::Ptr> cloud_vec;       // ssc.h:20 (what getCloud loads)
    std::vector<Pose> pose_vec;                                        // ssc.h:21
    std::vector<std::vector<float>> trans_vec;                         // ssc.h:22: velo_to_cam per loaded pose, row-major 4x4

    ~SSC();
    // The reference's SSC() reads the ROS parameter server; here the YAML file is named explicitly.
    // max_points: capacity of the device arena (points per scan).
    explicit SSC(const std::string& yaml_path, int device = 0, int max_points = 400000);

    void allocateMemory();  // ssc.cpp:67-77
    void reset();           // ssc.cpp:79-86

    // hot path (ssc.h:63-68, 91)
    void process(const pcl::PointCloud<pcl::PointXYZI>::Ptr& cloudIn_);
    pcl::PointCloud<pcl::PointXYZI>::Ptr extractGroudByPatchWork(const pcl::PointCloud<pcl::PointXYZI>::Ptr& cloudIn_);
    void makeApriVec(const pcl::PointCloud<pcl::PointXYZI>::Ptr& cloud_);
    void makeHashCloud(const std::vector<PointAPRI>& apriIn_);
    void tracking(Frame& frame_pre_, Frame& frame_next_, Pose pose_pre_, Pose pose_next_);
    // sequence driver and its loaders (ssc.h:93-97; KITTI branch of ssc.cpp:913-992, 1021-1125, 1428-1452).  The .pcd
    // branch (is_pcd_) needs PCL's reader and is not provided; the evaluation copies (rgb / ori clouds) are not produced.
    void getPose();
    void getCloud();
    void segDF();
    // GPU stand-in for segment() + recognize() when the reference's PCL host code is not linked: curved-voxel
    // clustering (ssc.cpp:299-393) + bounding-box refine / recognise rules (ssc.cpp:437-467, 849-872);
    // no intensity merge, no region growing (building and tree both become `tree`).
    void segmentGpu();
    // The per-scan body of getCloud (ssc.cpp:1063-1106) without the file I/O: label filter (label & 0xFFFF in {0, 1}
    // skipped), intensity * max_intensity, pcl::VoxelGrid 0.08 m -- the cloud that cloud_vec receives.
    pcl::PointCloud<pcl::PointXYZI>::Ptr filterAndDownsample(const std::vector<float>& values_cloud,
                                                             const std::vector<uint32_t>& values_label);

    scvod_ctx* ctx() const { return ctx_; }
    int dynamic_num_last = 0;  // what the reference only logs (ssc.cpp:1424)

  private:
    void fillHashCloud(const scvod_scan_result& r, const PointAPRI* apri);
    static void cloudToXyzi(const pcl::PointCloud<pcl::PointXYZI>& c, std::vector<float>& out);
    scvod_ctx* ctx_ = nullptr;
    std::vector<float> stage_;
};
#endif
