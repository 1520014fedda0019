// ssc.cpp -- hot-path methods of SSC on top of the C-ABI (include/scvod.h).  The arithmetic lives in
// the HIP kernels; this file converts between the reference's containers (pcl::PointCloud,
// std::vector<PointAPRI>, unordered_map<int, Voxel>) and the POD arrays of the boundary, and keeps the
// sequential label bookkeeping of SSC::tracking (src/ssc.cpp:1323-1421) on the host.
#include <cstdio>
#include "ssc.h"

#include <dirent.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

int SSC::id = 0;

namespace {
void chk(scvod_ctx* c, int rc, const char* what) {
    if (rc != SCVOD_OK) throw std::runtime_error(std::string(what) + ": " + (c ? scvod_last_error(c) : "no ctx"));
}
// Utility::deg2rad (utility.h:351-354)
inline float deg2rad_f(float degrees) { return (float)((double)degrees * M_PI / 180.0); }
}  // namespace

SSC::~SSC() { scvod_destroy(ctx_); }

SSC::SSC(const std::string& yaml_path, int device, int max_points) {
    if (!loadYaml(yaml_path)) throw std::invalid_argument("cannot read " + yaml_path);
    allocateMemory();
    scvod_params p = toScvodParams();
    scvod_grid_dims(&p, &range_num, &sector_num, &azimuth_num, &bin_num);  // ssc.cpp:36-39
    int rc = scvod_create(&p, nullptr, device, max_points, 1, &ctx_);
    if (rc != SCVOD_OK) throw std::runtime_error("scvod_create failed (status " + std::to_string(rc) + "): the SCV-OD hot path is GPU-only");
    PatchworkGroundSeg->attach(ctx_);
}

void SSC::allocateMemory() {
    PatchworkGroundSeg.reset(new PatchWork<pcl::PointXYZI>());
    cloud_use.reset(new pcl::PointCloud<pcl::PointXYZI>());
    cloud_eva_static.reset(new pcl::PointCloud<pcl::PointXYZI>());
}

void SSC::reset() {
    frame_ssc = Frame();
    apri_vec.clear();
    hash_cloud.clear();
    cloud_use->clear();
}

void SSC::cloudToXyzi(const pcl::PointCloud<pcl::PointXYZI>& c, std::vector<float>& out) {
    out.resize(c.points.size() * 4);
    for (size_t i = 0; i < c.points.size(); ++i) {
        out[4 * i] = c.points[i].x;
        out[4 * i + 1] = c.points[i].y;
        out[4 * i + 2] = c.points[i].z;
        out[4 * i + 3] = c.points[i].intensity;
    }
}

pcl::PointCloud<pcl::PointXYZI>::Ptr SSC::extractGroudByPatchWork(const pcl::PointCloud<pcl::PointXYZI>::Ptr& cloudIn_) {
    double time_pw;
    pcl::PointCloud<pcl::PointXYZI>::Ptr g_cloud(new pcl::PointCloud<pcl::PointXYZI>());
    pcl::PointCloud<pcl::PointXYZI>::Ptr ng_cloud(new pcl::PointCloud<pcl::PointXYZI>());
    g_cloud_vec.emplace_back(g_cloud);
    PatchworkGroundSeg->set_sensor(sensor_height);
    PatchworkGroundSeg->estimate_ground(*cloudIn_, *g_cloud, *ng_cloud, time_pw);
    return ng_cloud;
}

void SSC::makeApriVec(const pcl::PointCloud<pcl::PointXYZI>::Ptr& cloud_) {
    cloudToXyzi(*cloud_, stage_);
    scvod_scan_result r;
    chk(ctx_, scvod_bin_scan(ctx_, stage_.data(), (int)cloud_->points.size(), 1, 0, &r), "scvod_bin_scan");
    for (int k = 0; k < r.n_rejected; ++k) cloud_eva_static->points.emplace_back(cloud_->points[r.rejected_src[k]]);
    for (int k = 0; k < r.n_apri; ++k) {
        cloud_use->points.push_back(cloud_->points[r.apri_src[k]]);
        frame_ssc.cloud_use->points.push_back(cloud_->points[r.apri_src[k]]);
    }
    apri_vec.insert(apri_vec.end(), r.apri, r.apri + r.n_apri);
}

// Voxel records from the CSR arrays; the "center" (consumed only by dead code in the reference) is
// evaluated on the host with libm exactly as ssc.cpp:271-277 does.
void SSC::fillHashCloud(const scvod_scan_result& r, const PointAPRI* apri) {
    hash_cloud.reserve(r.n_voxels);
    for (int v = 0; v < r.n_voxels; ++v) {
        Voxel voxel;
        const int b = r.vox_pt_begin[v], e = r.vox_pt_begin[v + 1];
        voxel.ptIdx.assign(r.vox_pts + b, r.vox_pts + e);
        voxel.intensity_record.reserve(e - b);
        for (int k = b; k < e; ++k) voxel.intensity_record.push_back(apri[r.vox_pts[k]].intensity);
        const PointAPRI& first = apri[r.vox_pts[b]];
        voxel.range_idx = first.range_idx;
        voxel.sector_idx = first.sector_idx;
        voxel.azimuth_idx = first.azimuth_idx;
        const float range_center = (first.range_idx * 2 + 1) / 2 * range_res + min_dis;
        const float sector_center = deg2rad_f((first.sector_idx * 2 + 1) / 2 * sector_res) + min_angle;
        const float azimuth_center = deg2rad_f((first.azimuth_idx * 2 + 1) / 2 * azimuth_res) + deg2rad_f(min_azimuth);
        voxel.center.x = range_center * std::cos(sector_center);
        voxel.center.y = range_center * std::sin(sector_center);
        voxel.center.z = range_center * std::tan(azimuth_center);
        voxel.center.intensity = first.voxel_idx;
        voxel.intensity_av = r.vox_av[v];
        voxel.intensity_cov = r.vox_cov[v];
        hash_cloud.insert(std::make_pair(r.vox_key[v], voxel));
    }
}

void SSC::makeHashCloud(const std::vector<PointAPRI>& apriIn_) {
    scvod_scan_result r;
    chk(ctx_, scvod_voxelize(ctx_, apriIn_.data(), (int)apriIn_.size(), &r), "scvod_voxelize");
    fillHashCloud(r, apriIn_.data());
}

// getCloud's per-scan filter + downsample (ssc.cpp:1063-1076, 1103-1106) as one call; rgb_cloud / ori_cloud (plots and
// evaluation copies) are not produced
pcl::PointCloud<pcl::PointXYZI>::Ptr SSC::filterAndDownsample(const std::vector<float>& values_cloud,
                                                              const std::vector<uint32_t>& values_label) {
    const int num_points = (int)values_label.size();
    if (values_cloud.size() < (size_t)4 * num_points) throw std::runtime_error("filterAndDownsample: cloud shorter than labels");
    const float leaf[3] = {0.08f, 0.08f, 0.08f};  // sample.setLeafSize(0.08, 0.08, 0.08)
    std::vector<float> out((size_t)4 * (num_points > 0 ? num_points : 1));
    int n_out = 0;
    chk(ctx_, scvod_voxelgrid(ctx_, values_cloud.data(), values_label.data(), num_points, leaf, max_intensity, out.data(), num_points,
                              &n_out), "scvod_voxelgrid");
    pcl::PointCloud<pcl::PointXYZI>::Ptr raw_cloud(new pcl::PointCloud<pcl::PointXYZI>());
    raw_cloud->points.resize(n_out);
    for (int k = 0; k < n_out; ++k) {
        raw_cloud->points[k].x = out[4 * k];
        raw_cloud->points[k].y = out[4 * k + 1];
        raw_cloud->points[k].z = out[4 * k + 2];
        raw_cloud->points[k].intensity = out[4 * k + 3];
    }
    return raw_cloud;
}

// SSC::process up to makeHashCloud as ONE trip to the GPU (ssc.cpp:224-241); the debug dumps of the
// reference (intensityVisualization, recordIntensity) are not part of the hot path.
void SSC::process(const pcl::PointCloud<pcl::PointXYZI>::Ptr& cloudIn_) {
    frame_ssc.id = id;
    cloudToXyzi(*cloudIn_, stage_);
    scvod_scan_result r;
    chk(ctx_, scvod_process_scan(ctx_, stage_.data(), (int)cloudIn_->points.size(), &r), "scvod_process_scan");
    pcl::PointCloud<pcl::PointXYZI>::Ptr g_cloud(new pcl::PointCloud<pcl::PointXYZI>());
    g_cloud->points.reserve(r.n_ground);
    for (int k = 0; k < r.n_ground; ++k) g_cloud->points.push_back(cloudIn_->points[r.ground_idx[k]]);
    g_cloud_vec.emplace_back(g_cloud);
    for (int k = 0; k < r.n_rejected; ++k) cloud_eva_static->points.emplace_back(cloudIn_->points[r.rejected_src[k]]);
    cloud_use->points.reserve(r.n_apri);
    for (int k = 0; k < r.n_apri; ++k) {
        cloud_use->points.push_back(cloudIn_->points[r.apri_src[k]]);
        frame_ssc.cloud_use->points.push_back(cloudIn_->points[r.apri_src[k]]);
    }
    apri_vec.assign(r.apri, r.apri + r.n_apri);
    fillHashCloud(r, apri_vec.data());
}

void SSC::segmentGpu() {
    const int n = (int)apri_vec.size();
    std::vector<int> names(n ? n : 1), types(n ? n : 1);
    chk(ctx_, scvod_cluster(ctx_, apri_vec.data(), n, names.data()), "scvod_cluster");
    chk(ctx_, scvod_batch_cluster_types(ctx_, nullptr, 1), "scvod_batch_cluster_types");
    int rc = scvod_batch_fetch_cluster_types(ctx_, 0, car, tree, types.data(), n);
    if (rc < 0) chk(ctx_, rc, "scvod_batch_fetch_cluster_types");
    frame_ssc.cluster_set.clear();
    int max_name = 4;
    for (int i = 0; i < n; ++i) {
        if (types[i] == -1) continue;  // erased by the bounding-box refine: its voxels keep label -1
        const int nm = names[i] + 5;   // the reference's names start at 5 (ssc.cpp:300,346)
        Cluster& c = frame_ssc.cluster_set[nm];
        c.name = nm;
        c.type = types[i];
        c.occupy_pts.push_back(i);
        c.occupy_voxels.push_back(apri_vec[i].voxel_idx);
        c.cloud->points.push_back(cloud_use->points[i]);
        if (nm > max_name) max_name = nm;
    }
    for (auto& kv : frame_ssc.cluster_set) {
        std::vector<int>& v = kv.second.occupy_voxels;
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        for (int key : v) hash_cloud[key].label = kv.first;
    }
    // ssc.cpp:354: max_name = the last running number handed out.  Which cluster still carries it is the device's answer
    // (scvod_batch_cluster_last_name); -1: the number was merged away, it is as good as a fresh one
    int32_t last[4] = {-1, -1, 0, 0};
    rc = scvod_batch_cluster_last_name(ctx_, last, 1, nullptr);
    if (rc < 0) chk(ctx_, rc, "scvod_batch_cluster_last_name");
    if (last[2] != 0)  // (status 1 / 2: include/scvod.h) the reference would re-use the number of a cluster this scan could not name
        std::fprintf(stderr, "[scvod] max_name of this scan is undetermined (status %d): new clusters get fresh numbers\n", (int)last[2]);
    frame_ssc.name_floor = n + 6;  // (above every canonical name + 5, erased clusters included)
    frame_ssc.max_name = last[0] >= 0 ? last[0] + 5 : frame_ssc.name_floor;
    frame_ssc.hash_cloud = hash_cloud;  // ssc.cpp:651
}

// SSC::tracking: the transform + re-bin + probe of every `car` cluster of frame_pre_ runs on the GPU in
// one call (ssc.cpp:1274-1321); label grouping and the dynamic / split / fuse decisions mutate
// frame_next_ cluster by cluster and therefore stay sequential on the host (ssc.cpp:1323-1421).
void SSC::tracking(Frame& frame_pre_, Frame& frame_next_, Pose pose_pre_, Pose pose_next_) {
    const float pp[6] = {pose_pre_.x, pose_pre_.y, pose_pre_.z, pose_pre_.roll, pose_pre_.pitch, pose_pre_.yaw};
    const float pn[6] = {pose_next_.x, pose_next_.y, pose_next_.z, pose_next_.roll, pose_next_.pitch, pose_next_.yaw};
    float T[12];
    scvod_pose_delta(pp, pn, T);

    // next frame's voxel table, sorted by key
    std::vector<int> keys;
    keys.reserve(frame_next_.hash_cloud.size());
    for (auto& kv : frame_next_.hash_cloud) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    std::vector<int> labels(keys.size());
    for (size_t v = 0; v < keys.size(); ++v) labels[v] = frame_next_.hash_cloud[keys[v]].label;

    // clusters in the container's iteration order, as the reference walks them
    std::vector<Cluster*> cars;
    std::vector<int> offs(1, 0);
    std::vector<float> pts;
    for (auto& c : frame_pre_.cluster_set) {
        if (c.second.type != car) continue;
        cars.push_back(&c.second);
        for (auto& p : c.second.cloud->points) {
            pts.push_back(p.x);
            pts.push_back(p.y);
            pts.push_back(p.z);
            pts.push_back(p.intensity);
        }
        offs.push_back((int)(pts.size() / 4));
    }
    const int n_c = (int)cars.size();
    std::vector<int> hit(std::max<size_t>(pts.size() / 4, 1)), uq(std::max<size_t>(pts.size() / 4, 1)), ub(n_c + 1);
    // labels that are -1 NOW stay -1 for the whole call (re-labelling never writes -1), so the
    // found-and-labelled test can be taken in bulk; label VALUES are read below, at decision time.
    chk(ctx_, scvod_track_probe(ctx_, pts.data(), offs.data(), n_c, T, keys.data(), labels.data(), (int)keys.size(), hit.data(), uq.data(), ub.data()),
        "scvod_track_probe");

    int dynamic_num = 0;
    for (int ci = 0; ci < n_c; ++ci) {
        Cluster& c = *cars[ci];
        if (c.track_id == -1) c.track_id = name++;
        // group this cluster's unique hit voxels by the label they carry at this moment
        std::unordered_map<int, std::vector<int>> remap_name;
        for (int k = ub[ci]; k < ub[ci + 1]; ++k) {
            const int key = keys[uq[k]];
            remap_name[frame_next_.hash_cloud[key].label].push_back(key);  // keys ascending: already sampleVec'ed
        }
        if (remap_name.empty()) {
            c.state = 1;
            ++dynamic_num;
        } else if (remap_name.size() == 1) {
            auto it = remap_name.begin();
            Cluster& nc = frame_next_.cluster_set[it->first];
            const float ratio = (float)it->second.size() / (float)nc.occupy_voxels.size();
            if (ratio < occupancy) {
                if (nc.type == car) {
                    c.state = 1;
                    ++dynamic_num;
                } else {  // split the overlapped voxels off into a new cluster of the next frame
                    c.state = 0;
                    c.type = nc.type;
                    Cluster cluster_new;
                    cluster_new.track_id = c.track_id;
                    cluster_new.name = frame_next_.takeName();  // frame_next_.max_name ++ (ssc.cpp:1357, 1401)
                    cluster_new.type = nc.type;
                    std::copy(c.color, c.color + 3, cluster_new.color);
                    cluster_new.occupy_voxels = it->second;
                    for (int v : cluster_new.occupy_voxels)
                        nc.occupy_voxels.erase(std::remove(nc.occupy_voxels.begin(), nc.occupy_voxels.end(), v), nc.occupy_voxels.end());
                    for (int v : it->second) {
                        Voxel& vx = frame_next_.hash_cloud[v];
                        vx.label = cluster_new.name;
                        cluster_new.occupy_pts.insert(cluster_new.occupy_pts.end(), vx.ptIdx.begin(), vx.ptIdx.end());
                    }
                    for (int p : cluster_new.occupy_pts) cluster_new.cloud->points.push_back(frame_next_.cloud_use->points[p]);
                    for (int p : cluster_new.occupy_pts)
                        nc.occupy_pts.erase(std::remove(nc.occupy_pts.begin(), nc.occupy_pts.end(), p), nc.occupy_pts.end());
                    frame_next_.cluster_set.insert(std::make_pair(cluster_new.name, cluster_new));
                }
            } else if (nc.type == car) {  // confirmed static: hand the track over
                c.state = 0;
                nc.track_id = c.track_id;
                for (int k = offs[ci]; k < offs[ci + 1]; ++k) {  // *cloud += transformed cluster
                    pcl::PointXYZI q;
                    const float* in = &pts[4 * (size_t)k];
                    q.x = T[0] * in[0] + T[1] * in[1] + T[2] * in[2] + T[3];
                    q.y = T[4] * in[0] + T[5] * in[1] + T[6] * in[2] + T[7];
                    q.z = T[8] * in[0] + T[9] * in[1] + T[10] * in[2] + T[11];
                    q.intensity = in[3];
                    nc.cloud->points.push_back(q);
                }
                std::copy(c.color, c.color + 3, nc.color);
            }
        } else {  // several next-frame clusters overlap: fuse the `car` ones that are covered enough
            c.state = 0;
            Cluster cluster_new;
            cluster_new.track_id = c.track_id;
            cluster_new.name = frame_next_.takeName();  // frame_next_.max_name ++ (ssc.cpp:1357, 1401)
            cluster_new.type = car;
            std::copy(c.color, c.color + 3, cluster_new.color);
            for (auto& re : remap_name) {
                Cluster& nc = frame_next_.cluster_set[re.first];
                if (nc.type == car && ((float)re.second.size() / (float)nc.occupy_voxels.size()) >= occupancy) {
                    cluster_new.occupy_pts.insert(cluster_new.occupy_pts.end(), nc.occupy_pts.begin(), nc.occupy_pts.end());
                    cluster_new.occupy_voxels.insert(cluster_new.occupy_voxels.end(), nc.occupy_voxels.begin(), nc.occupy_voxels.end());
                    frame_next_.cluster_set.erase(re.first);
                }
            }
            for (int p : cluster_new.occupy_pts) cluster_new.cloud->points.push_back(frame_next_.cloud_use->points[p]);
            for (int v : cluster_new.occupy_voxels) frame_next_.hash_cloud[v].label = cluster_new.name;
            frame_next_.cluster_set.insert(std::make_pair(cluster_new.name, cluster_new));
        }
    }
    dynamic_num_last = dynamic_num;
}

// ---- sequence loaders + driver (KITTI layout: velodyne/*.bin, labels/*.label, poses.txt) -------------------------------
namespace {
// fileSort (ssc.cpp:12-22): numeric order of the file stems
int stem_number(const std::string& path) {
    const size_t a = path.find_last_of('/') + 1;
    const std::string file = path.substr(a);
    return std::atoi(file.substr(0, file.rfind('.')).c_str());
}
std::vector<std::string> list_sorted(const std::string& dir) {
    std::vector<std::string> names;
    DIR* d = opendir(dir.c_str());
    if (!d) throw std::runtime_error("cannot open directory " + dir);
    while (dirent* e = readdir(d)) {
        const std::string n = e->d_name;
        if (n == "." || n == "..") continue;
        names.push_back(dir + (dir.empty() || dir.back() == '/' ? "" : "/") + n);
    }
    closedir(d);
    std::sort(names.begin(), names.end(), [](const std::string& x, const std::string& y) { return stem_number(x) < stem_number(y); });
    return names;
}
}  // namespace

// SSC::getPose, KITTI branch (ssc.cpp:930-989): line `count` of poses.txt is used when count >= start, count < end and
// (count - start) % skip == 0; 12 numbers per line, split at blanks and read with atof
void SSC::getPose() {
    if (is_pcd) throw std::runtime_error("getPose: the .pcd pose branch (is_pcd_) needs PCL and is not provided");
    std::ifstream pose_file(pose_path);
    if (!pose_file) throw std::runtime_error("cannot open " + pose_path);
    std::string line;
    int count = 0;
    while (std::getline(pose_file, line)) {
        if (count < start || (count - start) % skip != 0) {
            count++;
            continue;
        }
        if (count >= end) break;
        float pose_v[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        std::istringstream is(line);
        std::string tok;
        int l = 0;
        while (l < 12 && is >> tok) pose_v[l++] = (float)std::atof(tok.c_str());
        Pose pose;
        float v2c[4][4];
        if (!kittiPose(pose_v, pose, v2c)) throw std::runtime_error("ssc/tr_ is singular");
        trans_vec.emplace_back(&v2c[0][0], &v2c[0][0] + 16);
        count++;
        pose_vec.emplace_back(pose);
    }
}

// SSC::getCloud, KITTI branch (ssc.cpp:1021-1125): scans start, start + skip, ... < end of the sorted .bin / .label lists;
// label & 0xFFFF in {0, 1} dropped, intensity * max_intensity, VoxelGrid 0.08 m (filterAndDownsample, on the GPU).  The
// reference's range test `dis >= min_dis || dis <= max_dis` (ssc.cpp:1094) is always true and keeps every point.
void SSC::getCloud() {
    if (is_pcd) throw std::runtime_error("getCloud: the .pcd branch (is_pcd_) needs PCL and is not provided");
    const std::vector<std::string> bin_name = list_sorted(data_path), label_name = list_sorted(label_path);
    if (bin_name.size() != label_name.size()) throw std::runtime_error("bins or labels load error");
    const int all = (int)bin_name.size();
    if (start < 0 || end > all) throw std::runtime_error("the start or end index set error");
    for (int i = start; i < end; i = i + skip) {
        std::ifstream in_label(label_name[i], std::ios::binary);
        if (!in_label.is_open()) throw std::runtime_error("can't open " + label_name[i]);
        in_label.seekg(0, std::ios::end);
        const uint32_t num_points = (uint32_t)(in_label.tellg() / sizeof(uint32_t));
        in_label.seekg(0, std::ios::beg);
        std::vector<uint32_t> values_label(num_points);
        in_label.read((char*)values_label.data(), num_points * sizeof(uint32_t));
        std::ifstream in_cloud(bin_name[i], std::ios::binary);
        if (!in_cloud.is_open()) throw std::runtime_error("can't open " + bin_name[i]);
        std::vector<float> values_cloud(4 * (size_t)num_points);
        in_cloud.read((char*)values_cloud.data(), 4 * (size_t)num_points * sizeof(float));
        cloud_vec.emplace_back(filterAndDownsample(values_cloud, values_label));
    }
}

// SSC::segDF (ssc.cpp:1428-1452): load, per scan process -> segment -> recognize -> keep the frame, then the tracking
// chain.  segment() / recognize() are the GPU stand-in segmentGpu() here (curved-voxel clustering + box rules); a build
// that links the reference's PCL host code calls its own segment() / recognize() instead.
void SSC::segDF() {
    id = start;
    getPose();
    getCloud();
    for (auto& cloud : cloud_vec) {
        process(cloud);
        segmentGpu();
        frame_set.emplace_back(frame_ssc);
        reset();
        id += skip;
    }
    if (pose_vec.size() < frame_set.size())  // the reference would read past pose_vec (ssc.cpp:1450); every other loader error throws here
        throw std::runtime_error("SSC::segDF: " + std::to_string(pose_vec.size()) + " poses for " + std::to_string(frame_set.size()) + " frames");
    for (int i = 0; i + 1 < (int)frame_set.size(); i++) tracking(frame_set[i], frame_set[i + 1], pose_vec[i], pose_vec[i + 1]);
}
