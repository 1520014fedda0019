// sequence_main.cpp -- SSC::segDF-shaped driver (src/ssc.cpp:1428-1452) on the facade: for every scan
// process() -> segmentGpu() -> keep the Frame; then tracking(frame[i], frame[i+1]) along the chain.
// Writes, per scan, the points of the clusters that ended up dynamic (state == 1).
//   usage: scvod_sequence <config.yaml> <dir with 0.f32 1.f32 ... and poses.txt> <n_scans> <out_dir>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "ssc.h"

static pcl::PointCloud<pcl::PointXYZI>::Ptr load(const std::string& path) {
    std::ifstream in(path, std::ios::binary);
    in.seekg(0, std::ios::end);
    size_t n = (size_t)in.tellg() / 16;
    in.seekg(0);
    std::vector<float> v(n * 4);
    in.read((char*)v.data(), n * 16);
    pcl::PointCloud<pcl::PointXYZI>::Ptr c(new pcl::PointCloud<pcl::PointXYZI>());
    c->points.resize(n);
    for (size_t i = 0; i < n; ++i) {
        c->points[i].x = v[4 * i];
        c->points[i].y = v[4 * i + 1];
        c->points[i].z = v[4 * i + 2];
        c->points[i].intensity = v[4 * i + 3];
    }
    return c;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        std::cerr << "usage: scvod_sequence <config.yaml> <dir> <n_scans> <out_dir>\n";
        return 2;
    }
    try {
        const std::string dir = argv[2], out = argv[4];
        const int n_scans = std::atoi(argv[3]);
        SSC ssc(argv[1]);
        std::vector<Pose> pose_vec(n_scans);
        {
            std::ifstream pf(dir + "/poses.txt");
            for (int i = 0; i < n_scans; ++i) pf >> pose_vec[i].x >> pose_vec[i].y >> pose_vec[i].z >> pose_vec[i].roll >> pose_vec[i].pitch >> pose_vec[i].yaw;
        }
        for (int i = 0; i < n_scans; ++i) {  // hot loop #1 (ssc.cpp:1435-1445)
            auto cloud = load(dir + "/" + std::to_string(i) + ".f32");
            ssc.process(cloud);
            ssc.segmentGpu();
            ssc.frame_set.emplace_back(ssc.frame_ssc);
            ssc.reset();
            SSC::id += ssc.skip;
        }
        int dyn_total = 0;
        for (int i = 0; i + 1 < n_scans; ++i) {  // hot loop #2, a sequential chain (ssc.cpp:1450-1452)
            ssc.tracking(ssc.frame_set[i], ssc.frame_set[i + 1], pose_vec[i], pose_vec[i + 1]);
            dyn_total += ssc.dynamic_num_last;
        }
        for (int i = 0; i < n_scans; ++i) {
            std::ofstream o(out + "/" + std::to_string(i) + "_dynamic.f32", std::ios::binary);
            int cars = 0, dyn = 0;
            for (auto& kv : ssc.frame_set[i].cluster_set) {
                const Cluster& c = kv.second;
                cars += (c.state != -1);
                if (c.state == 1) {
                    ++dyn;
                    for (int p : c.occupy_pts) o.write((const char*)&ssc.frame_set[i].cloud_use->points[p], 16);
                }
            }
            std::cout << "scan " << i << " clusters " << ssc.frame_set[i].cluster_set.size() << " tracked " << cars << " dynamic " << dyn << "\n";
        }
        std::cout << "dynamic_total " << dyn_total << "\n";
    } catch (const std::exception& e) {
        std::cerr << "scvod_sequence failed: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
