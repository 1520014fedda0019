// sequence_main.cpp -- the reference's node main (src/main.cpp:3-14: construct SSC, call segDF) on the facade, for a
// KITTI-layout sequence named by the YAML file's session/ keys (data_path_, label_path_, pose_path_, start_, end_) and
// common/skip_.  Writes, per loaded frame, the points of the clusters that ended up dynamic (state == 1).
//   usage: scvod_sequence <config.yaml> <out_dir> [--data DIR] [--labels DIR] [--poses FILE] [--start S] [--end E] [--skip K]
//          scvod_sequence --poses-only <config.yaml> [--poses FILE] [--start S] [--end E] [--skip K]   (no GPU needed)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "ssc.h"

struct PoseOnly : public Utility {  // getPose without a device: the pose maths live in Utility
    std::vector<Pose> pose_vec;
    void run() {
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
            count++;
            pose_vec.emplace_back(pose);
        }
    }
};

static void overrides(Utility& u, int argc, char** argv, int from) {
    for (int a = from; a + 1 < argc; a += 2) {
        const std::string k = argv[a], v = argv[a + 1];
        if (k == "--data") u.data_path = v;
        else if (k == "--labels") u.label_path = v;
        else if (k == "--poses") u.pose_path = v;
        else if (k == "--start") u.start = std::atoi(v.c_str());
        else if (k == "--end") u.end = std::atoi(v.c_str());
        else if (k == "--skip") u.skip = std::atoi(v.c_str());
        else throw std::invalid_argument("unknown option " + k);
    }
}

int main(int argc, char** argv) {
    try {
        if (argc >= 3 && std::strcmp(argv[1], "--poses-only") == 0) {
            PoseOnly p;
            if (!p.loadYaml(argv[2])) throw std::invalid_argument(std::string("cannot read ") + argv[2]);
            overrides(p, argc, argv, 3);
            p.run();
            std::cout.precision(9);
            for (auto& q : p.pose_vec) std::cout << q.x << " " << q.y << " " << q.z << " " << q.roll << " " << q.pitch << " " << q.yaw << "\n";
            return 0;
        }
        if (argc >= 6 && std::strcmp(argv[1], "--pcd-copy") == 0) {  // loadCloud + saveCloud, no device: <in.pcd> <out_root> <id> <name>
            Utility u;
            pcl::PointCloud<pcl::PointXYZI>::Ptr c(new pcl::PointCloud<pcl::PointXYZI>());
            u.loadCloud(c, argv[2]);
            u.saveCloud(c, argv[3], std::atoi(argv[4]), argv[5]);
            std::cout << "points " << c->points.size() << "\n";
            return 0;
        }
        if (argc < 3) {
            std::cerr << "usage: scvod_sequence <config.yaml> <out_dir> [--data DIR] [--labels DIR] [--poses FILE] [--start S] [--end E] [--skip K]\n";
            return 2;
        }
        const std::string out = argv[2];
        SSC ssc(argv[1]);
        overrides(ssc, argc, argv, 3);
        ssc.segDF();
        int dyn_total = 0;
        const int n = (int)ssc.frame_set.size();
        for (int i = 0; i < n; ++i) {
            {  // the frame as the path saw it (after the loader's filter + VoxelGrid)
                std::ofstream oc(out + "/" + std::to_string(ssc.frame_set[i].id) + "_cloud.f32", std::ios::binary);
                for (auto& p : ssc.cloud_vec[i]->points) oc.write((const char*)&p, 16);
            }
            std::ofstream o(out + "/" + std::to_string(ssc.frame_set[i].id) + "_dynamic.f32", std::ios::binary);
            int cars = 0, dyn = 0;
            for (auto& kv : ssc.frame_set[i].cluster_set) {
                const Cluster& c = kv.second;
                cars += (c.state != -1);
                if (c.state == 1) {
                    ++dyn;
                    for (int p : c.occupy_pts) o.write((const char*)&ssc.frame_set[i].cloud_use->points[p], 16);
                }
            }
            if (ssc.save && dyn) {  // the removed points of the frame as a .pcd (Utility::saveCloud, utility.h:408-419)
                pcl::PointCloud<pcl::PointXYZI>::Ptr removed(new pcl::PointCloud<pcl::PointXYZI>());
                for (auto& kv : ssc.frame_set[i].cluster_set)
                    if (kv.second.state == 1)
                        for (int p : kv.second.occupy_pts) removed->push_back(ssc.frame_set[i].cloud_use->points[p]);
                ssc.saveCloud(removed, out + "/", ssc.frame_set[i].id, "_dynamic.pcd");
            }
            dyn_total += dyn;
            std::cout << "frame " << ssc.frame_set[i].id << " points " << ssc.cloud_vec[i]->points.size() << " clusters " << ssc.frame_set[i].cluster_set.size()
                      << " tracked " << cars << " dynamic " << dyn << "\n";
        }
        std::cout << "frames " << n << " dynamic_total " << dyn_total << "\n";
    } catch (const std::exception& e) {
        std::cerr << "scvod_sequence failed: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
