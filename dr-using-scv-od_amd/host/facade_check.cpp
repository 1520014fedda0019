// facade_check.cpp -- drives the SSC facade the way SSC::segDF drives the reference (process per scan,
// tracking per pair) on raw float32 scans and dumps what the reference would hold in its members, so
// that tests/test_gpu_facade.py can check them against the CPU restatement.
//   usage: facade_check <config.yaml> <scan_a.f32> <scan_b.f32> <out_prefix>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "ssc.h"

static pcl::PointCloud<pcl::PointXYZI>::Ptr load(const char* path) {
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
static void dump_cloud(const std::string& path, const pcl::PointCloud<pcl::PointXYZI>& c) {
    std::ofstream o(path, std::ios::binary);
    for (auto& p : c.points) o.write((const char*)&p, 16);
}
static void dump_state(const std::string& pre, SSC& s) {
    dump_cloud(pre + "_ground.f32", *s.g_cloud_vec.back());
    dump_cloud(pre + "_cloud_use.f32", *s.cloud_use);
    dump_cloud(pre + "_eva_static.f32", *s.cloud_eva_static);
    std::ofstream a(pre + "_apri.bin", std::ios::binary);
    a.write((const char*)s.apri_vec.data(), s.apri_vec.size() * sizeof(PointAPRI));
    std::vector<int> keys;
    for (auto& kv : s.hash_cloud) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    std::ofstream h(pre + "_hash.txt");
    for (int k : keys) {
        const Voxel& v = s.hash_cloud[k];
        uint32_t av, cov, c[4];
        memcpy(&av, &v.intensity_av, 4);
        memcpy(&cov, &v.intensity_cov, 4);
        memcpy(c, &v.center, 16);
        h << k << " " << v.range_idx << " " << v.sector_idx << " " << v.azimuth_idx << " " << v.label << " " << av << " " << cov << " "
          << c[0] << " " << c[1] << " " << c[2] << " " << c[3] << " " << v.ptIdx.size();
        for (int p : v.ptIdx) h << " " << p;
        h << "\n";
    }
}

// toy segmentation standing in for SSC::segment/recognize (out of scope): voxels grouped in coarse
// (range, sector) blocks; small groups become `car`
static void toy_segment(SSC& s, Frame& f) {
    f.hash_cloud = s.hash_cloud;
    f.cluster_set.clear();
    std::vector<int> vkeys;
    for (auto& kv : f.hash_cloud) vkeys.push_back(kv.first);
    std::sort(vkeys.begin(), vkeys.end());  // clusters are created in ascending voxel-key order (deterministic)
    for (int key : vkeys) {
        Voxel& v = f.hash_cloud[key];
        int name = 5 + (v.range_idx / 6) * 64 + (v.sector_idx / 12);
        v.label = name;
        Cluster& c = f.cluster_set[name];
        c.name = name;
        c.occupy_voxels.push_back(key);
        c.occupy_pts.insert(c.occupy_pts.end(), v.ptIdx.begin(), v.ptIdx.end());
    }
    f.max_name = f.name_floor = 5 + 64 * 64;
    for (auto& kv : f.cluster_set) {
        Cluster& c = kv.second;
        std::sort(c.occupy_voxels.begin(), c.occupy_voxels.end());
        std::sort(c.occupy_pts.begin(), c.occupy_pts.end());
        for (int p : c.occupy_pts) c.cloud->points.push_back(f.cloud_use->points[p]);
        c.type = (c.occupy_pts.size() < 400) ? s.car : s.tree;
    }
}

int main(int argc, char** argv) {
    if (argc < 5) {
        std::cerr << "usage: facade_check <config.yaml> <scan_a.f32> <scan_b.f32> <out_prefix>\n";
        return 2;
    }
    try {
        SSC ssc(argv[1]);
        std::string pre = argv[4];
        std::cout << "grid " << ssc.range_num << " " << ssc.sector_num << " " << ssc.azimuth_num << " " << ssc.bin_num << "\n";
        auto a = load(argv[2]), b = load(argv[3]);
        // fused process()
        ssc.process(a);
        dump_state(pre + "_a", ssc);
        std::vector<PointAPRI> apri_fused = ssc.apri_vec;
        size_t vox_fused = ssc.hash_cloud.size(), use_fused = ssc.cloud_use->size();
        Frame fa = ssc.frame_ssc;
        toy_segment(ssc, fa);
        ssc.reset();
        // the same scan through the three separate entry points must give the same members
        auto ng = ssc.extractGroudByPatchWork(a);
        ssc.makeApriVec(ng);
        ssc.makeHashCloud(ssc.apri_vec);
        bool same = ssc.apri_vec.size() == apri_fused.size() && ssc.hash_cloud.size() == vox_fused && ssc.cloud_use->size() == use_fused &&
                    memcmp(ssc.apri_vec.data(), apri_fused.data(), apri_fused.size() * sizeof(PointAPRI)) == 0;
        std::cout << "stepwise_equals_fused " << (same ? 1 : 0) << "\n";
        ssc.reset();
        ssc.process(b);
        dump_state(pre + "_b", ssc);
        Frame fb = ssc.frame_ssc;
        toy_segment(ssc, fb);
        // tracking: (a -> a) with identical poses: every car cluster is confirmed static
        Pose p0, p1;
        p1.x = 1.0f;
        Frame fa2 = fa, fa3 = fa;
        ssc.tracking(fa2, fa3, p0, p0);
        int cars = 0, stat = 0;
        for (auto& kv : fa2.cluster_set)
            if (kv.second.type == ssc.car || kv.second.state != -1) {
                cars += (kv.second.state != -1);
                stat += kv.second.state == 0;
            }
        std::cout << "self_tracking cars " << cars << " static " << stat << " dynamic " << ssc.dynamic_num_last << "\n";
        ssc.tracking(fa, fb, p0, p1);
        int c2 = 0, d2 = 0;
        std::ofstream st(pre + "_states.txt");
        std::vector<int> names;
        for (auto& kv : fa.cluster_set) names.push_back(kv.first);
        std::sort(names.begin(), names.end());
        for (int nme : names) {
            Cluster& c = fa.cluster_set[nme];
            if (c.state != -1) {
                ++c2;
                d2 += c.state == 1;
                st << nme << " " << c.state << " " << c.occupy_voxels.size() << "\n";
            }
        }
        std::cout << "pair_tracking cars " << c2 << " dynamic " << d2 << " reported " << ssc.dynamic_num_last << " next_clusters "
                  << fb.cluster_set.size() << "\n";
        // loader step (getCloud's filter + VoxelGrid) on scan a with synthetic labels: every 9th point unlabeled,
        // every 13th an outlier, the rest some class
        {
            std::vector<float> raw(4 * a->points.size());
            std::vector<uint32_t> lab(a->points.size());
            for (size_t i = 0; i < a->points.size(); ++i) {
                raw[4 * i] = a->points[i].x;
                raw[4 * i + 1] = a->points[i].y;
                raw[4 * i + 2] = a->points[i].z;
                raw[4 * i + 3] = a->points[i].intensity;
                lab[i] = (i % 9 == 0) ? 0u : (i % 13 == 0) ? 0x00030001u : (40u + (uint32_t)(i % 5));
            }
            auto loaded = ssc.filterAndDownsample(raw, lab);
            dump_cloud(pre + "_a_loaded.f32", *loaded);
            std::cout << "loaded " << a->points.size() << " " << loaded->points.size() << "\n";
        }
        std::vector<int> bkeys;
        for (auto& kv : fb.hash_cloud) bkeys.push_back(kv.first);
        std::sort(bkeys.begin(), bkeys.end());
        std::ofstream nl(pre + "_next_labels.txt");
        for (int k : bkeys) nl << fb.hash_cloud[k].label << "\n";
    } catch (const std::exception& e) {
        std::cerr << "facade_check failed: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
