// patchwork.h -- PatchWork<PointT> with the reference's public signature
// (/root/reference/include/patchwork.h:105-111); the body is one call through the C-ABI.
#ifndef SCVOD_HOST_PATCHWORK_H_
#define SCVOD_HOST_PATCHWORK_H_
#include <chrono>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/scvod.h"
#include "pcl_shim.h"

template <typename PointT>
class PatchWork {
  public:
    // ctx is borrowed from the owning SSC (one ctx per GPU)
    explicit PatchWork(scvod_ctx* ctx = nullptr) : ctx_(ctx) {}
    void attach(scvod_ctx* ctx) { ctx_ = ctx; }

    void estimate_ground(const pcl::PointCloud<PointT>& cloudIn, pcl::PointCloud<PointT>& cloudOut,
                         pcl::PointCloud<PointT>& cloudNonground, double& time_taken) {
        if (!ctx_) throw std::invalid_argument("PatchWork: no scvod ctx attached (GPU-only, no CPU fallback)");
        auto t0 = std::chrono::steady_clock::now();
        const int n = (int)cloudIn.points.size();
        stage_.resize((size_t)n * 4);
        for (int i = 0; i < n; ++i) {
            stage_[4 * i] = cloudIn.points[i].x;
            stage_[4 * i + 1] = cloudIn.points[i].y;
            stage_[4 * i + 2] = cloudIn.points[i].z;
            stage_[4 * i + 3] = cloudIn.points[i].intensity;
        }
        scvod_scan_result r;
        int rc = scvod_patchwork(ctx_, stage_.data(), n, &r);
        if (rc != SCVOD_OK) throw std::runtime_error(std::string("scvod_patchwork: ") + scvod_last_error(ctx_));
        cloudOut.clear();
        cloudNonground.clear();
        cloudOut.points.reserve(r.n_ground);
        cloudNonground.points.reserve(r.n_nonground);
        for (int k = 0; k < r.n_ground; ++k) cloudOut.points.push_back(cloudIn.points[r.ground_idx[k]]);
        for (int k = 0; k < r.n_nonground; ++k) cloudNonground.points.push_back(cloudIn.points[r.nonground_idx[k]]);
        time_taken = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    // patchwork.h:111.  The height reaches the GPU through scvod_params.sensor_height when the ctx is created (SSC's constructor,
    // ssc.cpp:93 calls set_sensor once with that same value): a DIFFERENT height afterwards cannot be honoured by this ctx and is
    // refused instead of being ignored silently.
    void set_sensor(const double& height) {
        scvod_params p;
        if (scvod_get_params(ctx_, &p) == SCVOD_OK && (float)height != p.sensor_height)
            throw std::invalid_argument("PatchWork::set_sensor: the ctx was created for sensor_height " + std::to_string(p.sensor_height) +
                                        ", not " + std::to_string(height) + " (create the SSC / scvod ctx with the new height)");
        sensor_height_ = height;
    }

  private:
    scvod_ctx* ctx_;
    double sensor_height_ = 0;
    std::vector<float> stage_;
};
#endif
