// pcl_shim.h -- the handful of PCL types the SSC / PatchWork signatures mention, for builds without
// PCL (this image has none).  With -DSCVOD_WITH_PCL the real headers are used instead and the facade
// compiles against pcl::PointCloud unchanged.
#ifndef SCVOD_PCL_SHIM_H_
#define SCVOD_PCL_SHIM_H_
#ifdef SCVOD_WITH_PCL
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#else
#include <memory>
#include <vector>
namespace pcl {
struct PointXYZI {
    float x = 0.f, y = 0.f, z = 0.f, intensity = 0.f;
};
template <typename PointT>
class PointCloud {
  public:
    typedef std::shared_ptr<PointCloud<PointT>> Ptr;
    typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
    std::vector<PointT> points;
    unsigned width = 0, height = 1;
    bool is_dense = true;
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void clear() { points.clear(); }
    void push_back(const PointT& p) { points.push_back(p); }
    PointT& operator[](size_t i) { return points[i]; }
    const PointT& operator[](size_t i) const { return points[i]; }
    PointCloud& operator+=(const PointCloud& o) {
        points.insert(points.end(), o.points.begin(), o.points.end());
        return *this;
    }
};
}  // namespace pcl
#endif
#endif
