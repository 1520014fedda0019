// CPU build of the product's arithmetic spec header (dr-using-scv-od_amd/csrc/scvod_math.h)
// so that tests can compare it, function by function, with glibc and with the oracle.
// This is a test helper: the product never runs these on the CPU.
#include "../../dr-using-scv-od_amd/csrc/scvod_math.h"
#include <cmath>
#include <cstring>
extern "C" {
float spec_atan2f(float y, float x) { return scvod::atan2_f32(y, x); }
double spec_atan2(double y, double x) { return scvod::atan2_f64(y, x); }
// bulk comparisons against libm, returns number of mismatching results
long spec_cmp_atan2f(const float* y, const float* x, long n, long* first_bad) {
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        float a = scvod::atan2_f32(y[i], x[i]), b = atan2f(y[i], x[i]);
        if (scvod::f2u(a) != scvod::f2u(b) && !(a != a && b != b)) {
            if (!bad && first_bad) *first_bad = i;
            ++bad;
        }
    }
    return bad;
}
// max ulp distance of the fp64 spec to glibc atan2
double spec_cmp_atan2(const double* y, const double* x, long n) {
    double worst = 0;
    for (long i = 0; i < n; ++i) {
        double a = scvod::atan2_f64(y[i], x[i]), b = atan2(y[i], x[i]);
        long long ua = (long long)scvod::d2u(a), ub = (long long)scvod::d2u(b);
        double d = (double)(ua > ub ? ua - ub : ub - ua);
        if (d > worst) worst = d;
    }
    return worst;
}
// the branch-free angle estimate of czm_patch_of: largest |atan01_poly(t) - atan(t)| over the floats [bits_lo, bits_hi] (as bit patterns
// of non-negative floats, both inclusive), and largest |atan2_abs_rad_fast(|y|, x) - atan2(|y|, x)| over n points
double spec_atan01_worst(unsigned bits_lo, unsigned bits_hi, unsigned step) {
    double worst = 0;
    for (unsigned long b = bits_lo; b <= bits_hi; b += step) {
        float t;
        const unsigned u = (unsigned)b;
        std::memcpy(&t, &u, 4);
        const double e = std::fabs((double)scvod::atan01_poly(t) - std::atan((double)t));
        if (e > worst) worst = e;
    }
    return worst;
}
double spec_atan2_abs_worst(const float* ay, const float* x, long n) {
    double worst = 0;
    for (long i = 0; i < n; ++i) {
        const double e = std::fabs((double)scvod::atan2_abs_rad_fast(ay[i], x[i]) - std::atan2((double)ay[i], (double)x[i]));
        if (e > worst) worst = e;
    }
    return worst;
}
void spec_svd3(const float cov[9], float sv[3], float U[9]) {
    scvod::Svd3 s;
    scvod::svd3_jacobi(cov, s);
    for (int i = 0; i < 3; ++i) sv[i] = s.sv[i];
    for (int i = 0; i < 9; ++i) U[i] = s.U[i];
}
// patch ids with the reference's hard-coded Patchwork constants (patchwork.h:48-51,115-129)
void spec_patch_ids(float sensor_height, const float* xyzi, long n, int* pid) {
    scvod::CzmParams c;
    const double mn = 2.7, mx = 80.0;
    const int rings[4] = {2, 4, 4, 4}, secs[4] = {16, 32, 54, 32};
    c.min_range = mn; c.max_range = mx;
    c.zone_min[0] = mn; c.zone_min[1] = (7 * mn + mx) / 8.0; c.zone_min[2] = (3 * mn + mx) / 4.0; c.zone_min[3] = (mn + mx) / 2.0;
    c.ring_size[0] = (c.zone_min[1] - mn) / rings[0]; c.ring_size[1] = (c.zone_min[2] - c.zone_min[1]) / rings[1];
    c.ring_size[2] = (c.zone_min[3] - c.zone_min[2]) / rings[2]; c.ring_size[3] = (mx - c.zone_min[3]) / rings[3];
    int base = 0;
    for (int k = 0; k < 4; ++k) { c.sector_size[k] = 2 * M_PI / secs[k]; c.num_rings[k] = rings[k]; c.num_sectors[k] = secs[k]; c.patch_base[k] = base; base += rings[k] * secs[k]; }
    c.num_patches = base;
    c.z_cut = -1.8 * (double)sensor_height;
    scvod::czm_finalize(c);
    for (long i = 0; i < n; ++i) pid[i] = scvod::czm_patch_of(c, xyzi[4 * i], xyzi[4 * i + 1], xyzi[4 * i + 2]);
}
// fast range/FOV verdict vs the reference arithmetic on n points; returns the number of disagreements and, in
// *n_fast, how many points the shortcut decided
long spec_keep_compare(const float g[9], const float* xyz, long n, long* n_fast) {
    scvod::BinParams b;
    b.min_dis = g[0]; b.max_dis = g[1]; b.min_angle = g[2]; b.max_angle = g[3]; b.min_azimuth = g[4];
    b.max_azimuth = g[5]; b.range_res = g[6]; b.sector_res = g[7]; b.azimuth_res = g[8];
    b.range_num = b.sector_num = b.azimuth_num = b.bin_num = 1;
    const scvod::KeepFast k = scvod::keep_fast_of(b);
    scvod::KeepFast off = k;
    off.ok = 0;
    long bad = 0, fast = 0;
    for (long i = 0; i < n; ++i) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const int a = scvod::keep_of_point(b, k, x, y, z), e = scvod::keep_of_point(b, off, x, y, z);
        bad += (a != e);
        if (k.ok) {
            const float dis = scvod::point_distance2d(x, y);
            const float t = z / dis, m = 2.0e-6f * (1.0f + t * t);
            fast += (dis < b.min_dis || dis > b.max_dis || (t > k.tan_lo + m && t < k.tan_hi - m) || t < k.tan_lo - m || t > k.tan_hi + m);
        }
    }
    *n_fast = fast;
    return bad;
}
int spec_apri(const float g[9], const int dims[4], const float p[4], float out_f[7], int out_i[4]) {
    scvod::BinParams b;
    b.min_dis = g[0]; b.max_dis = g[1]; b.min_angle = g[2]; b.max_angle = g[3]; b.min_azimuth = g[4];
    b.max_azimuth = g[5]; b.range_res = g[6]; b.sector_res = g[7]; b.azimuth_res = g[8];
    b.range_num = dims[0]; b.sector_num = dims[1]; b.azimuth_num = dims[2]; b.bin_num = dims[3];
    scvod::Apri a;
    int keep = scvod::apri_of_point(b, p[0], p[1], p[2], p[3], a);
    out_f[0] = a.x; out_f[1] = a.y; out_f[2] = a.z; out_f[3] = a.range; out_f[4] = a.angle; out_f[5] = a.azimuth;
    out_f[6] = a.intensity;
    out_i[0] = a.range_idx; out_i[1] = a.sector_idx; out_i[2] = a.azimuth_idx; out_i[3] = a.voxel_idx;
    return keep;
}
// voxel_idx_fast against apri_of_point on n points: returns the number of DECIDED points whose index differs (must be 0);
// *n_fast = points the estimate decided; *worst_deg = largest |estimate - reference| of the two angles seen (degrees)
long spec_voxel_fast_compare(const float g[9], const int dims[3], const float* xyz, long n, long* n_fast, double* worst_deg) {
    scvod::BinParams b;
    b.min_dis = g[0]; b.max_dis = g[1]; b.min_angle = g[2]; b.max_angle = g[3]; b.min_azimuth = g[4];
    b.max_azimuth = g[5]; b.range_res = g[6]; b.sector_res = g[7]; b.azimuth_res = g[8];
    b.range_num = dims[0]; b.sector_num = dims[1]; b.azimuth_num = dims[2]; b.bin_num = dims[0] * dims[1] * dims[2];
    const scvod::BinFast f = scvod::bin_fast_of(b);
    long bad = 0, fast = 0;
    double worst = 0;
    for (long i = 0; i < n; ++i) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        scvod::Apri a;
        scvod::apri_of_point(b, x, y, z, 0.f, a);
        int32_t v = 0;
        if (scvod::voxel_idx_fast(b, f, x, y, z, &v)) {
            ++fast;
            int32_t ri, si, ai;
            scvod::idx3_fast(b, f, x, y, z, &ri, &si, &ai);
            if (v != a.voxel_idx || ri != a.range_idx || si != a.sector_idx || ai != a.azimuth_idx) ++bad;
        }
        if (scvod::fabs_f(y) > 0.0f) {
            float ang = scvod::atan2_abs_deg_fast(scvod::fabs_f(y), x);
            if (y < 0.0f) ang = 360.0f - ang;
            double d = std::fabs((double)ang - (double)a.angle);
            if (d > worst) worst = d;
            if (z != 0.0f) {
                float az = scvod::atan2_abs_deg_fast(scvod::fabs_f(z), a.range);
                if (z < 0.0f) az = -az;
                d = std::fabs((double)az - (double)a.azimuth);
                if (d > worst) worst = d;
            }
        }
    }
    if (n_fast) *n_fast = fast;
    if (worst_deg) *worst_deg = worst;
    return bad;
}
}
