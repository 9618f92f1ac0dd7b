/*
 * voxel_oracle.c -- CPU restatement of pcl::VoxelGrid<pcl::PointXYZI>::filter as the reference uses it
 * (leaf = (g, g, g), downsample_all_data = true, no field limits, min_points_per_voxel = 0):
 *   frontend/src/LidarOdometry.cpp:36,170-171          every incoming scan
 *   frontend/src/MapManager.cpp:78,192 -> common/pcp/pcp.hpp:14-28     every rebuilt sub-map
 *
 * TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker;
 * the product path (simpleslam_amd/) never calls it.
 *
 * PARITY UNPINNED: pcl::VoxelGrid lives in PCL (an external dependency of the reference, version not pinned; the root
 * CMakeLists hints at ROS noetic => PCL 1.10), not under /root/reference, and the reference holds no test or golden
 * vector for it.  Restated from the published algorithm, PCL 1.10 filters/include/pcl/filters/impl/voxel_grid.hpp
 * (VoxelGrid<PointT>::applyFilter) and common/include/pcl/common/impl/accumulators.hpp (CentroidPoint):
 *   getMinMax3D over the finite points; inverse_leaf_size = 1 / leaf (float);
 *   min_b = floor(min_p * inv), max_b = floor(max_p * inv), div_b = max_b - min_b + 1; (dx*dy*dz) > INT_MAX -> warning
 *   and output = input; per finite point ijk = (int)(floor(p * inv) - (float)min_b),
 *   idx = ijk0 + ijk1 * div_b0 + ijk2 * div_b0 * div_b1; sort by idx; one CentroidPoint per run of equal idx:
 *   float sums of xyz and intensity in the order the sort left the points in, divided by the count (as float).
 * PCL's std::sort is not stable, so the order inside a voxel -- and with it the last bits of the float sums -- is
 * unspecified there; this file uses input order (a stable sort), one of the admissible outcomes.
 */
#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int idx; uint32_t pt; } vox_ref;

static int cmp_ref(const void *a, const void *b)
{
    const vox_ref *x = (const vox_ref *)a, *y = (const vox_ref *)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pt < y->pt ? -1 : (x->pt > y->pt ? 1 : 0);     /* input order inside a voxel */
}

/* Returns 0 on success, 1 when PCL would have returned the input unfiltered (leaf too small), 2 when cap is too small
 * (n_out still holds the required size).  stride in floats (8 = pcl::PointXYZI, 4 = x y z intensity). */
int oracle_voxel_filter(const float *pts, size_t n, size_t stride, float leaf, float *out, size_t cap, size_t *n_out)
{
    *n_out = 0;
    const int ii = stride >= 8 ? 4 : (stride >= 4 ? 3 : -1);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    size_t n_fin = 0;
    for (size_t i = 0; i < n; ++i) {
        const float *p = pts + i * stride;
        if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
        for (int d = 0; d < 3; ++d) { if (p[d] < mn[d]) mn[d] = p[d]; if (p[d] > mx[d]) mx[d] = p[d]; }
        ++n_fin;
    }
    if (!n_fin) return 0;
    const float inv = 1.0f / leaf;
    int min_b[3], div_b[3];
    double cells = 1.0;
    for (int d = 0; d < 3; ++d) {
        const float lo = floorf(mn[d] * inv), hi = floorf(mx[d] * inv);
        cells *= (double)hi - (double)lo + 1.0;
        min_b[d] = fabsf(lo) < 2.0e9f ? (int)lo : 0;
        div_b[d] = (double)hi - (double)lo + 1.0 < 2.0e9 ? (int)((double)hi - (double)lo + 1.0) : INT_MAX;
    }
    if (cells > (double)INT_MAX) {          /* "Leaf size is too small for the input dataset" */
        *n_out = n;
        if (cap < n) return 2;
        memcpy(out, pts, n * stride * sizeof(float));
        return 1;
    }
    vox_ref *ref = (vox_ref *)malloc(sizeof(vox_ref) * n_fin);
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        const float *p = pts + i * stride;
        if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
        const int i0 = (int)(floorf(p[0] * inv) - (float)min_b[0]);
        const int i1 = (int)(floorf(p[1] * inv) - (float)min_b[1]);
        const int i2 = (int)(floorf(p[2] * inv) - (float)min_b[2]);
        ref[m].idx = i0 + i1 * div_b[0] + i2 * div_b[0] * div_b[1];
        ref[m].pt = (uint32_t)i;
        ++m;
    }
    qsort(ref, m, sizeof(vox_ref), cmp_ref);
    size_t k = 0, a = 0;
    int rc = 0;
    while (a < m) {
        size_t b = a;
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        while (b < m && ref[b].idx == ref[a].idx) {
            const float *p = pts + (size_t)ref[b].pt * stride;
            sx += p[0]; sy += p[1]; sz += p[2];
            if (ii >= 0) si += p[ii];
            ++b;
        }
        if (k < cap) {
            float *o = out + k * stride;
            const float cnt = (float)(b - a);
            for (size_t c = 0; c < stride; ++c) o[c] = 0.f;
            o[0] = sx / cnt; o[1] = sy / cnt; o[2] = sz / cnt;
            if (stride >= 8) o[3] = 1.0f;
            if (ii >= 0) o[ii] = si / cnt;
        } else rc = 2;
        ++k;
        a = b;
    }
    free(ref);
    *n_out = k;
    return rc;
}

/* voxel index of one point and the lattice, for membership checks: writes min_b[3], div_b[3]; returns 0 when the cloud
 * has no finite point */
int oracle_voxel_lattice(const float *pts, size_t n, size_t stride, float leaf, int min_b[3], int div_b[3])
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    size_t n_fin = 0;
    for (size_t i = 0; i < n; ++i) {
        const float *p = pts + i * stride;
        if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
        for (int d = 0; d < 3; ++d) { if (p[d] < mn[d]) mn[d] = p[d]; if (p[d] > mx[d]) mx[d] = p[d]; }
        ++n_fin;
    }
    if (!n_fin) return 0;
    const float inv = 1.0f / leaf;
    for (int d = 0; d < 3; ++d) { min_b[d] = (int)floorf(mn[d] * inv); div_b[d] = (int)floorf(mx[d] * inv) - min_b[d] + 1; }
    return 1;
}
