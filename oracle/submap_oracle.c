/*
 * submap_oracle.c -- CPU restatement of MapManager::updateMap's sub-map assembly
 * (frontend/src/MapManager.cpp:151-201):
 *   key frames within the search radius of the current position: KeyFramesKdtree::radiusSearch
 *   (third_parties/nanoflann/include/nanoflann/kfs_adaptor.hpp:57-75: squared L2 in double against radius*radius;
 *   nanoflann's RadiusResultSet keeps dist < radius, strictly),
 *   pcp::transformPointCloud with the pose cast to float (common/pcp/pcp.hpp:38-62: pto = tr * pfrom),
 *   concatenation, pcp::voxelDownSample (pcp.hpp:14-20 -> voxel_oracle.c).
 *
 * TEST INFRASTRUCTURE ONLY (tests/ and the bench's checker leg); the product path never calls it.
 * PARITY UNPINNED: the reference holds no test or golden vector for this step, and Eigen / PCL (the float transform and
 * the voxel filter) are external to /root/reference.  The transform is written as Eigen evaluates a 3x3 * 3x1
 * coefficient product followed by the translation, ((r0 x + r1 y) + r2 z) + t, without FMA contraction.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int oracle_voxel_filter(const float *pts, size_t n, size_t stride, float leaf, float *out, size_t cap, size_t *n_out);

/* clouds: concatenated key-frame points (stride floats each); counts[k] points per key frame; poses: 16 doubles each,
 * column-major.  selected (capacity n_kf) receives the indices of the key frames used, ascending.  out: capacity cap
 * points.  Returns oracle_voxel_filter's code. */
int oracle_submap_assemble(const float *clouds, const size_t *counts, const double *poses, size_t n_kf, size_t stride, const double position[3],
                           double radius, float grid, int64_t *selected, size_t *n_selected, float *out, size_t cap, size_t *n_out)
{
    *n_selected = 0; *n_out = 0;
    size_t total = 0, off = 0;
    const double r2 = radius * radius;
    for (size_t k = 0; k < n_kf; ++k) {
        double d = 0;
        for (int c = 0; c < 3; ++c) { const double e = position[c] - poses[k * 16 + 12 + c]; d += e * e; }
        if (d < r2) { selected[(*n_selected)++] = (int64_t)k; total += counts[k]; }
    }
    if (!total) return 0;
    float *cat = (float *)malloc(sizeof(float) * total * stride);
    size_t w = 0, s = 0;
    for (size_t k = 0; k < n_kf; ++k) {
        if (s < *n_selected && selected[s] == (int64_t)k) {
            float R[9], t[3];
            for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R[r * 3 + c] = (float)poses[k * 16 + c * 4 + r]; t[r] = (float)poses[k * 16 + 12 + r]; }
            for (size_t i = 0; i < counts[k]; ++i) {
                const float *p = clouds + (off + i) * stride;
                float *o = cat + w * stride;
                const float x = p[0], y = p[1], z = p[2];
                o[0] = ((R[0] * x + R[1] * y) + R[2] * z) + t[0];
                o[1] = ((R[3] * x + R[4] * y) + R[5] * z) + t[1];
                o[2] = ((R[6] * x + R[7] * y) + R[8] * z) + t[2];
                for (size_t c = 3; c < stride; ++c) o[c] = p[c];
                ++w;
            }
            ++s;
        }
        off += counts[k];
    }
    const int rc = oracle_voxel_filter(cat, total, stride, grid, out, cap, n_out);
    free(cat);
    return rc;
}
