// cov_math.h -- the arithmetic of one per-point covariance (fast_gicp_impl.hpp:241-297), shared by the covariance kernels of vgicp.hip
// (lane per query, search and arithmetic in one kernel: map-sized clouds) and cov_search.hip (search kernels that leave neighbour
// lists + one arithmetic kernel: scan-sized clouds).
#pragma once
#include "pcr_internal.h"

namespace pcr {

// ------------------------------------------------------------------------------
// symmetric 3x3 eigen-decomposition by cyclic Jacobi; eigenvalues descending, V columns
// ------------------------------------------------------------------------------
__device__ inline void sym3_eig(const double A[6] /* xx xy xz yy yz zz */, double w[3], double V[3][3]) {
    double a[3][3] = {{A[0], A[1], A[2]}, {A[1], A[3], A[4]}, {A[2], A[4], A[5]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-22 * diag) break;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] != 0.0) {
                    const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const double vkp = v[k][p], vkq = v[k][q]; v[k][p] = c * vkp - s * vkq; v[k][q] = s * vkp + c * vkq; }
                }
            }
        }
    }
    // sort descending with static indexing
    double e0 = a[0][0], e1 = a[1][1], e2 = a[2][2];
    double c0[3] = {v[0][0], v[1][0], v[2][0]}, c1[3] = {v[0][1], v[1][1], v[2][1]}, c2[3] = {v[0][2], v[1][2], v[2][2]};
#define SWAPCOL(ea, ca, eb, cb) if (eb > ea) { double t_ = ea; ea = eb; eb = t_; for (int k_ = 0; k_ < 3; ++k_) { double u_ = ca[k_]; ca[k_] = cb[k_]; cb[k_] = u_; } }
    // same selection order as the oracle: position 0 vs 1, 0 vs 2, then 1 vs 2
    SWAPCOL(e0, c0, e1, c1) SWAPCOL(e0, c0, e2, c2) SWAPCOL(e1, c1, e2, c2)
#undef SWAPCOL
    w[0] = e0; w[1] = e1; w[2] = e2;
#pragma unroll
    for (int k = 0; k < 3; ++k) { V[k][0] = c0[k]; V[k][1] = c1[k]; V[k][2] = c2[k]; }
}

// ------------------------------------------------------------------------------
// V2: covariance of every point of an indexed cloud (thread per cell-sorted point)
// cov6 is indexed by the ORIGINAL point index: xx xy xz yy yz zz
// ------------------------------------------------------------------------------
static constexpr int kCovK = 20;

// fast_gicp_impl.hpp:255-262: the K neighbours (ORIGINAL indices, in (distance, index) order; 0xffffffff = none: a cloud of fewer than K
// points) as f64, minus their mean, N N^T / k; JacobiSVD; PLANE regularisation.  Writes xx xy xz yy yz zz to dst; returns the neighbours found.
__device__ __forceinline__ int cov_from_neighbours(const uint32_t nb_idx[kCovK], const float* __restrict__ orig, uint32_t stride, double* __restrict__ dst) {
    double mx = 0, my = 0, mz = 0;
    int found = 0;
#pragma unroll
    for (int i = 0; i < kCovK; ++i) {
        if (nb_idx[i] != 0xffffffffu) {
            const float* p = orig + (size_t)nb_idx[i] * stride;
            mx += (double)p[0]; my += (double)p[1]; mz += (double)p[2];
            ++found;
        }
    }
    mx /= (double)kCovK; my /= (double)kCovK; mz /= (double)kCovK;
    double C[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < kCovK; ++i) {
        if (nb_idx[i] != 0xffffffffu) {
            const float* p = orig + (size_t)nb_idx[i] * stride;
            const double c0 = (double)p[0] - mx, c1 = (double)p[1] - my, c2 = (double)p[2] - mz;
            C[0] += c0 * c0; C[1] += c0 * c1; C[2] += c0 * c2; C[3] += c1 * c1; C[4] += c1 * c2; C[5] += c2 * c2;
        }
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) C[e] /= (double)kCovK;
    double w[3], V[3][3];
    sym3_eig(C, w, V);
    // PLANE: singular values replaced by (1, 1, 1e-3)   fast_gicp_impl.hpp:279-281,292
    const double val[3] = {1.0, 1.0, 1e-3};
    double out[6];
    int o = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = r; c < 3; ++c) {
            double s = 0;
#pragma unroll
            for (int e = 0; e < 3; ++e) s += V[r][e] * val[e] * V[c][e];
            out[o++] = s;
        }
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) dst[e] = out[e];
    return found;
}

}  // namespace pcr
