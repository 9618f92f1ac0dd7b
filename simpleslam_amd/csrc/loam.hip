// loam.hip -- LOAM scan-to-map Gauss-Newton on gfx950 (hand-written HIP).
//
// One launch = one Gauss-Newton iteration of the reference's
//   PCR::LoamRegister::scan2Map            PCR/src/LoamRegister.cpp:99-223
// with its per-point body fused into a single kernel:
//   L2 transform + f32 cast                LoamRegister.cpp:126-130
//   L3 exact 5-NN + squared-distance gate  LoamRegister.cpp:47-72 (nanoflann there, uniform grid here)
//   L4 plane LS (col-piv Householder QR)   LoamRegister.cpp:29-45
//   L5 residual, weight, weight gate       LoamRegister.cpp:144-151, LoamRegister.hpp:75-77
//   L6 Jacobian row s*n^T [I | -p^]        LoamRegister.cpp:153-159, manifolds.hpp:63-68
//   L7 J^T J / J^T r reduction             LoamRegister.cpp:170-188 (omp critical + dense GEMM there)
// and, in the PROLOGUE of the next launch, executed redundantly by every block
// (bitwise identical, so no broadcast and no extra kernel boundary):
//   L8 LDLT solve, convergence test, SE(3) exp update   LoamRegister.cpp:198-216, manifolds.hpp:33-60
// The last launch (loam_finalize_kernel) runs only that prologue plus
//   L9 T2SE3 re-orthonormalisation        LoamRegister.cpp:220, trans.hpp:54-65
//
// Arithmetic is f64 with contraction off (-ffp-contract=off), in the reference's
// operation order, so per-point results are bit-identical to the CPU oracle
// wherever only IEEE +,-,*,/,sqrt are involved.
#include "pcr_internal.h"

namespace pcr {

// ------------------------------------------------------------------------------
// small f64 routines, single lane
// ------------------------------------------------------------------------------

// Eigen::LDLT<Matrix6d> (lower, diagonal pivoting) restated; m: full symmetric 6x6 row-major.
__device__ void ldlt6_solve(const double* M, const double* rhs, double* x) {
    double m[6][6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) m[i][j] = (j <= i) ? M[i * 6 + j] : M[j * 6 + i];
    int tr[6];
    for (int k = 0; k < 6; ++k) {
        int p = k; double big = fabs(m[k][k]);
        for (int i = k + 1; i < 6; ++i) if (fabs(m[i][i]) > big) { big = fabs(m[i][i]); p = i; }
        tr[k] = p;
        if (p != k) {
            for (int j = 0; j < k; ++j) { double t = m[k][j]; m[k][j] = m[p][j]; m[p][j] = t; }
            for (int i = p + 1; i < 6; ++i) { double t = m[i][k]; m[i][k] = m[i][p]; m[i][p] = t; }
            for (int i = k + 1; i < p; ++i) { double t = m[i][k]; m[i][k] = m[p][i]; m[p][i] = t; }
            double t = m[k][k]; m[k][k] = m[p][p]; m[p][p] = t;
        }
        if (k > 0) {
            double temp[6];
            for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
            double s = 0; for (int j = 0; j < k; ++j) s += m[k][j] * temp[j];
            m[k][k] -= s;
            for (int i = k + 1; i < 6; ++i) {
                double s2 = 0; for (int j = 0; j < k; ++j) s2 += m[i][j] * temp[j];
                m[i][k] -= s2;
            }
        }
        const double piv = m[k][k];
        if (fabs(piv) > 0.0) for (int i = k + 1; i < 6; ++i) m[i][k] /= piv;
    }
    double y[6];
    for (int i = 0; i < 6; ++i) y[i] = rhs[i];
    for (int k = 0; k < 6; ++k) if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    for (int i = 0; i < 6; ++i) for (int j = 0; j < i; ++j) y[i] -= m[i][j] * y[j];
    for (int i = 0; i < 6; ++i) y[i] = (fabs(m[i][i]) > 2.2250738585072014e-308) ? y[i] / m[i][i] : 0.0;
    for (int i = 5; i >= 0; --i) for (int j = i + 1; j < 6; ++j) y[i] -= m[j][i] * y[j];
    for (int k = 5; k >= 0; --k) if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    for (int i = 0; i < 6; ++i) x[i] = y[i];
}

// manifolds::exp(V6) (manifolds.hpp:33-60): k = [rho; omega], T column-major.
__device__ void se3_exp(const double* k, double* T) {
    const double* p = k; const double* w = k + 3;
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    const double t = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    if (t < 1e-6) { T[12] = p[0]; T[13] = p[1]; T[14] = p[2]; return; }
    const double a[3] = {w[0] / t, w[1] / t, w[2] / t};
    const double ct = cos(t), st = sin(t);
    const double ah[3][3] = {{0, -a[2], a[1]}, {a[2], 0, -a[0]}, {-a[1], a[0], 0}};
    double V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double I = (i == j) ? 1.0 : 0.0, aa = a[i] * a[j];
            T[j * 4 + i] = ct * I + (1.0 - ct) * aa + st * ah[i][j];
            V[i][j] = st / t * I + (1.0 - st / t) * aa + ((1 - ct) / t) * ah[i][j];
        }
    for (int i = 0; i < 3; ++i) T[12 + i] = V[i][0] * p[0] + V[i][1] * p[1] + V[i][2] * p[2];
}

// trans::T2SE3 (trans.hpp:54-65): R <- Quaternion(R).normalized().toRotationMatrix().
__device__ void t2se3(double* T) {
#define M_(i, j) T[(j) * 4 + (i)]
    double q[4];
    double t = M_(0, 0) + M_(1, 1) + M_(2, 2);
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (M_(2, 1) - M_(1, 2)) * t; q[1] = (M_(0, 2) - M_(2, 0)) * t; q[2] = (M_(1, 0) - M_(0, 1)) * t;
    } else {
        int i = 0;
        if (M_(1, 1) > M_(0, 0)) i = 1;
        if (M_(2, 2) > M_(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(M_(i, i) - M_(j, j) - M_(k, k) + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (M_(k, j) - M_(j, k)) * t;
        q[j] = (M_(j, i) + M_(i, j)) * t;
        q[k] = (M_(k, i) + M_(i, k)) * t;
    }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
                 tyz = tz * y, tzz = tz * z;
    M_(0, 0) = 1 - (tyy + tzz); M_(0, 1) = txy - twz; M_(0, 2) = txz + twy;
    M_(1, 0) = txy + twz; M_(1, 1) = 1 - (txx + tzz); M_(1, 2) = tyz - twx;
    M_(2, 0) = txz - twy; M_(2, 1) = tyz + twx; M_(2, 2) = 1 - (txx + tyy);
#undef M_
}

// ------------------------------------------------------------------------------
// per-lane exact 5-NN on the grid
// ------------------------------------------------------------------------------
struct Knn5 {
    double d[5];
    uint32_t idx[5];   // original target index (tie-break key)
    uint32_t pos[5];   // position in the sorted array
};

__device__ __forceinline__ bool knn_less(double d, uint32_t i, double d2, uint32_t i2) {
    return d < d2 || (d == d2 && i < i2);
}

__device__ __forceinline__ void knn_insert(Knn5& s, double d, uint32_t idx, uint32_t pos) {
    // caller guarantees (d,idx) < (s.d[4], s.idx[4]); straight-line sorted insert
    const bool c3 = knn_less(d, idx, s.d[3], s.idx[3]);
    const bool c2 = knn_less(d, idx, s.d[2], s.idx[2]);
    const bool c1 = knn_less(d, idx, s.d[1], s.idx[1]);
    const bool c0 = knn_less(d, idx, s.d[0], s.idx[0]);
    s.d[4] = c3 ? s.d[3] : d;   s.idx[4] = c3 ? s.idx[3] : idx;   s.pos[4] = c3 ? s.pos[3] : pos;
    s.d[3] = c3 ? (c2 ? s.d[2] : d) : s.d[3];   s.idx[3] = c3 ? (c2 ? s.idx[2] : idx) : s.idx[3];   s.pos[3] = c3 ? (c2 ? s.pos[2] : pos) : s.pos[3];
    s.d[2] = c2 ? (c1 ? s.d[1] : d) : s.d[2];   s.idx[2] = c2 ? (c1 ? s.idx[1] : idx) : s.idx[2];   s.pos[2] = c2 ? (c1 ? s.pos[1] : pos) : s.pos[2];
    s.d[1] = c1 ? (c0 ? s.d[0] : d) : s.d[1];   s.idx[1] = c1 ? (c0 ? s.idx[0] : idx) : s.idx[1];   s.pos[1] = c1 ? (c0 ? s.pos[0] : pos) : s.pos[1];
    s.d[0] = c0 ? d : s.d[0];   s.idx[0] = c0 ? idx : s.idx[0];   s.pos[0] = c0 ? pos : s.pos[0];
}

// Exact 5 nearest target points with squared distance <= max_sq (ties on the original
// index), searching the 3x3x3 cell block as 9 contiguous x-runs.  Returns false when the
// query lies outside the searchable grid.  On return s.d[4] < max_sq  <=>  the reference's
// gate pointSearchSqDis[4] < mKdtreeMaxSearchDist (LoamRegister.cpp:59) passes.
__device__ __forceinline__ bool knn5_grid(const GridHeader& h, const float4* __restrict__ pts,
                                          const uint32_t* __restrict__ cell_start, double qx, double qy, double qz,
                                          double max_sq, Knn5& s) {
#pragma unroll
    for (int j = 0; j < 5; ++j) { s.d[j] = max_sq; s.idx[j] = 0xffffffffu; s.pos[j] = 0; }
    // cell coordinates (exact: q is a float widened to double, origin a multiple of cell)
    const double rx = qx - h.origin[0], ry = qy - h.origin[1], rz = qz - h.origin[2];
    const double fx = floor(rx * h.inv_cell), fy = floor(ry * h.inv_cell), fz = floor(rz * h.inv_cell);
    // queries in the outermost cell (or beyond, or NaN) are >= one cell away from every point
    if (!(fx >= 1.0 && fx <= (double)(h.dims[0] - 2) && fy >= 1.0 && fy <= (double)(h.dims[1] - 2) && fz >= 1.0 &&
          fz <= (double)(h.dims[2] - 2)))
        return false;
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
    // distance from the query to the lower / upper faces of its cell along y and z
    const double ylo = ry - fy * h.cell, yhi = (fy + 1.0) * h.cell - ry;
    const double zlo = rz - fz * h.cell, zhi = (fz + 1.0) * h.cell - rz;
    // rows ordered centre, faces, corners so that the bound prunes early
    constexpr int DY[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
    constexpr int DZ[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const double gy = DY[r] < 0 ? ylo : (DY[r] > 0 ? yhi : 0.0);
        const double gz = DZ[r] < 0 ? zlo : (DZ[r] > 0 ? zhi : 0.0);
        // every point of this row is at least sqrt(gy^2+gz^2) away (same rounding order as d below)
        const double bound = gy * gy + gz * gz;
        if (bound > s.d[4]) continue;
        const uint32_t key = ((uint32_t)(cz + DZ[r]) * (uint32_t)h.dims[1] + (uint32_t)(cy + DY[r])) * (uint32_t)h.dims[0] + (uint32_t)cx;
        uint32_t j = cell_start[key - 1];
        const uint32_t e = cell_start[key + 2];
        for (; j < e; ++j) {
            const float4 p = pts[j];
            const double dx = qx - (double)p.x, dy = qy - (double)p.y, dz = qz - (double)p.z;
            double d = dx * dx;      // nanoflann L2_Simple_Adaptor::evalMetric order
            d += dy * dy;
            d += dz * dz;
            const uint32_t idx = __float_as_uint(p.w);
            if (knn_less(d, idx, s.d[4], s.idx[4])) knn_insert(s, d, idx, j);
        }
    }
    return true;
}

// ------------------------------------------------------------------------------
// plane fit: Eigen::ColPivHouseholderQR<Matrix<double,5,3>>::solve(-1) restated
// (LoamRegister.cpp:29-35), all indices static so everything stays in registers.
// ------------------------------------------------------------------------------
__device__ __forceinline__ void swap_d(double& a, double& b) { const double t = a; a = b; b = t; }

__device__ __forceinline__ void plane_qr_solve(double a[5][3], double x[3]) {
    double c[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};
    int perm[3] = {0, 1, 2};
    double tau[3] = {0, 0, 0};
    double nu[3], nd[3];
    double maxn = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double s = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) s += a[i][j] * a[i][j];
        nd[j] = nu[j] = sqrt(s);
        if (nu[j] > maxn) maxn = nu[j];
    }
    const double eps = 2.220446049250313e-16;
    const double thr_helper = (maxn * eps) * (maxn * eps) / 5.0;
    const double downdate_thr = 1.4901161193847656e-08;  // sqrt(eps)
    int nonzero = 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int big = k; double bign = nu[k];
#pragma unroll
        for (int j = k + 1; j < 3; ++j) if (nu[j] > bign) { bign = nu[j]; big = j; }
        const double big_sq = bign * bign;
        if (nonzero == 3 && big_sq < thr_helper * (double)(5 - k)) nonzero = k;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (big == j) {
#pragma unroll
                for (int i = 0; i < 5; ++i) swap_d(a[i][k], a[i][j]);
                swap_d(nu[k], nu[j]); swap_d(nd[k], nd[j]);
                const int t = perm[k]; perm[k] = perm[j]; perm[j] = t;
            }
        }
        double tail_sq = 0;
#pragma unroll
        for (int i = k + 1; i < 5; ++i) tail_sq += a[i][k] * a[i][k];
        const double c0 = a[k][k];
        double beta;
        if (tail_sq <= 2.2250738585072014e-308) {
            tau[k] = 0; beta = c0;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) a[i][k] = 0;
        } else {
            beta = sqrt(c0 * c0 + tail_sq);
            if (c0 >= 0) beta = -beta;
            const double den = c0 - beta;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) a[i][k] /= den;
            tau[k] = (beta - c0) / beta;
        }
        a[k][k] = beta;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            double tmp = 0;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) tmp += a[i][k] * a[i][j];
            tmp += a[k][j];
            a[k][j] -= tau[k] * tmp;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) a[i][j] -= tau[k] * a[i][k] * tmp;
        }
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            if (nu[j] != 0) {
                double temp = fabs(a[k][j]) / nu[j];
                temp = (1.0 + temp) * (1.0 - temp);
                if (temp < 0) temp = 0;
                const double r = nu[j] / nd[j];
                const double temp2 = temp * r * r;
                if (temp2 <= downdate_thr) {
                    double s = 0;
#pragma unroll
                    for (int i = k + 1; i < 5; ++i) s += a[i][j] * a[i][j];
                    nd[j] = sqrt(s); nu[j] = nd[j];
                } else {
                    nu[j] *= sqrt(temp);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < nonzero) {
            double tmp = 0;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) tmp += a[i][k] * c[i];
            tmp += c[k];
            c[k] -= tau[k] * tmp;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) c[i] -= tau[k] * a[i][k] * tmp;
        }
    }
    double y[3] = {0, 0, 0};
#pragma unroll
    for (int i = 2; i >= 0; --i) {
        if (i < nonzero) {
            double s = c[i];
#pragma unroll
            for (int j = i + 1; j < 3; ++j) if (j < nonzero) s -= a[i][j] * y[j];
            y[i] = s / a[i][i];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double v = (i < nonzero) ? y[i] : 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) if (perm[i] == j) x[j] = v;
    }
}

// ------------------------------------------------------------------------------
// one scan point: returns status (0 accepted, 1 k-NN gate, 2 plane gate, 3 weight gate)
// row[0..5] = s*[n ; p x n], row[6] = s*d
// ------------------------------------------------------------------------------
__device__ __forceinline__ int loam_point(const LoamArgs& a, const GridHeader& h, const double* __restrict__ pose,
                                          const float* __restrict__ sp, double row[7], uint32_t nn_idx[5],
                                          bool* in_tile) {
    const float sx = sp[0], sy = sp[1], sz = sp[2];
    const double ox = (double)sx, oy = (double)sy, oz = (double)sz;
    // LoamRegister.cpp:126-130: Isometry3d * Vector4d in f64, then cast to f32
    const float px = (float)(pose[0] * ox + pose[4] * oy + pose[8] * oz + pose[12] * 1.0);
    const float py = (float)(pose[1] * ox + pose[5] * oy + pose[9] * oz + pose[13] * 1.0);
    const float pz = (float)(pose[2] * ox + pose[6] * oy + pose[10] * oz + pose[14] * 1.0);
    const double qx = (double)px, qy = (double)py, qz = (double)pz;
    *in_tile = true;
    if (a.use_tile) {
        *in_tile = qx >= a.tile_lo[0] && qx < a.tile_hi[0] && qy >= a.tile_lo[1] && qy < a.tile_hi[1] &&
                   qz >= a.tile_lo[2] && qz < a.tile_hi[2];
        if (!*in_tile) return 1;
    }
    Knn5 s;
    if (h.empty || h.overflow || !knn5_grid(h, a.grid.pts, a.grid.cell_start, qx, qy, qz, a.c.knn_max_sq, s)) return 1;
#pragma unroll
    for (int j = 0; j < 5; ++j) nn_idx[j] = s.idx[j];
    if (!(s.d[4] < a.c.knn_max_sq)) return 1;
    double A[5][3], Aq[5][3];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float4 p = a.grid.pts[s.pos[j]];
        A[j][0] = (double)p.x; A[j][1] = (double)p.y; A[j][2] = (double)p.z;
        Aq[j][0] = A[j][0]; Aq[j][1] = A[j][1]; Aq[j][2] = A[j][2];
    }
    double x[3];
    plane_qr_solve(Aq, x);
    const double xn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    // LoamRegister.cpp:38-43
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const double dot = x[0] * A[i][0] + x[1] * A[i][1] + x[2] * A[i][2];
        if (fabs(dot + 1.0) > a.c.plane_thresh * xn) ok = false;
    }
    if (!ok) return 2;
    const double dist = (qx * x[0] + qy * x[1] + qz * x[2] + 1.0) / xn;   // LoamRegister.hpp:75-77
    const float r2 = sx * sx + sy * sy + sz * sz;                           // LoamRegister.cpp:147-148 (float)
    const float rr = sqrtf(sqrtf(r2));
    const double w = 1 - 0.9 * fabs(dist) / (double)rr;
    if (!(w > a.c.point_thresh)) return 3;
    const double n0 = x[0] / xn, n1 = x[1] / xn, n2 = x[2] / xn;            // LoamRegister.hpp:70-73
    const double s0 = w * n0, s1 = w * n1, s2 = w * n2;
    row[0] = s0; row[1] = s1; row[2] = s2;
    row[3] = s1 * (-qz) + s2 * qy;       // s*n^T * (-skew(p)), manifolds.hpp:63-68, matrix.hpp:13-18
    row[4] = s0 * qz + s2 * (-qx);
    row[5] = s0 * (-qy) + s1 * qx;
    row[6] = w * dist;
    return 0;
}

// ------------------------------------------------------------------------------
// prologue: fold the previous launch's partial sums, solve, update the pose.
// Every block runs it and obtains bit-identical results.  Returns true when this
// launch must not linearise (loop finished).
// ------------------------------------------------------------------------------
struct Prologue {
    double pose[16];
    int done;
};

__device__ bool loam_prologue(const LoamArgs& a, int k, double* sh_sum /* 8*32 */, Prologue* sh) {
    const LoamState* prev = &a.state[(k + 1) & 1];
    LoamState* cur = &a.state[k & 1];
    const int t = threadIdx.x;
    if (k == 0) {
        if (t < 16) sh->pose[t] = a.init_pose[t];
        if (t == 0) sh->done = 0;
        if (blockIdx.x == 0 && t == 0) {
            for (int i = 0; i < 16; ++i) cur->pose[i] = a.init_pose[i];
            cur->done = 0; cur->converged = 0; cur->iters_run = 0; cur->fail = 0;
        }
        __syncthreads();
        return false;
    }
    if (prev->done) {
        if (t < 16) sh->pose[t] = prev->pose[t];
        if (t == 0) sh->done = 1;
        if (blockIdx.x == 0 && t == 0) { *cur = *prev; }
        __syncthreads();
        return true;
    }
    // fixed-order reduction of the partial sums of launch k-1
    const int comp = t & 31, slice = t >> 5;
    double acc = 0.0;
    if (a.reduced) {
        if (slice == 0) acc = a.reduced[comp];
    } else {
        const double* part = a.partials + (size_t)((k + 1) & 1) * kMaxPartials * kAccum;
        for (uint32_t b = slice; b < a.n_partials; b += 8) acc += part[(size_t)b * kAccum + comp];
    }
    sh_sum[slice * 32 + comp] = acc;
    __syncthreads();
    if (t < 32) {
        double v = sh_sum[t];
#pragma unroll
        for (int s = 1; s < 8; ++s) v += sh_sum[s * 32 + t];
        sh_sum[t] = v;
    }
    __syncthreads();
    if (t == 0) {
        double JtJ[36], rhs[6], x[6] = {0, 0, 0, 0, 0, 0};
        int q = 0;
        for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { JtJ[r * 6 + c] = JtJ[c * 6 + r] = sh_sum[q++]; }
        for (int r = 0; r < 6; ++r) rhs[r] = -sh_sum[21 + r];
        const double n = sh_sum[27];
        int done = 0, conv = 0, fail = 0;
        double pose[16];
        for (int i = 0; i < 16; ++i) pose[i] = prev->pose[i];
        if (n < 6.0) { done = 1; fail = 1; }                      // LoamRegister.cpp:173-176
        else {
            ldlt6_solve(JtJ, rhs, x);                              // LoamRegister.cpp:198
            const double np = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
            const double nr = sqrt(x[3] * x[3] + x[4] * x[4] + x[5] * x[5]);
            if (a.c.early_exit && np <= a.c.pos_conv && nr <= a.c.rot_conv) { done = 1; conv = 1; }   // :202-206
            else {
                double E[16], out[16];
                se3_exp(x, E);                                     // :213-216
                for (int c = 0; c < 4; ++c)
                    for (int r = 0; r < 4; ++r) {
                        double s = 0;
                        for (int kk = 0; kk < 4; ++kk) s += E[kk * 4 + r] * pose[c * 4 + kk];
                        out[c * 4 + r] = s;
                    }
                for (int i = 0; i < 16; ++i) pose[i] = out[i];
            }
        }
        if (k >= a.c.iters) done = 1;
        for (int i = 0; i < 16; ++i) sh->pose[i] = pose[i];
        sh->done = done;
        if (blockIdx.x == 0) {
            for (int i = 0; i < 16; ++i) cur->pose[i] = pose[i];
            cur->done = done; cur->converged = conv; cur->fail = fail; cur->iters_run = k;
            if (a.trace) {
                LoamTrace* tr = &a.trace[k - 1];
                for (int i = 0; i < 36; ++i) tr->JtJ[i] = JtJ[i];
                for (int i = 0; i < 6; ++i) { tr->JtE[i] = -rhs[i]; tr->x[i] = x[i]; }
                tr->n = (int64_t)n;
            }
        }
    }
    __syncthreads();
    return sh->done != 0;
}

// ------------------------------------------------------------------------------
// the iteration kernel
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void loam_iterate_kernel(const LoamArgs a, const int k) {
    __shared__ double sh_sum[8 * 32];
    __shared__ Prologue sh_pro;
    if (loam_prologue(a, k, sh_sum, &sh_pro)) return;
    const GridHeader h = *a.grid.hdr;
    double pose[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pose[i] = sh_pro.pose[i];

    // XCD-aware mapping: consecutive logical blocks (adjacent lidar rings) share an XCD's L2
    uint32_t blk = blockIdx.x;
    if ((gridDim.x & 7u) == 0) blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);

    double acc[28];
#pragma unroll
    for (int i = 0; i < 28; ++i) acc[i] = 0.0;
    for (uint32_t q = blk * 256 + threadIdx.x; q < a.n_src; q += gridDim.x * 256) {
        double row[7];
        uint32_t nn[5] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        bool in_tile;
        const int st = loam_point(a, h, pose, a.src + (size_t)q * a.src_stride, row, nn, &in_tile);
        if (a.dbg_status) a.dbg_status[q] = (int8_t)(in_tile ? st : 4);
        if (a.dbg_nn) { for (int j = 0; j < 5; ++j) a.dbg_nn[(size_t)q * 5 + j] = (int32_t)nn[j]; }
        if (a.dbg_rows) { for (int j = 0; j < 7; ++j) a.dbg_rows[(size_t)q * 7 + j] = st == 0 ? row[j] : 0.0; }
        if (st == 0) {
            int qq = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int c = r; c < 6; ++c) acc[qq++] += row[r] * row[c];
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[21 + r] += row[r] * row[6];
            acc[27] += 1.0;
        }
    }
    // wave64 butterfly, then the 4 waves through LDS, fixed order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 28; ++i) {
        double v = acc[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        acc[i] = v;
    }
    __syncthreads();   // sh_sum is reused
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 28; ++i) sh_sum[wave * 32 + i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        double v = 0.0;
        if (threadIdx.x < 28) v = ((sh_sum[threadIdx.x] + sh_sum[32 + threadIdx.x]) + sh_sum[64 + threadIdx.x]) + sh_sum[96 + threadIdx.x];
        a.partials[((size_t)(k & 1) * kMaxPartials + blockIdx.x) * kAccum + threadIdx.x] = v;
    }
}

// last launch: prologue only, then T2SE3 and the output pose (LoamRegister.cpp:220)
__global__ __launch_bounds__(256) void loam_finalize_kernel(const LoamArgs a, const int k) {
    __shared__ double sh_sum[8 * 32];
    __shared__ Prologue sh_pro;
    loam_prologue(a, k, sh_sum, &sh_pro);
    if (threadIdx.x == 0) {
        double T[16];
        for (int i = 0; i < 16; ++i) T[i] = sh_pro.pose[i];
        t2se3(T);
        const LoamState* cur = &a.state[k & 1];   // written by this thread in the prologue
        LoamResult* r = a.result;
        for (int i = 0; i < 16; ++i) r->pose[i] = T[i];
        r->converged = cur->converged; r->iters_run = cur->iters_run; r->fail = cur->fail;
        r->grid_overflow = a.grid.hdr->overflow; r->grid_empty = a.grid.hdr->empty; r->grid_cells = a.grid.hdr->n_cells;
        r->pad = 1;   // completion marker
    }
}

// sharded (multi-GPU) mode: fold this rank's partial sums into kAccum doubles so that
// RCCL can all-reduce them before the next launch's prologue reads a.reduced
__global__ __launch_bounds__(256) void loam_reduce_kernel(const LoamArgs a, const int k, double* __restrict__ out) {
    __shared__ double sh_sum[8 * 32];
    const int t = threadIdx.x, comp = t & 31, slice = t >> 5;
    const LoamState* st = &a.state[k & 1];
    double acc = 0.0;
    if (!st->done) {
        const double* part = a.partials + (size_t)(k & 1) * kMaxPartials * kAccum;
        for (uint32_t b = slice; b < a.n_partials; b += 8) acc += part[(size_t)b * kAccum + comp];
    }
    sh_sum[slice * 32 + comp] = acc;
    __syncthreads();
    if (t < 32) {
        double v = sh_sum[t];
#pragma unroll
        for (int s = 1; s < 8; ++s) v += sh_sum[s * 32 + t];
        out[t] = v;
    }
}

uint32_t loam_grid_blocks(uint32_t n_src) {
    uint32_t b = (n_src + 255) / 256;
    if (b < 1) b = 1;
    if (b > (uint32_t)kMaxPartials) b = kMaxPartials;
    return b;
}

hipError_t loam_launch_iteration(const LoamArgs& a, int k, hipStream_t s) {
    hipLaunchKernelGGL(loam_iterate_kernel, dim3(a.n_partials), dim3(256), 0, s, a, k);
    return hipGetLastError();
}
hipError_t loam_launch_finalize(const LoamArgs& a, int k, hipStream_t s) {
    hipLaunchKernelGGL(loam_finalize_kernel, dim3(1), dim3(256), 0, s, a, k);
    return hipGetLastError();
}
hipError_t loam_launch_reduce(const LoamArgs& a, int k, double* d_out, hipStream_t s) {
    hipLaunchKernelGGL(loam_reduce_kernel, dim3(1), dim3(256), 0, s, a, k, d_out);
    return hipGetLastError();
}

}  // namespace pcr
