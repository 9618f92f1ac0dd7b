// small_math.h -- tiny f64 linear-algebra routines shared by device kernels and host drivers.
// Restated from the Eigen calls of the reference (citations at each routine).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace pcr {

// ------------------------------------------------------------------------------
// small f64 routines, single lane
// ------------------------------------------------------------------------------

// Eigen::LDLT<Matrix6d> (lower, diagonal pivoting) restated; M: full symmetric 6x6 row-major.
// Fully unrolled with static indices (pivot swaps are `if (p == pp)` over the static candidates)
// so the factor lives in registers: the single solving lane never touches scratch memory.
__host__ __device__ inline void ldlt6_solve(const double* M, const double* rhs, double* x) {
    double m[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) m[i][j] = (j <= i) ? M[i * 6 + j] : 0.0;
    }
    int tr[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int p = k; double big = fabs(m[k][k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i) if (fabs(m[i][i]) > big) { big = fabs(m[i][i]); p = i; }
        tr[k] = p;
#pragma unroll
        for (int pp = k + 1; pp < 6; ++pp) {
            if (p == pp) {
#pragma unroll
                for (int j = 0; j < k; ++j) { const double t = m[k][j]; m[k][j] = m[pp][j]; m[pp][j] = t; }
#pragma unroll
                for (int i = pp + 1; i < 6; ++i) { const double t = m[i][k]; m[i][k] = m[i][pp]; m[i][pp] = t; }
#pragma unroll
                for (int i = k + 1; i < pp; ++i) { const double t = m[i][k]; m[i][k] = m[pp][i]; m[pp][i] = t; }
                const double t = m[k][k]; m[k][k] = m[pp][pp]; m[pp][pp] = t;
            }
        }
        if (k > 0) {
            double temp[6];
#pragma unroll
            for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
            double s = 0;
#pragma unroll
            for (int j = 0; j < k; ++j) s += m[k][j] * temp[j];
            m[k][k] -= s;
#pragma unroll
            for (int i = k + 1; i < 6; ++i) {
                double s2 = 0;
#pragma unroll
                for (int j = 0; j < k; ++j) s2 += m[i][j] * temp[j];
                m[i][k] -= s2;
            }
        }
        const double piv = m[k][k];
        if (fabs(piv) > 0.0) {
#pragma unroll
            for (int i = k + 1; i < 6; ++i) m[i][k] /= piv;
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = rhs[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int pp = k + 1; pp < 6; ++pp) if (tr[k] == pp) { const double t = y[k]; y[k] = y[pp]; y[pp] = t; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j < i; ++j) y[i] -= m[i][j] * y[j];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = (fabs(m[i][i]) > 2.2250738585072014e-308) ? y[i] / m[i][i] : 0.0;
#pragma unroll
    for (int i = 5; i >= 0; --i) {
#pragma unroll
        for (int j = i + 1; j < 6; ++j) y[i] -= m[j][i] * y[j];
    }
#pragma unroll
    for (int k = 5; k >= 0; --k) {
#pragma unroll
        for (int pp = k + 1; pp < 6; ++pp) if (tr[k] == pp) { const double t = y[k]; y[k] = y[pp]; y[pp] = t; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = y[i];
}

// manifolds::exp(V6) (manifolds.hpp:33-60): k = [rho; omega], T column-major.
__host__ __device__ inline void se3_exp(const double* k, double* T) {
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
__host__ __device__ inline void t2se3(double* T) {
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


}  // namespace pcr
