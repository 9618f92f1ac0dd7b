/*
 * scancontext_oracle.c -- CPU restatement of the reference's loop-closure descriptor
 * (backend/src/ScanContext.cpp, itself adapted from irapkaist/scancontext; constants backend/include/backend/ScanContext.hpp:17-19):
 *   oracle_sc_make      makeScanContext :151-196 (sequential "keep the maximum z per bin" loop)
 *   oracle_sc_keys      makeRingkeyFromScanContext :198-213, makeSectorkeyFromScanContext :215-229
 *   oracle_sc_distance  circshift :34-54, computeSimularity :68-92, fastAlignUsingVkey :94-114, distanceBtnScanContext :116-150
 * The stateful part of query (:231-279) is restated in oracle/__init__.py (class ScanContextOracle).
 *
 * TEST INFRASTRUCTURE ONLY (tests/); the product path never calls it.
 * PARITY UNPINNED: the reference holds no test or golden vector for ScanContext, and Eigen / PCL / nanoflann's kd-tree
 * (an exact k-NN) are external to it; this file follows the source text.  Matrices are column-major as Eigen's MatrixXd.
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define SC_RING 20
#define SC_SECTOR 60
#define SC_MAX_RADIUS 80.0f     /* scpr_t float */

/* xy2theta<float> (:27-32) with trans::rad2deg<float> = rad * 180.0 / M_PI evaluated in double, returned as float */
static float xy2theta(float x, float y)
{
    float res = atan2f(y, x) + M_PI;
    const float lo = 0.0f, hi = (float)(2 * M_PI);
    res = res < hi ? res : hi;          /* std::min(T(2*M_PI), res) */
    res = lo < res ? res : lo;          /* std::max(T(0), ...) */
    return (float)(res * 180.0 / M_PI);
}

/* desc: SC_RING x SC_SECTOR, column-major */
void oracle_sc_make(const float *pts, size_t n, size_t stride, double lidar_height_d, double *desc)
{
    const float lidar_height = (float)lidar_height_d;       /* cfg.get<float>() */
    const int NO_POINT = -1000;
    for (int i = 0; i < SC_RING * SC_SECTOR; ++i) desc[i] = NO_POINT;
    for (size_t i = 0; i < n; ++i) {
        const float x = pts[i * stride], y = pts[i * stride + 1];
        const float z = pts[i * stride + 2] + lidar_height;
        const float azim_range = sqrtf(x * x + y * y);
        const float azim_angle = xy2theta(x, y);
        if (azim_range > SC_MAX_RADIUS) continue;
        int ring = (int)ceil((azim_range / SC_MAX_RADIUS) * SC_RING);      /* float arithmetic */
        int sctor = (int)ceil((azim_angle / 360.0) * SC_SECTOR);
        ring = ring < SC_RING ? ring : SC_RING; ring = ring > 1 ? ring : 1;
        sctor = sctor < SC_SECTOR ? sctor : SC_SECTOR; sctor = sctor > 1 ? sctor : 1;
        double *cell = &desc[(sctor - 1) * SC_RING + (ring - 1)];
        if (*cell < z) *cell = z;
    }
    for (int i = 0; i < SC_RING * SC_SECTOR; ++i) if (desc[i] == NO_POINT) desc[i] = 0;
}

void oracle_sc_keys(const double *desc, double *ring_key, double *sector_key)
{
    for (int r = 0; r < SC_RING; ++r) { double s = 0; for (int c = 0; c < SC_SECTOR; ++c) s += desc[c * SC_RING + r]; ring_key[r] = s / SC_SECTOR; }
    for (int c = 0; c < SC_SECTOR; ++c) { double s = 0; for (int r = 0; r < SC_RING; ++r) s += desc[c * SC_RING + r]; sector_key[c] = s / SC_RING; }
}

/* shift the columns of a rows x SC_SECTOR matrix to the right */
static void circshift(const double *m, int rows, int shift, double *out)
{
    if (shift == 0) { memcpy(out, m, sizeof(double) * rows * SC_SECTOR); return; }
    for (int c = 0; c < SC_SECTOR; ++c) memcpy(out + ((c + shift) % SC_SECTOR) * rows, m + c * rows, sizeof(double) * rows);
}

static double similarity(const double *a, const double *b)
{
    int eff = 0;
    double sum = 0;
    for (int c = 0; c < SC_SECTOR; ++c) {
        double na = 0, nb = 0, dot = 0;
        for (int r = 0; r < SC_RING; ++r) { na += a[c * SC_RING + r] * a[c * SC_RING + r]; nb += b[c * SC_RING + r] * b[c * SC_RING + r]; dot += a[c * SC_RING + r] * b[c * SC_RING + r]; }
        na = sqrt(na); nb = sqrt(nb);
        if (na == 0 || nb == 0) continue;
        sum = sum + dot / (na * nb);
        eff = eff + 1;
    }
    return 1.0 - sum / eff;
}

void oracle_sc_distance(const double *sc1, const double *sc2, double search_ratio_d, double *dist, int *shift)
{
    const float search_ratio = (float)search_ratio_d;
    double k1[SC_SECTOR], k2[SC_SECTOR], rk[SC_RING], k2s[SC_SECTOR];
    oracle_sc_keys(sc1, rk, k1);
    oracle_sc_keys(sc2, rk, k2);
    int argmin_vkey_shift = 0;
    double min_veky_diff_norm = DBL_MAX;
    for (int s = 0; s < SC_SECTOR; ++s) {
        circshift(k2, 1, s, k2s);
        double nrm = 0;
        for (int c = 0; c < SC_SECTOR; ++c) nrm += (k1[c] - k2s[c]) * (k1[c] - k2s[c]);
        nrm = sqrt(nrm);
        if (nrm < min_veky_diff_norm) { argmin_vkey_shift = s; min_veky_diff_norm = nrm; }
    }
    const int radius = (int)round(0.5 * search_ratio * SC_SECTOR);
    int space[2 * SC_SECTOR + 1], ns = 0;
    space[ns++] = argmin_vkey_shift;
    for (int ii = 1; ii < radius + 1; ++ii) {
        space[ns++] = (argmin_vkey_shift + ii + SC_SECTOR) % SC_SECTOR;
        space[ns++] = (argmin_vkey_shift - ii + SC_SECTOR) % SC_SECTOR;
    }
    for (int i = 1; i < ns; ++i) { const int v = space[i]; int j = i; while (j > 0 && space[j - 1] > v) { space[j] = space[j - 1]; --j; } space[j] = v; }
    int arg = 0;
    double best = DBL_MAX;
    double shifted[SC_RING * SC_SECTOR];
    for (int i = 0; i < ns; ++i) {
        circshift(sc2, SC_RING, space[i], shifted);
        const double d = similarity(sc1, shifted);
        if (d < best) { arg = space[i]; best = d; }
    }
    *dist = best;
    *shift = arg;
}
