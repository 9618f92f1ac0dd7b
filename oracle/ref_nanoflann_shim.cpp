// oracle/ref_nanoflann_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin C entry points around the REFERENCE's own vendored nanoflann.hpp,
// compiled from where it lies under /root/reference (never copied into this
// repo) into oracle/_ref/libref_nanoflann.so by oracle/Makefile.  It pins the
// k-NN stage of the oracle (and of the HIP path) to the reference's actual
// index: KDTreeSingleIndexAdaptor over float xyz with f64 distances, exactly
// as PointCloudKdtree<pt_t, double> instantiates it
// (third_parties/nanoflann/include/nanoflann/pcl_adaptor.hpp:33-34,47-58;
// the adaptor there needs PCL, so this file supplies a raw-array data source
// with the same three members).  Also used as the "reference" flavour of the
// CPU baseline's kd-tree (build per call + 5-NN), bench.py cpu_baseline.
#include <nanoflann.hpp>
#include <cstddef>
#include <cstdint>

namespace {
struct RawCloud {
    const float* p; size_t n; size_t stride;
    size_t kdtree_get_point_count() const { return n; }
    double kdtree_get_pt(size_t i, int d) const { return p[i * stride + d]; }
    template <class B> bool kdtree_get_bbox(B&) const { return false; }
};
using Metric = nanoflann::metric_L2_Simple::traits<double, RawCloud>::distance_t;
using Tree = nanoflann::KDTreeSingleIndexAdaptor<Metric, RawCloud, 3, size_t>;
struct Index { RawCloud cloud; Tree tree; Index(const float* p, size_t n, size_t s) : cloud{p, n, s}, tree(3, cloud) { tree.buildIndex(); } };
}

extern "C" {
void* ref_kd_build(const float* pts, size_t n, size_t stride_floats) { return new Index(pts, n, stride_floats); }
void ref_kd_free(void* h) { delete static_cast<Index*>(h); }
// k nearest of one query (f64 coordinates), ascending; returns count found
int ref_kd_knn(void* h, const double* q, int k, size_t* idx, double* d2)
{
    nanoflann::KNNResultSet<double> rs(k);
    rs.init(idx, d2);
    static_cast<Index*>(h)->tree.findNeighbors(rs, q);
    return (int)rs.size();
}
// batch: queries are float xyz (stride floats), converted to f64 like LoamRegister.cpp:54-55
void ref_kd_knn_batch(void* h, const float* q, size_t nq, size_t stride_floats, int k, int64_t* idx, double* d2)
{
    Index* ix = static_cast<Index*>(h);
    std::vector<size_t> ii(k);
    for (size_t i = 0; i < nq; ++i) {
        double qq[3] = {q[i * stride_floats], q[i * stride_floats + 1], q[i * stride_floats + 2]};
        nanoflann::KNNResultSet<double> rs(k);
        rs.init(ii.data(), d2 + i * k);
        ix->tree.findNeighbors(rs, qq);
        for (int j = 0; j < k; ++j) idx[i * k + j] = j < (int)rs.size() ? (int64_t)ii[j] : -1;
    }
}
}
