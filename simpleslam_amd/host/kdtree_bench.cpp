// kdtree_bench.cpp -- counterpart of the reference's test/benchmark/kdtree.cpp:58-127 (Google Benchmark of PCL-FLANN vs nanoflann:
// index build, k = 5 and radius queries) for the spatial index of this library:
//   kdtree_bench <map.f32> [queries.f32] [repeats]
// prints the build time of the uniform-grid index over the map (seconds, device timeline) and the time per exact 5-NN query
// (ns/query; the reference benchmarks one query at the origin in a loop, a GPU answers a whole cloud of queries per call: the map's
// own points, or the given query cloud).  The reference's nanoflann on the same map is timed by bench.py's cpu_baseline leg
// ("index": "nanoflann(_ref)") -- this program never touches the checker.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "PCR/HipRegister.hpp"

static std::vector<float> load(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    const size_t bytes = (size_t)f.tellg();
    std::vector<float> v(bytes / 4);
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(v.size() * 4));
    return v;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <map.f32> [queries.f32] [repeats]\n", argv[0]); return 2; }
    try {
        const std::vector<float> map = load(argv[1]);
        const std::vector<float> qry = argc > 2 ? load(argv[2]) : map;
        const int reps = argc > 3 ? std::atoi(argv[3]) : 20;
        const size_t n_map = map.size() / 4, n_q = qry.size() / 4;
        pcr_handle* h = pcr_create("loam", nullptr);
        if (!h) throw std::runtime_error(pcr_last_error(nullptr));
        double build_s = 1e30;
        for (int r = 0; r < reps; ++r) {                          // host clock around the call: upload + index, then device-resident rebuilds
            const auto t0 = std::chrono::steady_clock::now();
            if (pcr_set_target(h, map.data(), n_map, 16, 0)) throw std::runtime_error(pcr_last_error(h));
            build_s = std::min(build_s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        const double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        double JtJ[36], JtE[6];
        int64_t n_acc = 0;
        std::vector<int32_t> nn(n_q * 5);
        std::vector<int8_t> status(n_q);
        double query_s = 1e30;
        for (int r = 0; r < reps; ++r) {
            const auto t0 = std::chrono::steady_clock::now();
            if (pcr_loam_linearize(h, qry.data(), n_q, 16, 0, I, JtJ, JtE, &n_acc, status.data(), nullptr, nn.data())) throw std::runtime_error(pcr_last_error(h));
            query_s = std::min(query_s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        size_t found = 0;
        for (size_t i = 0; i < n_q; ++i) found += status[i] != 1;
        std::printf("grid index: %zu points, build (host copy + index, best of %d) %.6f s\n", n_map, reps, build_s);
        std::printf("k = 5 exact queries inside the 1 m gate (+ plane fit and row, one linearisation): %zu queries, %.1f ns/query, %zu with 5 neighbours\n",
                    n_q, 1e9 * query_s / (double)n_q, found);
        pcr_destroy(h);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
