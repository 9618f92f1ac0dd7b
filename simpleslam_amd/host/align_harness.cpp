// align_harness.cpp -- ROS/PCL-free counterpart of the reference's test/align.cpp:
//   align_harness <target.pcd|.f32> <source.pcd|.f32> <loam|ndt|vgicp> [init_pose.txt]
// loads two clouds (PCD files as align.cpp:97-107 does -- pcp/pcd_io.hpp -- or raw float32 x y z intensity records), reads the optional initial pose (4x4 row-major text,
// align.cpp:85-93), voxel-filters BOTH clouds at 0.1 m (align.cpp:128-129), runs ONE scan2Map through the plugin mirror
// (align.cpp:144), and prints what align.cpp logs: the cloud sizes before and after the filter, the elapsed time, the gated
// fitness score (mean squared nearest-neighbour distance over the points within 1 m, align.cpp:29-61) and the final 4x4.
// (The PCL visualiser at the end of align.cpp has no counterpart.)
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>

#include "PCR/HipRegister.hpp"
#include "pcp/pcd_io.hpp"

static PCR::PC_Ptr load(const char* path) {
    const std::string name = path;
    if (name.size() >= 4 && name.compare(name.size() - 4, 4, ".pcd") == 0) {
        auto pc = std::make_shared<PCR::PointCloud>();
        if (pcp::loadPCDFile(name, *pc) == -1) throw std::runtime_error(std::string("failed to load: ") + path);      // align.cpp:98-101
        return pc;
    }
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(std::string("failed to load: ") + path);
    auto pc = std::make_shared<PCR::PointCloud>();
    float r[4];
    while (f.read(reinterpret_cast<char*>(r), sizeof r)) {
        PCR::PointXYZI p;
        p.x = r[0]; p.y = r[1]; p.z = r[2]; p.intensity = r[3];
        pc->points.push_back(p);
    }
    return pc;
}

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: align target.f32 source.f32 method [init_pose.txt]\n"); return 0; }      // align.cpp:66-69
    try {
        PCR::pose_t before_pose, init_pose;
        if (argc > 4) {
            std::ifstream inf(argv[4]);
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (!(inf >> before_pose(i, j))) throw std::runtime_error("bad init_pose.txt");
        }
        init_pose = before_pose;
        auto target_cloud = load(argv[1]);
        auto source_cloud = load(argv[2]);
        const std::string method = argv[3];
        PCR::PointCloudRegister::Ptr pcr;
        if (method == "loam") pcr = std::make_shared<PCR::LoamRegister>();
        else if (method == "ndt") pcr = std::make_shared<PCR::NdtRegister>();
        else if (method == "vgicp") pcr = std::make_shared<PCR::VgicpRegister>();
        else { std::fprintf(stderr, "no such method!!\n"); return -1; }                                                   // align.cpp:118-121
        std::printf("target cloud size: %zu\nsource cloud size: %zu\n", target_cloud->size(), source_cloud->size());
        auto t0 = std::chrono::steady_clock::now();
        pcp::voxelDownSample(target_cloud, 0.1f);
        pcp::voxelDownSample(source_cloud, 0.1f);
        std::printf("downsample pc elapsed %.6fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        std::printf("--------- after downsample ---------\ntarget cloud size: %zu\nsource cloud size: %zu\n", target_cloud->size(), source_cloud->size());
        t0 = std::chrono::steady_clock::now();
        const bool conv = pcr->scan2Map(source_cloud, target_cloud, init_pose);
        if (!conv) std::printf("not converge!!\n");
        std::printf("scan to map elapsed %.3fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        int64_t n_in = 0;
        const double fit = dynamic_cast<PCR::HipRegister*>(pcr.get())->gatedFitness(source_cloud, init_pose, 1.0, &n_in);
        std::printf("get fitness score: %.9g (%lld points within 1 m)\n", fit, (long long)n_in);
        std::printf("trans:\n");
        for (int r = 0; r < 4; ++r) std::printf("%.17g %.17g %.17g %.17g\n", init_pose(r, 0), init_pose(r, 1), init_pose(r, 2), init_pose(r, 3));
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
