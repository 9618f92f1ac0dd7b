// io_check.cpp -- what the harness's readers make of a configuration file or a PCD file, as text (no GPU call):
//   io_check --params <params.json>   -> cores, downSampleVoxelGridSize, pcd_file, frontend.pcr (the keys test/loc.cpp and the
//                                        registration path read: config/params.hpp, config/params.json:5,8,10,58)
//   io_check --pcd <file.pcd> [n]     -> number of points, then the first n points (x y z intensity, %.9g)
//   io_check --repack <in.pcd> <out.pcd> <ascii|binary>   -> read, then write with pcp::savePCDFile
// Exit code 1 with the reader's message on stderr when a file is malformed; 4 when a PCD cannot be opened (loadPCDFile's -1).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "config/params.hpp"
#include "pcp/pcd_io.hpp"

int main(int argc, char** argv) {
    try {
        const std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "--params" && argc > 2) {
            config::Params::load(argv[2]);
            auto cfg = config::Params::getInstance();
            const std::string pcd_file = cfg["pcd_file"];
            std::printf("cores %d\ndownSampleVoxelGridSize %.9g\npcd_file %s\nfrontend.pcr %s\n", cfg["cores"].get<int>(),
                        (double)cfg["downSampleVoxelGridSize"].get<float>(), pcd_file.c_str(), cfg["frontend"]["pcr"].get<std::string>().c_str());
            return 0;
        }
        if (mode == "--pcd" && argc > 2) {
            PCR::PointCloud pc;
            if (pcp::loadPCDFile(argv[2], pc) == -1) { std::fprintf(stderr, "cannot open %s\n", argv[2]); return 4; }
            const size_t n = argc > 3 ? (size_t)std::atoll(argv[3]) : pc.size();
            std::printf("points %zu\n", pc.size());
            for (size_t i = 0; i < pc.size() && i < n; ++i) std::printf("%.9g %.9g %.9g %.9g\n", pc.points[i].x, pc.points[i].y, pc.points[i].z, pc.points[i].intensity);
            return 0;
        }
        if (mode == "--repack" && argc > 4) {
            PCR::PointCloud pc;
            if (pcp::loadPCDFile(argv[2], pc) == -1) { std::fprintf(stderr, "cannot open %s\n", argv[2]); return 4; }
            return pcp::savePCDFile(argv[3], pc, std::string(argv[4]) == "binary") == 0 ? 0 : 1;
        }
        std::fprintf(stderr, "usage: io_check --params <params.json> | --pcd <file.pcd> [n] | --repack <in.pcd> <out.pcd> <ascii|binary>\n");
        return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
