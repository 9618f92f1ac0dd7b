// pcp/pcd_io.hpp -- PCD files in and out, without PCL: what test/loc.cpp and test/align.cpp feed the registration path with.
//
// Mirrors (reference): pcl::io::loadPCDFile<pcl::PointXYZI> as called at frontend/src/MapManager.cpp:68 (the map named by
// cfg["pcd_file"]), test/align.cpp:97-107 (target and source clouds) and common/pcp/pcp.hpp (savePCDFile).  The PCD format
// itself is PCL's (absent from the reference tree; restated from its published v0.7 layout): a text header
//   VERSION / FIELDS / SIZE / TYPE / COUNT / WIDTH / HEIGHT / VIEWPOINT / POINTS / DATA ascii|binary|binary_compressed
// followed by one record per point.  Read: the fields named x, y, z (required) and intensity (optional, 0 when absent -- PCL
// warns and leaves the default) of any scalar TYPE/SIZE, in any order, other fields skipped ("_" padding of PCL's own binary
// files included); binary_compressed is PCL's LZF stream holding the fields one after the other.  Written: x y z intensity as
// F 4, ascii or binary.  Errors throw std::runtime_error; loadPCDFile returns -1 where PCL's returns -1 (file cannot be opened).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../PCR/HipRegister.hpp"

namespace pcp {

namespace detail {

struct PcdField { std::string name; int size = 4; char type = 'F'; int count = 1; size_t offset = 0; };

inline double pcd_scalar(const unsigned char* p, const PcdField& f) {
    switch (f.type) {
        case 'F': if (f.size == 4) { float v; std::memcpy(&v, p, 4); return v; } if (f.size == 8) { double v; std::memcpy(&v, p, 8); return v; } break;
        case 'U': if (f.size == 1) return *p; if (f.size == 2) { uint16_t v; std::memcpy(&v, p, 2); return v; } if (f.size == 4) { uint32_t v; std::memcpy(&v, p, 4); return v; }
                  if (f.size == 8) { uint64_t v; std::memcpy(&v, p, 8); return (double)v; } break;
        case 'I': if (f.size == 1) return (int8_t)*p; if (f.size == 2) { int16_t v; std::memcpy(&v, p, 2); return v; } if (f.size == 4) { int32_t v; std::memcpy(&v, p, 4); return v; }
                  if (f.size == 8) { int64_t v; std::memcpy(&v, p, 8); return (double)v; } break;
    }
    throw std::runtime_error("pcd: unsupported field type " + std::string(1, f.type) + std::to_string(f.size));
}

// LZF (Marc Lehmann's format, what PCL's lzfDecompress reads): literal runs and back references
inline size_t lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        unsigned ctrl = in[ip++];
        if (ctrl < 32) {                               // literal run of ctrl + 1 bytes
            ++ctrl;
            if (op + ctrl > out_len || ip + ctrl > in_len) return 0;
            std::memcpy(out + op, in + ip, ctrl);
            op += ctrl; ip += ctrl;
        } else {                                       // back reference
            size_t len = ctrl >> 5;
            if (ip >= in_len) return 0;
            size_t ref_off = ((size_t)(ctrl & 0x1f) << 8) + 1;
            if (len == 7) { len += in[ip++]; if (ip >= in_len) return 0; }
            ref_off += in[ip++];
            len += 2;
            if (ref_off > op || op + len > out_len) return 0;
            size_t ref = op - ref_off;
            for (size_t k = 0; k < len; ++k) out[op++] = out[ref++];      // (may overlap: byte by byte)
        }
    }
    return op;
}

}  // namespace detail

// pcl::io::loadPCDFile<pcl::PointXYZI>(file, cloud): 0 on success, -1 when the file cannot be opened; a malformed file throws.
inline int loadPCDFile(const std::string& path, PCR::PointCloud& cloud) {
    using detail::PcdField;
    std::ifstream f(path, std::ios::binary);
    if (!f) return -1;
    std::vector<PcdField> fields;
    size_t width = 0, height = 1, points = 0;
    bool have_points = false;
    std::string data_kind, line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ls(line);
        std::string key;
        ls >> key;
        std::vector<std::string> tok;
        for (std::string t; ls >> t;) tok.push_back(t);
        if (key == "VERSION" || key == "VIEWPOINT") continue;
        if (key == "FIELDS" || key == "COLUMNS") { fields.resize(tok.size()); for (size_t i = 0; i < tok.size(); ++i) fields[i].name = tok[i]; }
        else if (key == "SIZE") { if (tok.size() != fields.size()) throw std::runtime_error("pcd: SIZE does not match FIELDS"); for (size_t i = 0; i < tok.size(); ++i) fields[i].size = std::stoi(tok[i]); }
        else if (key == "TYPE") { if (tok.size() != fields.size()) throw std::runtime_error("pcd: TYPE does not match FIELDS"); for (size_t i = 0; i < tok.size(); ++i) fields[i].type = tok[i].empty() ? 'F' : tok[i][0]; }
        else if (key == "COUNT") { if (tok.size() != fields.size()) throw std::runtime_error("pcd: COUNT does not match FIELDS"); for (size_t i = 0; i < tok.size(); ++i) fields[i].count = std::stoi(tok[i]); }
        else if (key == "WIDTH") { if (tok.empty()) throw std::runtime_error("pcd: WIDTH without a value"); width = std::stoull(tok[0]); }
        else if (key == "HEIGHT") { if (tok.empty()) throw std::runtime_error("pcd: HEIGHT without a value"); height = std::stoull(tok[0]); }
        else if (key == "POINTS") { if (tok.empty()) throw std::runtime_error("pcd: POINTS without a value"); points = std::stoull(tok[0]); have_points = true; }
        else if (key == "DATA") { if (tok.empty()) throw std::runtime_error("pcd: DATA without a kind"); data_kind = tok[0]; break; }
        else throw std::runtime_error("pcd: unknown header entry " + key);
    }
    if (data_kind.empty()) throw std::runtime_error("pcd: no DATA line in " + path);
    if (!have_points) points = width * height;
    size_t stride = 0;
    int ix = -1, iy = -1, iz = -1, ii = -1;
    for (size_t i = 0; i < fields.size(); ++i) {
        if (fields[i].size <= 0 || fields[i].count < 0) throw std::runtime_error("pcd: bad SIZE / COUNT");
        fields[i].offset = stride;
        stride += (size_t)fields[i].size * (size_t)fields[i].count;
        if (fields[i].name == "x") ix = (int)i; else if (fields[i].name == "y") iy = (int)i; else if (fields[i].name == "z") iz = (int)i;
        else if (fields[i].name == "intensity") ii = (int)i;
    }
    if (ix < 0 || iy < 0 || iz < 0) throw std::runtime_error("pcd: fields x, y, z are required (" + path + ")");
    for (int k : {ix, iy, iz, ii})      // a field that is READ must hold at least one element (COUNT 0 would read past the record)
        if (k >= 0 && fields[(size_t)k].count < 1) throw std::runtime_error("pcd: COUNT of field " + fields[(size_t)k].name + " must be at least 1 (" + path + ")");
    cloud.points.clear();
    cloud.points.resize(points);
    if (data_kind == "ascii") {
        for (size_t p = 0; p < points; ++p) {
            if (!std::getline(f, line)) throw std::runtime_error("pcd: " + path + " ends after " + std::to_string(p) + " of " + std::to_string(points) + " points");
            std::istringstream ls(line);
            PCR::PointXYZI& q = cloud.points[p];
            for (size_t i = 0; i < fields.size(); ++i)
                for (int c = 0; c < fields[i].count; ++c) {
                    std::string t;
                    if (!(ls >> t)) throw std::runtime_error("pcd: short record at point " + std::to_string(p));
                    if (c) continue;
                    const float v = (t == "nan" || t == "-nan" || t == "NaN") ? std::nanf("") : std::strtof(t.c_str(), nullptr);
                    if ((int)i == ix) q.x = v; else if ((int)i == iy) q.y = v; else if ((int)i == iz) q.z = v; else if ((int)i == ii) q.intensity = v;
                }
        }
        return 0;
    }
    std::vector<unsigned char> raw;
    bool soa = false;
    if (data_kind == "binary") {
        raw.resize(points * stride);
        f.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)raw.size());
        if ((size_t)f.gcount() != raw.size()) throw std::runtime_error("pcd: " + path + " holds fewer than " + std::to_string(points) + " binary records");
    } else if (data_kind == "binary_compressed") {
        uint32_t csize = 0, usize = 0;
        f.read(reinterpret_cast<char*>(&csize), 4); f.read(reinterpret_cast<char*>(&usize), 4);
        if (!f || (size_t)usize != points * stride) throw std::runtime_error("pcd: bad compressed block sizes in " + path);
        std::vector<unsigned char> comp(csize);
        f.read(reinterpret_cast<char*>(comp.data()), csize);
        if ((size_t)f.gcount() != csize) throw std::runtime_error("pcd: truncated compressed block in " + path);
        raw.resize(usize);
        if (detail::lzf_decompress(comp.data(), csize, raw.data(), usize) != usize) throw std::runtime_error("pcd: LZF stream of " + path + " is corrupt");
        soa = true;      // the fields lie one after the other: all x, then all y, ...
    } else throw std::runtime_error("pcd: unknown DATA kind " + data_kind);
    auto at = [&](size_t p, int fi) -> const unsigned char* {
        const PcdField& fd = fields[(size_t)fi];
        return soa ? raw.data() + fd.offset * points + p * (size_t)fd.size * (size_t)fd.count : raw.data() + p * stride + fd.offset;
    };
    for (size_t p = 0; p < points; ++p) {
        PCR::PointXYZI& q = cloud.points[p];
        q.x = (float)detail::pcd_scalar(at(p, ix), fields[(size_t)ix]);
        q.y = (float)detail::pcd_scalar(at(p, iy), fields[(size_t)iy]);
        q.z = (float)detail::pcd_scalar(at(p, iz), fields[(size_t)iz]);
        if (ii >= 0) q.intensity = (float)detail::pcd_scalar(at(p, ii), fields[(size_t)ii]);
    }
    return 0;
}

// pcl::io::savePCDFileBinary / savePCDFileASCII for PointXYZI clouds
inline int savePCDFile(const std::string& path, const PCR::PointCloud& cloud, bool binary = true) {
    std::ofstream f(path, std::ios::binary);
    if (!f) return -1;
    f << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
      << "WIDTH " << cloud.size() << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << cloud.size() << "\nDATA " << (binary ? "binary" : "ascii") << "\n";
    for (const PCR::PointXYZI& p : cloud.points) {
        if (binary) { const float r[4] = {p.x, p.y, p.z, p.intensity}; f.write(reinterpret_cast<const char*>(r), sizeof r); }
        else { char buf[96]; std::snprintf(buf, sizeof buf, "%.9g %.9g %.9g %.9g\n", p.x, p.y, p.z, p.intensity); f << buf; }
    }
    return f ? 0 : -1;
}

}  // namespace pcp
