// config/params.hpp -- the configuration the registration path and its harness read, without nlohmann/json.
//
// Mirrors (reference): config/params.hpp:38-46 -- `config::Params::getInstance()` hands out the parsed params.json, which is
// JSON WITH COMMENTS (`json::parse(inf, nullptr, true, true)`, params.hpp:30) -- and the keys the path and test/loc.cpp read:
//   cfg["cores"]                      PCR/include/PCR/PointCloudRegister.hpp:28-32   (config/params.json:5)
//   cfg["downSampleVoxelGridSize"]    frontend/src/LidarOdometry.cpp:33, MapManager.cpp:57   (params.json:8)
//   cfg["pcd_file"]                   test/loc.cpp:34, frontend/src/MapManager.cpp:68        (params.json:10)
//   cfg["frontend"]["pcr"]            frontend/src/LidarOdometry.cpp:32                      (params.json:58)
// The reference bakes the file name in at compile time (CONFIG_FILE); here the harness names it: Params::load(path) once, then
// getInstance() anywhere, with the same `cfg["a"]["b"].get<T>()` access.  A small recursive-descent reader: objects, arrays,
// strings (with the usual escapes), numbers, true / false / null, `//` and `/* */` comments.  Errors throw std::runtime_error
// with line and column, as a failed json::parse throws.
#pragma once
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace config {

class json {
public:
    enum Kind { Null, Bool, Number, String, Array, Object };

private:
    Kind kind_ = Null;
    bool b_ = false;
    double num_ = 0.0;
    std::string str_;
    std::vector<json> arr_;
    std::map<std::string, json> obj_;

    struct Reader {
        const std::string& s;
        size_t i = 0;
        [[noreturn]] void fail(const std::string& what) const {
            size_t line = 1, col = 1;
            for (size_t k = 0; k < i && k < s.size(); ++k) { if (s[k] == '\n') { ++line; col = 1; } else ++col; }
            throw std::runtime_error("params: " + what + " at line " + std::to_string(line) + ", column " + std::to_string(col));
        }
        void skip() {      // white space and comments
            for (;;) {
                while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i;
                if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '/') { while (i < s.size() && s[i] != '\n') ++i; continue; }
                if (i + 1 < s.size() && s[i] == '/' && s[i + 1] == '*') {
                    const size_t e = s.find("*/", i + 2);
                    if (e == std::string::npos) fail("unterminated comment");
                    i = e + 2;
                    continue;
                }
                return;
            }
        }
        std::string string() {
            if (i >= s.size()) fail("unexpected end of input");
            if (s[i] != '"') fail("expected a string");
            ++i;
            std::string out;
            while (i < s.size() && s[i] != '"') {
                char c = s[i++];
                if (c == '\\') {
                    if (i >= s.size()) fail("unterminated escape");
                    const char e = s[i++];
                    switch (e) {
                        case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
                        case 'b': out += '\b'; break; case 'f': out += '\f'; break; case 'n': out += '\n'; break;
                        case 'r': out += '\r'; break; case 't': out += '\t'; break;
                        case 'u': {
                            if (i + 4 > s.size()) fail("short \\u escape");
                            const unsigned cp = (unsigned)std::strtoul(s.substr(i, 4).c_str(), nullptr, 16);
                            i += 4;
                            if (cp < 0x80) out += (char)cp;
                            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                            break;
                        }
                        default: fail("bad escape");
                    }
                } else out += c;
            }
            if (i >= s.size()) fail("unterminated string");
            ++i;
            return out;
        }
        json value() {
            skip();
            if (i >= s.size()) fail("unexpected end of input");
            json v;
            const char c = s[i];
            if (c == '{') {
                ++i; v.kind_ = Object;
                skip();
                if (i < s.size() && s[i] == '}') { ++i; return v; }
                for (;;) {
                    skip();
                    std::string key = string();
                    skip();
                    if (i >= s.size() || s[i] != ':') fail("expected ':'");
                    ++i;
                    v.obj_[key] = value();
                    skip();
                    if (i < s.size() && s[i] == ',') { ++i; continue; }
                    if (i < s.size() && s[i] == '}') { ++i; return v; }
                    fail("expected ',' or '}'");
                }
            }
            if (c == '[') {
                ++i; v.kind_ = Array;
                skip();
                if (i < s.size() && s[i] == ']') { ++i; return v; }
                for (;;) {
                    v.arr_.push_back(value());
                    skip();
                    if (i < s.size() && s[i] == ',') { ++i; continue; }
                    if (i < s.size() && s[i] == ']') { ++i; return v; }
                    fail("expected ',' or ']'");
                }
            }
            if (c == '"') { v.kind_ = String; v.str_ = string(); return v; }
            if (s.compare(i, 4, "true") == 0) { i += 4; v.kind_ = Bool; v.b_ = true; return v; }
            if (s.compare(i, 5, "false") == 0) { i += 5; v.kind_ = Bool; v.b_ = false; return v; }
            if (s.compare(i, 4, "null") == 0) { i += 4; return v; }
            if (c == '-' || (c >= '0' && c <= '9')) {
                const char* b = s.c_str() + i;
                char* e = nullptr;
                v.num_ = std::strtod(b, &e);
                if (e == b) fail("bad number");
                i += (size_t)(e - b);
                v.kind_ = Number;
                return v;
            }
            fail(std::string("unexpected character '") + c + "'");
        }
    };

public:
    static json parse(const std::string& text) {
        Reader r{text};
        json v = r.value();
        r.skip();
        if (r.i != text.size()) r.fail("trailing characters");
        return v;
    }
    Kind kind() const { return kind_; }
    bool contains(const std::string& key) const { return kind_ == Object && obj_.count(key) != 0; }
    // cfg["key"]: a missing key throws, as nlohmann's const operator[] / at() on a const json would
    const json& operator[](const std::string& key) const {
        if (kind_ != Object) throw std::runtime_error("params: [\"" + key + "\"] on a value that is not an object");
        auto it = obj_.find(key);
        if (it == obj_.end()) throw std::runtime_error("params: key \"" + key + "\" is missing");
        return it->second;
    }
    const json& operator[](size_t idx) const {
        if (kind_ != Array || idx >= arr_.size()) throw std::runtime_error("params: array index out of range");
        return arr_[idx];
    }
    size_t size() const { return kind_ == Array ? arr_.size() : (kind_ == Object ? obj_.size() : 0); }
    template <class T> T get() const {
        if constexpr (std::is_same<T, std::string>::value) {
            if (kind_ != String) throw std::runtime_error("params: value is not a string");
            return str_;
        } else if constexpr (std::is_same<T, bool>::value) {
            if (kind_ != Bool) throw std::runtime_error("params: value is not a boolean");
            return b_;
        } else {
            static_assert(std::is_arithmetic<T>::value, "get<T>: string, bool or a number");
            if (kind_ != Number) throw std::runtime_error("params: value is not a number");
            return static_cast<T>(num_);
        }
    }
    operator std::string() const { return get<std::string>(); }      // `string pcd_file = cfg["pcd_file"];` (test/loc.cpp:34)
};

class Params {
    json jps_;
    static std::shared_ptr<Params>& slot() { static std::shared_ptr<Params> p; return p; }
public:
    // replaces the reference's compile-time CONFIG_FILE
    static void load(const std::string& path) {
        std::ifstream inf(path);
        if (!inf) throw std::runtime_error("params: cannot open " + path);
        std::stringstream ss;
        ss << inf.rdbuf();
        auto p = std::make_shared<Params>();
        p->jps_ = json::parse(ss.str());
        slot() = p;
    }
    static void loadText(const std::string& text) { auto p = std::make_shared<Params>(); p->jps_ = json::parse(text); slot() = p; }
    static bool loaded() { return (bool)slot(); }
    static json getInstance() {
        if (!slot()) throw std::runtime_error("params: no configuration loaded (config::Params::load)");
        return slot()->jps_;      // a copy, like the reference's get()
    }
};

}  // namespace config
