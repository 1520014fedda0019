// yaml_lite.h -- reads the two-level `group:\n  key_: value` files the reference loads through
// roslaunch <rosparam file=...> (launch/run_semantickitti.launch:6) into "group/key_" -> text.
// Accepts config/semantickitti.yaml and config/parkinglot.yaml verbatim (comments, quoted strings,
// flow sequences spanning lines).  Not a general YAML parser.
#ifndef SCVOD_YAML_LITE_H_
#define SCVOD_YAML_LITE_H_
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
namespace scvod_host {
class YamlLite {
  public:
    bool load(const std::string& path) {
        std::ifstream in(path);
        if (!in) return false;
        std::string line, group, open_key;
        while (std::getline(in, line)) {
            std::string s = strip_comment(line);
            if (trim(s).empty()) continue;
            if (!open_key.empty()) {  // continuation of a [ ... ] sequence
                kv_[open_key] += " " + trim(s);
                if (s.find(']') != std::string::npos) open_key.clear();
                continue;
            }
            size_t indent = s.find_first_not_of(" \t");
            size_t colon = s.find(':');
            if (colon == std::string::npos) continue;
            std::string key = trim(s.substr(0, colon)), val = trim(s.substr(colon + 1));
            if (indent == 0) {
                group = key;
                continue;
            }
            std::string full = group + "/" + key;
            kv_[full] = unquote(val);
            if (val.find('[') != std::string::npos && val.find(']') == std::string::npos) open_key = full;
        }
        return true;
    }
    bool has(const std::string& k) const { return kv_.count(k) != 0; }
    template <typename T>
    void param(const std::string& k, T& out, const T& dflt) const {  // nh.param<T>(key, out, default)
        auto it = kv_.find(k);
        if (it == kv_.end()) {
            out = dflt;
            return;
        }
        convert(it->second, out);
    }
    std::vector<float> floats(const std::string& k) const {
        std::vector<float> v;
        auto it = kv_.find(k);
        if (it == kv_.end()) return v;
        std::string s = it->second;
        for (char& c : s)
            if (c == '[' || c == ']' || c == ',') c = ' ';
        std::istringstream is(s);
        double d;
        while (is >> d) v.push_back((float)d);
        return v;
    }

  private:
    static std::string trim(const std::string& s) {
        size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? "" : s.substr(a, b - a + 1);
    }
    static std::string strip_comment(const std::string& s) {
        bool q = false;
        for (size_t i = 0; i < s.size(); ++i) {
            if (s[i] == '"') q = !q;
            if (s[i] == '#' && !q) return s.substr(0, i);
        }
        return s;
    }
    static std::string unquote(const std::string& s) {
        if (s.size() >= 2 && s.front() == '"' && s.back() == '"') return s.substr(1, s.size() - 2);
        return s;
    }
    static void convert(const std::string& s, std::string& o) { o = s; }
    static void convert(const std::string& s, int& o) { o = std::atoi(s.c_str()); }
    static void convert(const std::string& s, float& o) { o = (float)std::atof(s.c_str()); }
    static void convert(const std::string& s, double& o) { o = std::atof(s.c_str()); }
    static void convert(const std::string& s, bool& o) { o = (s == "true" || s == "True" || s == "1"); }
    std::map<std::string, std::string> kv_;
};
}  // namespace scvod_host
#endif
