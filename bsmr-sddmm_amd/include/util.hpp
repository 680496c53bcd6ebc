#pragma once
// Small string / path helpers with the behaviour of the reference's
// include/util.hpp (tokenizer :182-197, to_trimmed_string :141-155, path
// helpers :157-180).  Own implementation.

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <iomanip>
#include <sstream>
#include <string>

namespace util {

// One "word" of `line` starting at `pos`; a word ends at ' ', '\t' or '\r'.
// Afterwards `pos` stands on the first character of the next word.  A leading
// separator therefore produces an empty word, as in the reference.
inline std::string iterateOneWordFromLine(const std::string& line, int& pos) {
    const int n = static_cast<int>(line.size());
    auto is_sep = [](char c) { return c == ' ' || c == '\t' || c == '\r'; };
    const int first = pos;
    while (pos < n && !is_sep(line[pos])) ++pos;
    const int last = pos;
    while (pos < n && is_sep(line[pos])) ++pos;
    return last > first ? line.substr(first, last - first) : std::string();
}

// Fixed notation with `precision` digits, then trailing zeros (and a dangling
// '.') removed: 0.3f -> "0.3", 1.1f -> "1.1", 32 -> "32", 0.0f -> "0".
template <typename T>
inline std::string to_trimmed_string(T value, int precision = 6) {
    std::ostringstream os;
    os << std::fixed << std::setprecision(precision) << value;
    std::string s = os.str();
    if (s.find('.') != std::string::npos) {
        s.erase(s.find_last_not_of('0') + 1);
        if (!s.empty() && s.back() == '.') s.pop_back();
    }
    return s;
}

inline std::string getParentFolderPath(const std::string& path) {
    const size_t cut = path.find_last_of("/\\");
    return cut == std::string::npos ? std::string() : path.substr(0, cut + 1);
}

inline std::string getFileName(const std::string& path) {
    const size_t cut = path.find_last_of("/\\");
    return cut == std::string::npos ? path : path.substr(cut + 1);
}

inline std::string getFileSuffix(const std::string& filename) {
    const size_t dot = filename.find_last_of('.');
    return dot == std::string::npos ? std::string() : filename.substr(dot);
}

// OpenMP threads the host pipeline should use: omp_get_max_threads() capped by the CPU quota of
// the control group (a container that sees 256 CPUs but may use 16 runs 256 spinning threads
// several times slower than 16) and by BSMR_HOST_THREADS.  Call with omp_get_max_threads().
inline int hostThreads(int ompMax) {
    static int cached = 0;
    if (cached > 0) return cached < ompMax ? cached : ompMax;
    long limit = ompMax;
    if (const char* env = std::getenv("BSMR_HOST_THREADS")) {
        const long v = std::strtol(env, nullptr, 10);
        if (v > 0) limit = v;
    } else if (!std::getenv("OMP_NUM_THREADS")) {
        long long quota = -1, period = -1;
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            if (std::fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = std::strtoll(q, nullptr, 10);
            std::fclose(f);
        } else {
            if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
                std::fclose(g);
            }
            if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(g, "%lld", &period) != 1) period = -1;
                std::fclose(g);
            }
        }
        if (quota > 0 && period > 0) {
            const long cpus = static_cast<long>((quota + period - 1) / period);
            if (cpus < limit) limit = cpus;
        }
    }
    cached = static_cast<int>(limit < 1 ? 1 : limit);
    return cached < ompMax ? cached : ompMax;
}

}  // namespace util
