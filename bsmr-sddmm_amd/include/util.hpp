#pragma once
// Small string / path helpers with the behaviour of the reference's
// include/util.hpp (tokenizer :182-197, to_trimmed_string :141-155, path
// helpers :157-180).  Own implementation.

#include <cstddef>
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

}  // namespace util
