// Dense / sparse containers and the file loaders of the BSMR-SDDMM engine.
//
// Behavioural spec: reference src/Matrix.cpp (MatrixMarket loader :398-480,
// DLMC .smtx loader :296-371, SNAP edge-list loader :482-585, row-offset
// builder :236-250, accessors :176-234, makeData :117-138) as summarised in
// SURVEY.md appendix A.1/A.2.  Implementation is new: whole-file buffered
// parsing, sort-based duplicate detection (the reference inserts every entry
// into a std::set), size_t offsets.

#include <omp.h>

#include "Matrix.hpp"

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <numeric>
#include <random>
#include <unordered_map>

#include "npzReader.hpp"
#include "util.hpp"

// ---------------------------------------------------------------------------
// Matrix<T>
// ---------------------------------------------------------------------------
template <typename T>
Matrix<T>::Matrix(const sparseMatrix::COO<T>& s)
    : row_(s.row()), col_(s.col()), storageOrder_(row_major), leadingDimension_(s.col()),
      values_(static_cast<size_t>(s.row()) * s.col(), T(0)) {
    for (size_t i = 0; i < s.nnz(); ++i)
        values_[static_cast<size_t>(s.rowIndices()[i]) * leadingDimension_ + s.colIndices()[i]] =
            s.values()[i];
}

template <typename T>
bool Matrix<T>::initializeValue(const std::vector<T>& src) {
    if (src.size() != static_cast<size_t>(row_) * col_) {
        std::cerr << "Warning! Matrix value size mismatch" << std::endl;
        return false;
    }
    values_ = src;
    return true;
}

template <typename T>
void Matrix<T>::changeStorageOrder() {
    std::vector<T> t(values_.size());
    if (storageOrder_ == row_major) {
        for (size_t r = 0; r < row_; ++r)
            for (size_t c = 0; c < col_; ++c) t[c * row_ + r] = values_[r * col_ + c];
        storageOrder_ = col_major;
        leadingDimension_ = row_;
    } else {
        for (size_t c = 0; c < col_; ++c)
            for (size_t r = 0; r < row_; ++r) t[r * col_ + c] = values_[c * row_ + r];
        storageOrder_ = row_major;
        leadingDimension_ = col_;
    }
    values_.swap(t);
}

template <typename T>
UIN Matrix<T>::rowOfValueIndex(UIN idx) const {
    return storageOrder_ == row_major ? idx / leadingDimension_ : idx % leadingDimension_;
}

template <typename T>
UIN Matrix<T>::colOfValueIndex(UIN idx) const {
    return storageOrder_ == row_major ? idx % leadingDimension_ : idx / leadingDimension_;
}

template <typename T>
T Matrix<T>::getOneValue(UIN row, UIN col) const {
    const size_t ld = leadingDimension_;
    return storageOrder_ == row_major ? values_[row * ld + col] : values_[col * ld + row];
}

template <typename T>
T Matrix<T>::getOneValueForMultiplication(MatrixMultiplicationOrder order, UIN rowMtxC,
                                          UIN colMtxC, UIN k) const {
    // left operand: element (rowMtxC, k); right operand: element (k, colMtxC)
    return order == left_multiplication ? getOneValue(rowMtxC, k) : getOneValue(k, colMtxC);
}

template <typename T>
void Matrix<T>::makeData() {
    // Fixed seeds so that every run, the CLI and the benchmarks see the same
    // operands: 5489 (the std::mt19937 default) for row-major, 5490 for col-major.
    makeDataSeeded(storageOrder_ == row_major ? 5489u : 5490u);
}

template <typename T>
void Matrix<T>::makeData(UIN numRow, UIN numCol) {
    row_ = numRow;
    col_ = numCol;
    leadingDimension_ = storageOrder_ == row_major ? numCol : numRow;
    values_.assign(static_cast<size_t>(numRow) * numCol, T(0));
    makeData();
}

template <typename T>
void Matrix<T>::makeDataSeeded(uint32_t seed) {
    std::mt19937 gen(seed);
    for (auto& v : values_) {
        const uint32_t u = gen();
        v = static_cast<T>(2.0f * static_cast<float>(u >> 8) * (1.0f / 16777216.0f));
    }
}

template <typename T>
void Matrix<T>::print() const {
    for (UIN r = 0; r < row_; ++r) {
        for (UIN c = 0; c < col_; ++c) std::cout << getOneValue(r, c) << " ";
        std::cout << "\n";
    }
}

template <typename T>
std::vector<T> Matrix<T>::getRowVector(UIN row) const {
    std::vector<T> v(col_);
    for (UIN c = 0; c < col_; ++c) v[c] = getOneValue(row, c);
    return v;
}

template <typename T>
std::vector<T> Matrix<T>::getColVector(UIN col) const {
    std::vector<T> v(row_);
    for (UIN r = 0; r < row_; ++r) v[r] = getOneValue(r, col);
    return v;
}

// ---------------------------------------------------------------------------
// helpers shared by the loaders
// ---------------------------------------------------------------------------
void getCsrRowOffsets(const UIN row, const std::vector<UIN>& rowIndices,
                      std::vector<UIN>& rowOffsets) {
    rowOffsets.assign(static_cast<size_t>(row) + 1, 0);
    for (const UIN r : rowIndices) ++rowOffsets[static_cast<size_t>(r) + 1];
    for (size_t r = 0; r < row; ++r) rowOffsets[r + 1] += rowOffsets[r];
}

namespace {

bool readWholeFile(const std::string& file, std::string& out) {
    std::ifstream in(file, std::ios::in | std::ios::binary);
    if (!in.is_open()) {
        std::cerr << "Error, file cannot be opened : " << file << std::endl;
        return false;
    }
    in.seekg(0, std::ios::end);
    const std::streamoff n = in.tellg();
    in.seekg(0, std::ios::beg);
    out.resize(static_cast<size_t>(n));
    if (n > 0) in.read(&out[0], n);
    return true;
}

// Splits `buf` into lines exactly like std::getline: '\n' terminates a line, a
// final unterminated piece is a line, a final '\n' does not open an empty one.
struct LineReader {
    const std::string& buf;
    size_t pos = 0;
    explicit LineReader(const std::string& b) : buf(b) {}
    bool next(const char*& begin, const char*& end) {
        if (pos >= buf.size()) return false;
        const char* b = buf.data() + pos;
        const char* e = static_cast<const char*>(memchr(b, '\n', buf.size() - pos));
        if (!e) e = buf.data() + buf.size();
        begin = b;
        end = e;
        pos = static_cast<size_t>(e - buf.data()) + 1;
        return true;
    }
};

inline bool isSep(char c) { return c == ' ' || c == '\t' || c == '\r'; }

// Tokenizer with the semantics of util::iterateOneWordFromLine on [p, end).
inline void nextWord(const char*& p, const char* end, const char*& wb, const char*& we) {
    wb = p;
    while (p < end && !isSep(*p)) ++p;
    we = p;
    while (p < end && isSep(*p)) ++p;
}

// std::stoi semantics on a word: optional leading whitespace/sign, at least one
// digit, trailing garbage ignored.  Returns false where stoi would throw.
inline bool wordToInt(const char* wb, const char* we, long long& v) {
    if (wb == we) return false;
    char tmp[64];
    const size_t n = std::min<size_t>(static_cast<size_t>(we - wb), sizeof(tmp) - 1);
    memcpy(tmp, wb, n);
    tmp[n] = 0;
    char* endp = nullptr;
    errno = 0;
    v = strtoll(tmp, &endp, 10);
    if (endp == tmp) return false;
    if (errno == ERANGE || v > 2147483647LL || v < -2147483648LL) return false;
    return true;
}

// Value column: empty -> 0; std::stod otherwise; out-of-range -> 0 with a warning
// (reference src/Matrix.cpp:373-396).  Returns false where stod would throw
// invalid_argument.
inline bool wordToDouble(const char* wb, const char* we, double& v) {
    if (wb == we) { v = 0.0; return true; }
    char tmp[128];
    const size_t n = std::min<size_t>(static_cast<size_t>(we - wb), sizeof(tmp) - 1);
    memcpy(tmp, wb, n);
    tmp[n] = 0;
    char* endp = nullptr;
    errno = 0;
    v = strtod(tmp, &endp);
    if (endp == tmp) return false;
    if (errno == ERANGE && (v == HUGE_VAL || v == -HUGE_VAL)) {
        std::cout << "Warning: valueStr out of range: " << tmp << std::endl;
        v = 0.0;
    }
    return true;
}

// Stable counting sort of (col, val) by row; fills rowOffsets.  Equivalent to
// the reference's stable sort_by_key on row ids (src/Matrix.cpp:467-470):
// inside a row the entries keep their file order.
template <typename T>
void stableSortByRow(UIN rows, const std::vector<UIN>& ri, std::vector<UIN>& ci,
                     std::vector<T>& va, std::vector<UIN>& rowOffsets) {
    getCsrRowOffsets(rows, ri, rowOffsets);
    std::vector<UIN> cursor(rowOffsets.begin(), rowOffsets.end() - 1);
    std::vector<UIN> c2(ci.size());
    std::vector<T> v2(va.size());
    for (size_t i = 0; i < ri.size(); ++i) {
        const UIN dst = cursor[ri[i]]++;
        c2[dst] = ci[i];
        v2[dst] = va[i];
    }
    ci.swap(c2);
    va.swap(v2);
}

// true iff some row holds the same column twice.
bool hasDuplicateInRows(const std::vector<UIN>& rowOffsets, const std::vector<UIN>& ci,
                        UIN* badRow = nullptr, UIN* badCol = nullptr) {
    const long long rows = static_cast<long long>(rowOffsets.size()) - 1;
    bool dup = false;
#pragma omp parallel for schedule(dynamic, 256) num_threads(util::hostThreads(omp_get_max_threads()))
    for (long long r = 0; r < rows; ++r) {
        if (dup) continue;
        const UIN b = rowOffsets[r], e = rowOffsets[r + 1];
        if (e - b < 2) continue;
        std::vector<UIN> t(ci.begin() + b, ci.begin() + e);
        std::sort(t.begin(), t.end());
        const auto it = std::adjacent_find(t.begin(), t.end());
        if (it != t.end()) {
#pragma omp critical
            {
                dup = true;
                if (badRow) *badRow = static_cast<UIN>(r);
                if (badCol) *badCol = *it;
            }
        }
    }
    return dup;
}

}  // namespace

// ---------------------------------------------------------------------------
// CSR<T> loaders
// ---------------------------------------------------------------------------
template <typename T>
bool sparseMatrix::CSR<T>::initializeFromMatrixFile(const std::string& file) {
    const std::string suffix = util::getFileSuffix(file);
    if (suffix == ".mtx" || suffix == ".mmio") return initializeFromMtxFile(file);
    if (suffix == ".smtx") return initializeFromSmtxFile(file);
    if (suffix == ".txt") return initializeFromGraphDataset(file);
    if (suffix == ".npz") return initializeFromNpzFile(file);
    std::cerr << "Error, file format is not supported : " << file << std::endl;
    return false;
}

template <typename T>
bool sparseMatrix::CSR<T>::initializeFromMtxFile(const std::string& file) {
    std::string buf;
    if (!readWholeFile(file, buf)) return false;
    std::cout << "sparseMatrix::CSR initialize from file : " << file << std::endl;

    LineReader lr(buf);
    const char *lb = nullptr, *le = nullptr;
    bool haveHeader = false;
    while (lr.next(lb, le)) {  // leading '%' lines are comments; the banner is ignored
        if (lb < le && *lb == '%') continue;
        haveHeader = true;
        break;
    }
    if (!haveHeader || lb == le) {
        std::cerr << "Error, file " << file << " format is incorrect!" << std::endl;
        return false;
    }
    {
        const char* p = lb;
        const char *wb, *we;
        long long r = 0, c = 0;
        double n = 0;
        nextWord(p, le, wb, we);
        const bool okR = wordToInt(wb, we, r);
        nextWord(p, le, wb, we);
        const bool okC = wordToInt(wb, we, c);
        nextWord(p, le, wb, we);
        const bool okN = wordToDouble(wb, we, n);
        if (!okR || !okC || !okN || r < 0 || c < 0 || n < 0) {
            std::cerr << "Error, file " << file << " format is incorrect!" << std::endl;
            return false;
        }
        row_ = static_cast<UIN>(r);
        col_ = static_cast<UIN>(c);
        nnz_ = static_cast<UIN>(n);
    }

    std::vector<UIN> ri(nnz_), ci(nnz_);
    std::vector<T> va(nnz_);
    size_t idx = 0;
    while (lr.next(lb, le)) {
        if (lb == le) continue;  // empty lines are skipped
        const char* p = lb;
        const char *wb, *we;
        long long r = 0, c = 0;
        double v = 0;
        nextWord(p, le, wb, we);
        const bool okR = wordToInt(wb, we, r);
        nextWord(p, le, wb, we);
        const bool okC = wordToInt(wb, we, c);
        nextWord(p, le, wb, we);
        const bool okV = wordToDouble(wb, we, v);
        if (!okR || !okC || !okV) {
            std::cerr << "Error, file " << file << " has a malformed entry line!" << std::endl;
            return false;
        }
        if (idx >= nnz_) {
            std::cerr << "Error, file " << file << " too many elements, exceeding the number nnz!"
                      << std::endl;
            return false;
        }
        // 1-based -> 0-based in unsigned arithmetic: an index of 0 wraps to
        // 0xFFFFFFFF and is rejected by the range check below.
        ri[idx] = static_cast<UIN>(r) - 1u;
        ci[idx] = static_cast<UIN>(c) - 1u;
        va[idx] = static_cast<T>(v);
        ++idx;
    }
    if (idx < nnz_) {
        std::cerr << "Error, file " << file << " elements is not enough!" << std::endl;
        return false;
    }
    for (size_t i = 0; i < nnz_; ++i) {
        if (ri[i] >= row_ || ci[i] >= col_) {
            std::cerr << "Error, file " << file << " row or col is too big!" << std::endl;
            return false;
        }
    }
    std::vector<UIN> ro;
    stableSortByRow(row_, ri, ci, va, ro);
    if (hasDuplicateInRows(ro, ci)) {
        std::cerr << "Error, matrix has duplicate data!" << std::endl;
        return false;
    }
    if (nnz_ <= 1) {
        std::cerr << "Warning, file " << file << " nnz is 1, this is not a valid matrix!" << std::endl;
        return false;
    }
    rowOffsets_.swap(ro);
    colIndices_.swap(ci);
    values_.swap(va);
    return true;
}

template <typename T>
bool sparseMatrix::CSR<T>::initializeFromSmtxFile(const std::string& file) {
    std::string buf;
    if (!readWholeFile(file, buf)) return false;
    std::cout << "sparseMatrix::CSR initialize From file : " << file << std::endl;

    LineReader lr(buf);
    const char *lb = nullptr, *le = nullptr;
    bool haveHeader = false;
    while (lr.next(lb, le)) {
        if (lb < le && *lb == '%') continue;
        haveHeader = true;
        break;
    }
    if (!haveHeader) return false;
    // "rows, cols, nnz": words are split on blanks and parsed with stoi
    // semantics, so the trailing commas are tolerated.
    {
        const char* p = lb;
        const char *wb, *we;
        long long v[3] = {0, 0, 0};
        for (int i = 0; i < 3; ++i) {
            nextWord(p, le, wb, we);
            if (!wordToInt(wb, we, v[i]) || v[i] < 0) {
                std::cerr << "Error, file " << file << " format is incorrect!" << std::endl;
                return false;
            }
        }
        row_ = static_cast<UIN>(v[0]);
        col_ = static_cast<UIN>(v[1]);
        nnz_ = static_cast<UIN>(v[2]);
    }
    if (nnz_ == 0) {
        std::cerr << "Error, file " << file << " nnz is 0!" << std::endl;
        return false;
    }
    std::vector<UIN> ro(static_cast<size_t>(row_) + 1), ci(nnz_);
    auto readList = [&](std::vector<UIN>& dst, const char* what) {
        if (!lr.next(lb, le)) lb = le = nullptr;
        const char* p = lb;
        const char *wb, *we;
        for (size_t i = 0; i < dst.size(); ++i) {
            long long v = 0;
            if (p >= le) {
                std::cerr << "Error, file " << file << " " << what << " is not enough!" << std::endl;
                return false;
            }
            nextWord(p, le, wb, we);
            if (!wordToInt(wb, we, v) || v < 0) {
                std::cerr << "Error, file " << file << " " << what << " is malformed!" << std::endl;
                return false;
            }
            dst[i] = static_cast<UIN>(v);
        }
        return true;
    };
    if (!readList(ro, "rowOffsets")) return false;
    if (!readList(ci, "nnz")) return false;
    if (ro.front() != 0 || ro.back() != nnz_ || !std::is_sorted(ro.begin(), ro.end())) {
        std::cerr << "Error, file " << file << " rowOffsets are inconsistent!" << std::endl;
        return false;
    }
    for (const UIN c : ci) {
        if (c >= col_) {
            std::cerr << "Error, file " << file << " row or col is too big!" << std::endl;
            return false;
        }
    }
    if (hasDuplicateInRows(ro, ci)) {
        std::cerr << "Error, matrix has duplicate data!" << std::endl;
        return false;
    }
    rowOffsets_.swap(ro);
    colIndices_.swap(ci);
    values_.assign(nnz_, static_cast<T>(1));
    return true;
}

template <typename T>
bool sparseMatrix::CSR<T>::initializeFromGraphDataset(const std::string& file) {
    // SNAP edge list: '#' comment lines carrying "Nodes: n" and "Edges: m", then
    // "src dst [w]" lines; node ids are renumbered in order of first appearance.
    std::string buf;
    if (!readWholeFile(file, buf)) return false;
    std::cout << "sparseMatrix::CSR initialize From file : " << file << std::endl;

    row_ = col_ = nnz_ = 0;
    LineReader lr(buf);
    const char *lb = nullptr, *le = nullptr;
    bool haveLine = false;
    while (lr.next(lb, le)) {
        if (!(lb < le && *lb == '#')) { haveLine = true; break; }
        const std::string line(lb, le);
        auto grab = [&](const char* key, UIN& dst) {
            const size_t at = line.find(key);
            if (at == std::string::npos) return;
            int it = static_cast<int>(at + strlen(key));
            long long v = 0;
            const std::string w = util::iterateOneWordFromLine(line, it);
            if (wordToInt(w.data(), w.data() + w.size(), v) && v >= 0) dst = static_cast<UIN>(v);
        };
        UIN nodes = 0;
        grab("Nodes: ", nodes);
        if (nodes) row_ = col_ = nodes;
        grab("Edges: ", nnz_);
    }
    if (!row_ || !col_ || !nnz_) {
        std::cerr << "Error, file " << file << " row or col or nnz not initialized!" << std::endl;
        return false;
    }
    std::vector<UIN> ri(nnz_), ci(nnz_);
    std::vector<T> va(nnz_, T(0));
    std::unordered_map<UIN, UIN> nodeId;
    auto idOf = [&](UIN node) {
        const auto it = nodeId.find(node);
        if (it != nodeId.end()) return it->second;
        const UIN id = static_cast<UIN>(nodeId.size());
        nodeId.emplace(node, id);
        return id;
    };
    size_t idx = 0;
    while (haveLine) {
        if (lb != le) {
            const char* p = lb;
            const char *wb, *we;
            long long a = 0, b = 0;
            double v = 0;
            nextWord(p, le, wb, we);
            const bool okA = wordToInt(wb, we, a);
            nextWord(p, le, wb, we);
            const bool okB = wordToInt(wb, we, b);
            nextWord(p, le, wb, we);
            const bool okV = wordToDouble(wb, we, v);
            if (!okA || !okB || !okV) {
                std::cerr << "Error, file " << file << " has a malformed entry line!" << std::endl;
                return false;
            }
            const UIN ia = idOf(static_cast<UIN>(a));
            const UIN ib = idOf(static_cast<UIN>(b));
            if (idx >= nnz_) {
                std::cerr << "Error, file " << file
                          << " too many elements, exceeding the number nnz!" << std::endl;
                return false;
            }
            ri[idx] = ia;
            ci[idx] = ib;
            va[idx] = static_cast<T>(v);
            ++idx;
        }
        haveLine = lr.next(lb, le);
    }
    if (idx < nnz_) {
        std::cerr << "Error, file " << file << " elements is not enough!" << std::endl;
        return false;
    }
    for (size_t i = 0; i < nnz_; ++i) {
        if (ri[i] >= row_ || ci[i] >= col_) {
            std::cerr << "Error, file " << file << " row or col is too big!" << std::endl;
            return false;
        }
    }
    std::vector<UIN> ro;
    stableSortByRow(row_, ri, ci, va, ro);
    UIN br = 0, bc = 0;
    if (hasDuplicateInRows(ro, ci, &br, &bc)) {
        fprintf(stderr, "Error, matrix has duplicate data! row:%u, col:%u\n", br, bc);
        return false;
    }
    rowOffsets_.swap(ro);
    colIndices_.swap(ci);
    values_.swap(va);
    return true;
}

template <typename T>
bool sparseMatrix::CSR<T>::initializeFromNpzFile(const std::string& file) {
    // Graph archive of the reference's scripts/convert_mtx_to_npz.py:9-41 (np.savez): src_li / dst_li are the
    // 0-based row and column of every edge, num_nodes_src x num_nodes_dst the shape, num_edges the count.
    // Values are not stored: the pattern gets zeros (SDDMM never reads S's values, src/host.cpp:62-73).
    std::map<std::string, npz::Array> arrays;
    std::string why;
    if (!npz::readIntegerArrays(file, arrays, why)) {
        std::cerr << "Error, file " << file << " : " << why << std::endl;
        return false;
    }
    std::cout << "sparseMatrix::CSR initialize From file : " << file << std::endl;
    auto scalar = [&](const char* name, long long& v) {
        const auto it = arrays.find(name);
        if (it == arrays.end() || it->second.values.size() != 1) return false;
        v = it->second.values[0];
        return true;
    };
    long long rows = 0, cols = 0, edges = 0;
    const auto src = arrays.find("src_li"), dst = arrays.find("dst_li");
    if (src == arrays.end() || dst == arrays.end() || !scalar("num_nodes_src", rows) || !scalar("num_nodes_dst", cols)) {
        std::cerr << "Error, file " << file << " lacks src_li / dst_li / num_nodes_src / num_nodes_dst!" << std::endl;
        return false;
    }
    if (!scalar("num_edges", edges)) edges = static_cast<long long>(src->second.values.size());
    if (rows <= 0 || cols <= 0 || edges <= 1 || rows > 0xFFFFFFFELL || cols > 0xFFFFFFFELL || edges > 0xFFFFFFFELL ||
        src->second.values.size() != static_cast<size_t>(edges) || dst->second.values.size() != static_cast<size_t>(edges)) {
        std::cerr << "Error, file " << file << " has inconsistent sizes!" << std::endl;
        return false;
    }
    row_ = static_cast<UIN>(rows);
    col_ = static_cast<UIN>(cols);
    nnz_ = static_cast<UIN>(edges);
    std::vector<UIN> ri(nnz_), ci(nnz_);
    std::vector<T> va(nnz_, T(0));
    for (size_t i = 0; i < nnz_; ++i) {
        const long long r = src->second.values[i], c = dst->second.values[i];
        if (r < 0 || c < 0 || r >= rows || c >= cols) {
            std::cerr << "Error, file " << file << " row or col is too big!" << std::endl;
            return false;
        }
        ri[i] = static_cast<UIN>(r);
        ci[i] = static_cast<UIN>(c);
    }
    std::vector<UIN> ro;
    stableSortByRow(row_, ri, ci, va, ro);
    UIN br = 0, bc = 0;
    if (hasDuplicateInRows(ro, ci, &br, &bc)) {
        fprintf(stderr, "Error, matrix has duplicate data! row:%u, col:%u\n", br, bc);
        return false;
    }
    rowOffsets_.swap(ro);
    colIndices_.swap(ci);
    values_.swap(va);
    return true;
}

template <typename T>
bool sparseMatrix::CSR<T>::outputToMarketMatrixFile() const {
    return outputToMarketMatrixFile("matrix_" + std::to_string(row_) + "_" + std::to_string(col_) +
                                    "_" + std::to_string(nnz_));
}

template <typename T>
bool sparseMatrix::CSR<T>::outputToMarketMatrixFile(const std::string& fileName) const {
    return sparseMatrix::COO<T>(*this).outputToMarketMatrixFile(fileName);
}

// ---------------------------------------------------------------------------
// COO<T>
// ---------------------------------------------------------------------------
template <typename T>
sparseMatrix::COO<T>::COO(const CSR<T>& csr) {
    row_ = csr.row();
    col_ = csr.col();
    nnz_ = csr.nnz();
    rowIndices_.resize(nnz_);
    colIndices_ = csr.colIndices();
    values_ = csr.values();
    for (size_t r = 0; r < row_; ++r)
        for (UIN e = csr.rowOffsets()[r]; e < csr.rowOffsets()[r + 1]; ++e)
            rowIndices_[e] = static_cast<UIN>(r);
}

template <typename T>
bool sparseMatrix::COO<T>::initializeFromMatrixMarketFile(const std::string& file) {
    CSR<T> csr;
    if (!csr.initializeFromMtxFile(file)) return false;
    *this = COO<T>(csr);
    return true;
}

template <typename T>
sparseMatrix::CSR<T> sparseMatrix::COO<T>::getCsrData() const {
    std::vector<UIN> ci = colIndices_;
    std::vector<T> va = values_;
    std::vector<UIN> ro;
    stableSortByRow(row_, rowIndices_, ci, va, ro);
    return CSR<T>(row_, col_, nnz_, ro, ci, va);
}

template <typename T>
bool sparseMatrix::COO<T>::outputToMarketMatrixFile() const {
    return outputToMarketMatrixFile("matrix_" + std::to_string(row_) + "_" + std::to_string(col_) +
                                    "_" + std::to_string(nnz_));
}

template <typename T>
bool sparseMatrix::COO<T>::outputToMarketMatrixFile(const std::string& fileName) const {
    const std::string name = util::getFileSuffix(fileName) == ".mtx" ? fileName : fileName + ".mtx";
    std::ofstream out(name);
    if (!out.is_open()) {
        std::cerr << "Error, file cannot be created : " << name << std::endl;
        return false;
    }
    out << "%%MatrixMarket matrix coordinate real general\n";
    out << row_ << " " << col_ << " " << nnz_ << "\n";
    for (size_t i = 0; i < nnz_; ++i)
        out << rowIndices_[i] + 1 << " " << colIndices_[i] + 1 << " " << values_[i] << "\n";
    return true;
}

template <typename T>
bool checkMatrixData(const sparseMatrix::CSR<T>& csr) {
    const auto& ro = csr.rowOffsets();
    const auto& ci = csr.colIndices();
    if (ro.size() != static_cast<size_t>(csr.row()) + 1 || ro.front() != 0 || ro.back() != csr.nnz())
        return false;
    if (!std::is_sorted(ro.begin(), ro.end())) return false;
    if (ci.size() != csr.nnz() || csr.values().size() != csr.nnz()) return false;
    for (const UIN c : ci)
        if (c >= csr.col()) return false;
    return !hasDuplicateInRows(ro, ci);
}

// ---------------------------------------------------------------------------
// explicit instantiations (same set as the reference, src/Matrix.cpp:252-277)
// ---------------------------------------------------------------------------
template class Matrix<int>;
template class Matrix<float>;
template class Matrix<double>;
template class sparseMatrix::CSR<int>;
template class sparseMatrix::CSR<float>;
template class sparseMatrix::CSR<double>;
template class sparseMatrix::COO<int>;
template class sparseMatrix::COO<float>;
template class sparseMatrix::COO<double>;
template bool checkMatrixData<int>(const sparseMatrix::CSR<int>&);
template bool checkMatrixData<float>(const sparseMatrix::CSR<float>&);
template bool checkMatrixData<double>(const sparseMatrix::CSR<double>&);
