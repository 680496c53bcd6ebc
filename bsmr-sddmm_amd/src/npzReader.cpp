// .npz reader (include/npzReader.hpp).  Format facts used here:
//   ZIP: the end-of-central-directory record (signature 0x06054b50) is in the last 64 KiB + 22 bytes; a ZIP64
//        locator (0x07064b50) in front of it points to the ZIP64 end record (0x06064b50).  Every central
//        directory entry (0x02014b50) carries method, sizes and the offset of the member's local header
//        (0x04034b50), whose own name / extra lengths say where the data starts.  Sizes of 0xFFFFFFFF are
//        replaced by the ZIP64 extra field (id 0x0001).  np.savez writes members through a stream, so the local
//        header's sizes are not reliable: only the central directory is trusted.
//   NPY: "\x93NUMPY", major, minor, header length (2 bytes for version 1, 4 for 2 and 3), then a Python dict
//        literal with 'descr', 'fortran_order' and 'shape', then the raw data.
#include "npzReader.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace npz {
namespace {

uint64_t le(const uint8_t* p, int bytes) {
    uint64_t v = 0;
    for (int i = bytes - 1; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}

bool readAt(FILE* f, uint64_t offset, void* dst, size_t n) {
    if (fseeko(f, static_cast<off_t>(offset), SEEK_SET) != 0) return false;
    return fread(dst, 1, n, f) == n;
}

struct Member {
    std::string name;
    uint16_t method = 0;
    uint64_t compressed = 0, size = 0, localHeader = 0;
};

bool listMembers(FILE* f, uint64_t fileSize, std::vector<Member>& members, std::string& error) {
    const uint64_t tail = fileSize < 65557 ? fileSize : 65557;
    std::vector<uint8_t> buf(tail);
    if (tail < 22 || !readAt(f, fileSize - tail, buf.data(), tail)) {
        error = "too short for a zip archive";
        return false;
    }
    int64_t at = -1;
    for (int64_t i = static_cast<int64_t>(tail) - 22; i >= 0; --i)
        if (le(&buf[i], 4) == 0x06054b50u) {
            at = i;
            break;
        }
    if (at < 0) {
        error = "no end-of-central-directory record";
        return false;
    }
    uint64_t entries = le(&buf[at + 10], 2), dirSize = le(&buf[at + 12], 4), dirOffset = le(&buf[at + 16], 4);
    if (at >= 20 && le(&buf[at - 20], 4) == 0x07064b50u) {  // ZIP64 locator
        const uint64_t endOffset = le(&buf[at - 20 + 8], 8);
        uint8_t rec[56];
        if (!readAt(f, endOffset, rec, sizeof rec) || le(rec, 4) != 0x06064b50u) {
            error = "bad ZIP64 end record";
            return false;
        }
        entries = le(rec + 32, 8);
        dirSize = le(rec + 40, 8);
        dirOffset = le(rec + 48, 8);
    }
    if (dirOffset > fileSize || dirSize > fileSize - dirOffset) {
        error = "central directory outside the file";
        return false;
    }
    std::vector<uint8_t> dir(dirSize);
    if (dirSize && !readAt(f, dirOffset, dir.data(), dirSize)) {
        error = "cannot read the central directory";
        return false;
    }
    size_t p = 0;
    for (uint64_t e = 0; e < entries; ++e) {
        if (p + 46 > dir.size() || le(&dir[p], 4) != 0x02014b50u) {
            error = "bad central directory entry";
            return false;
        }
        Member m;
        m.method = static_cast<uint16_t>(le(&dir[p + 10], 2));
        m.compressed = le(&dir[p + 20], 4);
        m.size = le(&dir[p + 24], 4);
        const size_t nameLen = le(&dir[p + 28], 2), extraLen = le(&dir[p + 30], 2), commentLen = le(&dir[p + 32], 2);
        m.localHeader = le(&dir[p + 42], 4);
        if (p + 46 + nameLen + extraLen + commentLen > dir.size()) {
            error = "truncated central directory entry";
            return false;
        }
        m.name.assign(reinterpret_cast<const char*>(&dir[p + 46]), nameLen);
        // ZIP64 extra: the fields present are exactly those whose 32-bit value is all ones, in this order
        size_t x = p + 46 + nameLen;
        const size_t xEnd = x + extraLen;
        while (x + 4 <= xEnd) {
            const uint64_t id = le(&dir[x], 2), len = le(&dir[x + 2], 2);
            size_t q = x + 4;
            if (id == 0x0001) {
                if (m.size == 0xFFFFFFFFu && q + 8 <= xEnd) { m.size = le(&dir[q], 8); q += 8; }
                if (m.compressed == 0xFFFFFFFFu && q + 8 <= xEnd) { m.compressed = le(&dir[q], 8); q += 8; }
                if (m.localHeader == 0xFFFFFFFFu && q + 8 <= xEnd) { m.localHeader = le(&dir[q], 8); q += 8; }
            }
            x += 4 + len;
        }
        members.push_back(m);
        p += 46 + nameLen + extraLen + commentLen;
    }
    return true;
}

bool readMember(FILE* f, uint64_t fileSize, const Member& m, std::vector<uint8_t>& data, std::string& error) {
    uint8_t head[30];
    if (!readAt(f, m.localHeader, head, sizeof head) || le(head, 4) != 0x04034b50u) {
        error = "bad local header of " + m.name;
        return false;
    }
    const uint64_t start = m.localHeader + 30 + le(head + 26, 2) + le(head + 28, 2);
    if (start > fileSize || m.compressed > fileSize - start) {
        error = "member " + m.name + " reaches beyond the file";
        return false;
    }
    if (m.method == 0) {
        data.resize(m.size);
        if (m.size != m.compressed || (m.size && !readAt(f, start, data.data(), m.size))) {
            error = "cannot read " + m.name;
            return false;
        }
        return true;
    }
    if (m.method != 8) {
        error = "member " + m.name + " uses an unsupported compression method";
        return false;
    }
    std::vector<uint8_t> packed(m.compressed);
    if (m.compressed && !readAt(f, start, packed.data(), m.compressed)) {
        error = "cannot read " + m.name;
        return false;
    }
    data.resize(m.size);
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, -MAX_WBITS) != Z_OK) {
        error = "zlib initialisation failed";
        return false;
    }
    // (zlib counts in 32-bit uInt: feed and drain in pieces so that members beyond 4 GiB work)
    size_t in = 0, outPos = 0;
    int rc = Z_OK;
    while (rc == Z_OK) {
        if (z.avail_in == 0 && in < packed.size()) {
            const size_t n = std::min<size_t>(packed.size() - in, 1u << 30);
            z.next_in = packed.data() + in;
            z.avail_in = static_cast<uInt>(n);
            in += n;
        }
        if (z.avail_out == 0 && outPos < data.size()) {
            const size_t n = std::min<size_t>(data.size() - outPos, 1u << 30);
            z.next_out = data.data() + outPos;
            z.avail_out = static_cast<uInt>(n);
            outPos += n;
        }
        rc = inflate(&z, Z_NO_FLUSH);  // Z_BUF_ERROR once neither more input nor more room can be offered
    }
    const bool complete = rc == Z_STREAM_END && z.total_out == data.size();
    inflateEnd(&z);
    if (!complete) {
        error = "member " + m.name + " does not inflate to its recorded size";
        return false;
    }
    return true;
}

// value of 'key' in the header dict: the text up to the matching ',' at depth 0 or the closing '}'
std::string dictValue(const std::string& header, const std::string& key) {
    const size_t k = header.find("'" + key + "'");
    if (k == std::string::npos) return "";
    size_t p = header.find(':', k);
    if (p == std::string::npos) return "";
    ++p;
    int depth = 0;
    size_t e = p;
    for (; e < header.size(); ++e) {
        const char c = header[e];
        if (c == '(' || c == '[') ++depth;
        else if (c == ')' || c == ']') --depth;
        else if ((c == ',' || c == '}') && depth == 0) break;
    }
    std::string v = header.substr(p, e - p);
    while (!v.empty() && (v.front() == ' ')) v.erase(v.begin());
    while (!v.empty() && (v.back() == ' ')) v.pop_back();
    return v;
}

// true if `data` is an .npy with a little-endian (or single-byte) integer dtype; fills `out`
bool parseNpy(const std::vector<uint8_t>& data, Array& out, bool& isInteger, std::string& error) {
    isInteger = false;
    if (data.size() < 10 || memcmp(data.data(), "\x93NUMPY", 6) != 0) {
        error = "not an .npy member";
        return false;
    }
    const int major = data[6];
    const size_t lenBytes = major == 1 ? 2 : 4;
    if (data.size() < 8 + lenBytes) {
        error = "truncated .npy header";
        return false;
    }
    const size_t headerLen = static_cast<size_t>(le(&data[8], static_cast<int>(lenBytes)));
    const size_t dataStart = 8 + lenBytes + headerLen;
    if (dataStart > data.size()) {
        error = "truncated .npy header";
        return false;
    }
    const std::string header(reinterpret_cast<const char*>(&data[8 + lenBytes]), headerLen);
    std::string descr = dictValue(header, "descr");
    if (descr.size() >= 2 && (descr.front() == '\'' || descr.front() == '"')) descr = descr.substr(1, descr.size() - 2);
    if (descr.size() < 3) return true;  // structured or unknown: skipped
    const char order = descr[0], kind = descr[1];
    const int width = atoi(descr.c_str() + 2);
    if ((kind != 'i' && kind != 'u') || (width != 1 && width != 2 && width != 4 && width != 8)) return true;
    if (order == '>' && width > 1) return true;  // big-endian files are not produced by the converter
    if (dictValue(header, "fortran_order").rfind("True", 0) == 0) return true;
    const std::string shape = dictValue(header, "shape");
    out.shape.clear();
    uint64_t count = 1;
    for (size_t i = 0; i < shape.size();) {
        if (shape[i] >= '0' && shape[i] <= '9') {
            uint64_t v = 0;
            while (i < shape.size() && shape[i] >= '0' && shape[i] <= '9') v = v * 10 + static_cast<uint64_t>(shape[i++] - '0');
            out.shape.push_back(v);
            if (v != 0 && count > UINT64_MAX / v) {
                error = "array too large";
                return false;
            }
            count *= v;
        } else {
            ++i;
        }
    }
    if (count > (data.size() - dataStart) / static_cast<size_t>(width)) {
        error = "array data shorter than its shape";
        return false;
    }
    out.values.resize(count);
    const uint8_t* p = data.data() + dataStart;
    for (uint64_t i = 0; i < count; ++i, p += width) {
        const uint64_t raw = le(p, width);
        if (kind == 'u' || width == 8) out.values[i] = static_cast<int64_t>(raw);
        else {
            const uint64_t sign = 1ull << (8 * width - 1);
            out.values[i] = static_cast<int64_t>((raw ^ sign)) - static_cast<int64_t>(sign);
        }
    }
    isInteger = true;
    return true;
}

}  // namespace

bool readIntegerArrays(const std::string& file, std::map<std::string, Array>& out, std::string& error) {
    out.clear();
    FILE* f = fopen(file.c_str(), "rb");
    if (!f) {
        error = "cannot open the file";
        return false;
    }
    bool ok = fseeko(f, 0, SEEK_END) == 0;
    const uint64_t fileSize = ok ? static_cast<uint64_t>(ftello(f)) : 0;
    std::vector<Member> members;
    ok = ok && listMembers(f, fileSize, members, error);
    for (size_t i = 0; ok && i < members.size(); ++i) {
        const Member& m = members[i];
        if (m.name.size() < 4 || m.name.compare(m.name.size() - 4, 4, ".npy") != 0) continue;
        std::vector<uint8_t> data;
        Array a;
        bool isInteger = false;
        ok = readMember(f, fileSize, m, data, error) && parseNpy(data, a, isInteger, error);
        if (ok && isInteger) out[m.name.substr(0, m.name.size() - 4)] = std::move(a);
    }
    fclose(f);
    return ok;
}

}  // namespace npz
