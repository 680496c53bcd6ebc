// Reader for NumPy .npz archives (a ZIP of .npy members), enough for the graph files the reference's
// scripts/convert_mtx_to_npz.py writes with np.savez: members src_li / dst_li (int32 edge lists) and the 0-d
// integers num_nodes_src / num_nodes_dst / num_edges.  Stored and deflated members, ZIP64 sizes, little-endian
// integer dtypes of 1-8 bytes.  No NumPy, no Python: the CLI and the C ABI load such files directly.
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace npz {

struct Array {
    std::vector<uint64_t> shape;   // empty for a 0-d array
    std::vector<int64_t> values;   // every element widened to int64 (C order)
};

// Reads every integer member of `file` into `out` (key = member name without ".npy").  Members of other
// dtypes are skipped.  Returns false with a message in `error` when the file is not a readable .npz.
bool readIntegerArrays(const std::string& file, std::map<std::string, Array>& out, std::string& error);

}  // namespace npz
