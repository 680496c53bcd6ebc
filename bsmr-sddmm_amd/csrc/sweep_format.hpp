// Device format of the dense part for the panel-SWEEP engine (round 3), and its host packer.  Pure host C++.
//
// Every other dense format of this library follows the reference's shape: per 16-row panel a list of (gathered)
// 16-column blocks (reference src/BSMR.cpp:143-174, src/sddmmKernel.cu:213-351).  On matrices whose natural
// 16 x 16 tiles are almost all occupied (nips-like: 97 %, 4096^2 Bernoulli(0.1): 100 %) the per-panel column
// compaction saves little and costs a gather of B per panel.  The sweep engine computes the masked product the
// way a GEMM would: a work item is R = 16 * W * PW reordered rows (W waves x PW panels) times a STRIP of consecutive
// 16-column blocks of B in natural column order; B is streamed once per item, contiguously (a block of 16
// columns of the column-major B is one contiguous run of 16 * K elements), and each wave multiplies the block
// against its PW panels.  The sparse mask is applied by construction, as in the tiles engine: the accumulators
// of a (wave, block) step go to an LDS slab and one lane per stored entry carries its value to P.
//
//   panelRows [G * 16 W PW]  original row ids, 16 per panel, panels in RPHM order (padding: row 0 of the list)
//   items     [S * G]        {group, first block, blocks, index of its first `starts` word}, strip-major: the
//                            items of one strip (same B columns) are neighbours and run on one XCD
//   starts    per item [W][blocks + 1]   wave w's entry words of block b are words[starts[w][b] .. starts[w][b+1]);
//                            every list is padded to a multiple of 4 words (the loaders move them in 16-byte
//                            pieces), the padding words are kSweepNoEntry
//   rowStart  [items][W][16 PW]          smallest CSR index among the entries the item holds of that row
//   words     [u32]          one per stored dense entry:  slab slot (10 bits) | row in wave (6) << 10 | offset << 16
//                            slab slot = (j * 64 + 16 (r >> 2) + c) * 4 + (r & 3)  for row r, column c of the wave's
//                            panel j (the MFMA accumulator layout: lane = 16 (r >> 2) + c, register r & 3; a lane
//                            writes its four registers of a panel as one 16-byte piece)
//                            offset    = CSR index - rowStart of its row (< 65536, checked)
// An entry is listed iff the RPHM (after the plan's own promotion / folding) has it in a dense block, so the
// dense / sparse assignment of every nnz is unchanged; rows need not be sorted.
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

#include "bsmr_hip.h"
#include "tile_format.hpp"

namespace bsmr {

constexpr uint32_t kSweepMaxWaves = 8;         // consumer waves per workgroup: 4 or 8
constexpr uint32_t kSweepMaxBlocks = 63;       // blocks per strip (lane b of a wave holds starts[b])
constexpr uint32_t kSweepMaxPanelsPerWave = 4; // PW
constexpr uint32_t kSweepWordSlack = 8192;      // words behind the last list that the kernel may read (never use)
constexpr uint32_t kSweepNoEntry = 0xFFFFFFFFu; // padding word of an entry list (row 63 of a wave does not exist: PW * 16 <= 64 ... and offset 65535)

struct SweepItem {
    uint32_t group;
    uint32_t firstBlock;
    uint32_t numBlocks;
    uint32_t startsBase;
};

struct SweepFormatHost {
    uint32_t W = 0, PW = 0, stripBlocks = 0, numGroups = 0, numStrips = 0, numColBlocks = 0;
    std::vector<uint32_t> panelRows;
    std::vector<SweepItem> items;
    std::vector<uint32_t> starts;
    std::vector<uint32_t> rowStart;
    std::vector<uint32_t> words;
    uint32_t maxStepEntries = 0;    // most entries one wave writes after one block
    uint32_t maxItemWords = 0;      // most entry words (padding included) of one item: the kernel keeps an item's words in LDS
    size_t bytes() const {
        return 4 * (panelRows.size() + starts.size() + rowStart.size() + words.size()) + sizeof(SweepItem) * items.size();
    }
};

// Packs the dense entries for W consumer waves, PW panels per wave and strips of `stripBlocks` 16-column blocks.
// BSMR_ERR_INVALID_ARG: parameters out of range; BSMR_ERR_BAD_PLAN: a row's entries inside one strip span
// 65535 or more CSR positions (the 16-bit offset does not reach; the caller keeps the other engines).
inline int packSweep(const HostDense& hd, uint32_t W, uint32_t PW, uint32_t stripBlocks, SweepFormatHost& out) {
    if (PW == 0 || PW > kSweepMaxPanelsPerWave || (PW & (PW - 1)) || stripBlocks == 0 || stripBlocks > kSweepMaxBlocks ||
        (W != 4 && W != 8))
        return BSMR_ERR_INVALID_ARG;
    const uint32_t kSweepWaves = W;
    const uint32_t P = hd.numPanels, GP = kSweepWaves * PW;   // panels per group
    const uint32_t G = (P + GP - 1) / GP;
    const uint32_t NCB = (hd.N + 15) / 16;
    const uint32_t S = (NCB + stripBlocks - 1) / stripBlocks;
    out = SweepFormatHost();
    out.W = W; out.PW = PW; out.stripBlocks = stripBlocks; out.numGroups = G; out.numStrips = S; out.numColBlocks = NCB;
    if (P == 0 || NCB == 0) return BSMR_OK;
    if ((uint64_t)G * S > 0x7FFFFFFFull) return BSMR_ERR_INVALID_ARG;
    const uint32_t rowsPerWave = 16 * PW;
    out.panelRows.assign((size_t)G * GP * 16, hd.panelRows.empty() ? 0u : hd.panelRows[0]);
    std::copy(hd.panelRows.begin(), hd.panelRows.end(), out.panelRows.begin());
    const size_t numItems = (size_t)G * S;
    out.items.resize(numItems);
    size_t startsTotal = 0;
    for (uint32_t s = 0; s < S; ++s) {
        const uint32_t b0 = s * stripBlocks, nb = std::min(stripBlocks, NCB - b0);
        for (uint32_t g = 0; g < G; ++g) {
            SweepItem& it = out.items[(size_t)s * G + g];
            it.group = g; it.firstBlock = b0; it.numBlocks = nb; it.startsBase = (uint32_t)startsTotal;
            startsTotal += (size_t)kSweepWaves * (nb + 1);
        }
    }
    if (startsTotal > 0xFFFFFFFFull || hd.entries() > 0xFFFFFFFFull) return BSMR_ERR_INVALID_ARG;
    out.starts.assign(startsTotal, 0);
    out.rowStart.assign(numItems * kSweepWaves * rowsPerWave, 0xFFFFFFFFu);
    // 1. entries per (item, wave, block) and the smallest CSR index per (item, row)
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t g = p / GP, w = (p % GP) / PW, j = p % PW;
        for (uint64_t e = hd.offsets[p]; e < hd.offsets[p + 1]; ++e) {
            const uint32_t cb = hd.col[e] >> 4, s = cb / stripBlocks, b = cb - s * stripBlocks;
            const size_t item = (size_t)s * G + g;
            ++out.starts[(size_t)out.items[item].startsBase + (size_t)w * (out.items[item].numBlocks + 1) + b + 1];
            uint32_t& rs = out.rowStart[(item * kSweepWaves + w) * rowsPerWave + j * 16 + hd.row[e]];
            rs = std::min(rs, hd.idx[e]);
        }
    }
    for (uint32_t& v : out.rowStart)
        if (v == 0xFFFFFFFFu) v = 0;
    // 2. prefix sums: every (item, wave) list is one run of `words`
    uint64_t at = 0;
    for (size_t item = 0; item < numItems; ++item) {
        const uint32_t nb = out.items[item].numBlocks;
        for (uint32_t w = 0; w < kSweepWaves; ++w) {
            uint32_t* st = &out.starts[(size_t)out.items[item].startsBase + (size_t)w * (nb + 1)];
            uint64_t run = at;
            for (uint32_t b = 0; b <= nb; ++b) {
                const uint32_t n = st[b];     // st[0] is 0, st[b + 1] held block b's count
                out.maxStepEntries = std::max(out.maxStepEntries, n);
                run += (n + 3u) & ~3u;
                st[b] = (uint32_t)run;
            }
            at = run;
        }
        out.maxItemWords = std::max<uint32_t>(out.maxItemWords, (uint32_t)(at - out.starts[out.items[item].startsBase]));
    }
    out.words.assign((size_t)at + kSweepWordSlack, kSweepNoEntry);  // (+ slack: the loaders move an item's words in whole pseudo-images)
    // 3. the words, in panel order inside a (item, wave, block) list
    std::vector<uint32_t> fill(out.starts.size());
    for (size_t item = 0; item < numItems; ++item) {
        const uint32_t nb = out.items[item].numBlocks;
        for (uint32_t w = 0; w < kSweepWaves; ++w) {
            const size_t base = (size_t)out.items[item].startsBase + (size_t)w * (nb + 1);
            for (uint32_t b = 0; b < nb; ++b) fill[base + b] = out.starts[base + b];
        }
    }
    for (uint32_t p = 0; p < P; ++p) {
        const uint32_t g = p / GP, w = (p % GP) / PW, j = p % PW;
        for (uint64_t e = hd.offsets[p]; e < hd.offsets[p + 1]; ++e) {
            const uint32_t col = hd.col[e], cb = col >> 4, s = cb / stripBlocks, b = cb - s * stripBlocks;
            const size_t item = (size_t)s * G + g;
            const uint32_t r = hd.row[e], c = col & 15u;
            const uint32_t rowInWave = j * 16 + r;
            const uint32_t off = hd.idx[e] - out.rowStart[(item * kSweepWaves + w) * rowsPerWave + rowInWave];
            if (off >= 0xFFFFu) return BSMR_ERR_BAD_PLAN;   // (0xFFFF is left to the padding word)
            const uint32_t slot = (j * 64 + (r >> 2) * 16 + c) * 4 + (r & 3u);
            uint32_t& pos = fill[(size_t)out.items[item].startsBase + (size_t)w * (out.items[item].numBlocks + 1) + b];
            out.words[pos++] = slot | (rowInWave << 10) | (off << 16);
        }
    }
    return BSMR_OK;
}

}  // namespace bsmr
