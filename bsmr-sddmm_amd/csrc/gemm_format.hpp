// Device format of the dense part for the output-stationary GEMM engine (round 4), and its host packer.  Pure host C++.
//
// What is computed is the dense part of the reference's SDDMM (reference src/sddmmKernel.cu:213-351: per 16 x 16 block
// a K loop of tensor-core steps, then the masked scatter of the accumulators, :331-348).  The sweep engine (round 3,
// csrc/sweep_format.hpp) already ignored the blocks' column choice and multiplied natural 16-column blocks of B; it kept
// the A fragments of a wave for the whole K in registers, so one B fragment fed at most four MFMAs.  This format cuts the
// product the way a GEMM does: a work item is a MACRO-TILE of C, TM = 16 * PM reordered rows (PM panels, RPHM order) times
// TN = 16 * NB natural columns; the kernel keeps the whole macro-tile's accumulators in registers over the K loop, each of
// its WM x WN waves owning an m x n block of 16 x 16 tiles (m = PM / WM, n = NB / WN), so one fragment read of A feeds n
// MFMAs and one of B feeds m.  The sparse mask is applied once per macro-tile, by construction as in the tiles / sweep
// engines: a wave drops its accumulators into a private LDS slab (16 tiles at a time: a PASS) and one lane per STORED entry
// carries its value to P.
//
//   panelRows [G * TM]        original row ids of every macro-tile row.  NOT the RPHM's panel order: rows are dealt over the
//                             2 G row halves (a half = the rows of one wave row) by descending number of dense entries,
//                             back and forth, so that every wave's entry lists are about equally long - the clustered
//                             order puts heavy rows side by side (nips-like: 198 k entries in one row group, 44 k in
//                             another; the fullest wave three times the mean).  Rows of A are gathered by id and a row's
//                             entries are a run of P wherever the row sits, so the order is free (padding: row 0 of the list)
//   colOf     [S * TN]        the column of B in every column slot.  Slots follow the natural column order EXCEPT for hot
//                             columns (more than twice the average number of dense entries), which are dealt over the S
//                             strips by descending count (boustrophedon), so that no strip gathers them.  The MFMA work of a
//                             macro-tile does not depend on what it holds, its mask epilogue is one store per entry: in
//                             natural order the six macro-tiles of the nips-like pattern's first 320 columns (Zipf column
//                             weights) hold 80 000 entries each against an average of 3 200, and the launch waits for
//                             their epilogues (probe, K = 128: 32 us against 13).  B's columns are gathered by id like A's
//                             rows (a column of the column-major B is K contiguous elements), so any order reads the same
//                             bytes; what the order changes is how a row's entries of one macro-tile lie in P, hence
//                             natural order for everything that is not hot.  Unused slots name column 0 (no entry refers
//                             to them).
//   items     [I]             {row group, first 16-column block, index of its first `lists` word}; macro-tiles WITHOUT a
//                             dense entry are not listed.  Order (gemmItemPlace): column slabs of kGemmSlabStrips strips,
//                             inside a slab row group by row group - a launch's eighth (what one XCD works on) is a few row
//                             groups x one slab and shares its rows of A and columns of B in that XCD's L2.  When every
//                             macro-tile is listed (fullGrid) the kernel computes an item's place instead of loading it.
//   rowStart  [I][TM]         smallest CSR index among the entries the item holds of that row (0 if none)
//   lists     [I][W * Q + 1]  wave w, pass q: words[lists[w * Q + q] .. lists[w * Q + q + 1]); Q = ceil(m n / 16) passes;
//                             every list is padded to a multiple of 4 words with kGemmNoEntry
//   words     [u32]           one per stored dense entry, lists ordered by (row, position in P):
//                             slab slot (12 bits) | row in the wave's rows (7) << 12 | offset (13) << 19
//                             slab slot = gemmSlabSlot(tile in pass, r, c) for row r, column c of a 16 x 16 tile (the MFMA
//                             accumulator layout: lane 16 (r >> 2) + c, register r & 3; a register of the 64 lanes is one
//                             run of 256 bytes in the slab); tile t of a wave = (row tile) * n + (column tile), pass
//                             t / 16, tile in pass t % 16
//                             offset = CSR index - rowStart of its row (< 8191: sorted CSR rows give < TN)
// An entry is listed iff the RPHM (after the plan's own promotion / folding) has it in a dense block: the dense / sparse
// assignment of every nnz is unchanged.  Rows need not be sorted, but one row's entries inside one macro-tile must lie
// within 8191 CSR positions of each other (BSMR_ERR_BAD_PLAN otherwise: the caller keeps the other engines).
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

#include "bsmr_hip.h"
#include "tile_format.hpp"

namespace bsmr {

constexpr uint32_t kGemmWavesM = 2, kGemmWavesN = 4, kGemmWaves = kGemmWavesM * kGemmWavesN;
// where the epilogue's dump leaves accumulator cell (r, c) of the pass's tile tp, in floats from the wave's slab: register
// r & 3 of lane 16 (r >> 2) + c; one register of all 64 lanes = 256 contiguous bytes (one ds_write_addtid_b32)
__host__ __device__ constexpr uint32_t gemmSlabSlot(uint32_t tp, uint32_t r, uint32_t c) { return tp * 256u + (r & 3u) * 64u + (r >> 2) * 16u + c; }
constexpr uint32_t kGemmPassTiles = 16;        // 16 x 16 tiles of a wave per slab pass (16 KiB of fp32 per wave)
constexpr uint32_t kGemmNoEntry = 0xFFFFFFFFu; // padding word (offset 8191 is never a real offset)
constexpr uint32_t kGemmMaxOffset = 8191u;
constexpr uint32_t kGemmSlabStrips = 8;        // column strips per slab of the item order
constexpr uint32_t kGemmWordSlack = 1024;      // words behind the last list that the kernel may read (never use)

struct GemmItem {
    uint32_t group;        // row group: panels [group * PM, group * PM + PM)
    uint32_t firstBlock;   // first 16-column block of B
    uint32_t listBase;     // index of lists[item][0]
    uint32_t pad;
};

// place of item i in a full grid of G row groups x S column strips: slabs of kGemmSlabStrips strips (the last one narrower),
// inside a slab row-major
__host__ __device__ inline void gemmItemPlace(uint32_t i, uint32_t G, uint32_t S, uint32_t& group, uint32_t& strip) {
    const uint32_t slabs = (S + kGemmSlabStrips - 1u) / kGemmSlabStrips;
    uint32_t slab = i / (kGemmSlabStrips * G);
    if (slab >= slabs) slab = slabs - 1u;
    const uint32_t width = slab + 1u < slabs ? kGemmSlabStrips : S - kGemmSlabStrips * (slabs - 1u);
    const uint32_t rem = i - slab * kGemmSlabStrips * G;
    group = rem / width;
    strip = slab * kGemmSlabStrips + rem % width;
}

struct GemmFormatHost {
    uint32_t PM = 0, NB = 0, numGroups = 0, numStrips = 0, passes = 0;
    bool fullGrid = false;          // every macro-tile of the G x S grid is an item
    std::vector<uint32_t> panelRows;
    std::vector<uint32_t> colOf;
    std::vector<GemmItem> items;
    std::vector<uint32_t> rowStart;
    std::vector<uint32_t> lists;
    std::vector<uint32_t> words;
    uint64_t numTiles = 0;          // 16 x 16 tiles multiplied: items * PM * NB
    uint32_t maxListWords = 0;      // longest (wave, pass) list
    size_t bytes() const {
        return 4 * (panelRows.size() + colOf.size() + rowStart.size() + lists.size() + words.size()) + sizeof(GemmItem) * items.size();
    }
};

// m = PM / 2 row tiles x n = NB / 4 column tiles per wave: 4 m n accumulator registers (at most 160 of the 256)
inline bool gemmShapeOk(uint32_t PM, uint32_t NB) {
    return (PM == 8 || PM == 16) && (NB == 8 || NB == 12 || NB == 16 || NB == 20);
}

// Packs the dense entries for macro-tiles of PM panels x NB 16-column blocks.  balanceColumns = false keeps the natural column
// order (slot = column id).
inline int packGemm(const HostDense& hd, uint32_t PM, uint32_t NB, GemmFormatHost& out, bool balanceColumns = true) {
    if (!gemmShapeOk(PM, NB)) return BSMR_ERR_INVALID_ARG;
    const uint32_t P = hd.numPanels, TM = PM * 16;
    const uint32_t G = (P + PM - 1) / PM, NCB = (hd.N + 15) / 16, S = (NCB + NB - 1) / NB;
    const uint32_t m = PM / kGemmWavesM, n = NB / kGemmWavesN, Q = (m * n + kGemmPassTiles - 1) / kGemmPassTiles, L = kGemmWaves * Q;
    out = GemmFormatHost();
    out.PM = PM; out.NB = NB; out.numGroups = G; out.numStrips = S; out.passes = Q;
    if (P == 0 || NCB == 0 || hd.entries() == 0) return BSMR_OK;
    if ((uint64_t)G * S > 0x3FFFFFFFull || hd.entries() > 0xFFFFFFF0ull) return BSMR_ERR_INVALID_ARG;
    out.panelRows.assign((size_t)G * TM, hd.panelRows.empty() ? 0u : hd.panelRows[0]);
    // 0a. the place of every row: rowPos[panel * 16 + row in panel] = group * TM + row in macro-tile
    std::vector<uint32_t> rowPos((size_t)P * 16);
    if (balanceColumns) {
        std::vector<uint32_t> weight((size_t)P * 16, 0), order((size_t)P * 16);
        for (uint32_t p = 0; p < P; ++p)
            for (uint64_t e = hd.offsets[p]; e < hd.offsets[p + 1]; ++e) ++weight[(size_t)p * 16 + hd.row[e]];
        for (uint32_t i = 0; i < P * 16; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return weight[a] > weight[b]; });
        const uint32_t bins = 2 * G, half = TM / 2;
        for (uint32_t rank = 0; rank < P * 16; ++rank) {   // round `rank / bins` gives one row to every half
            const uint32_t round = rank / bins, pos = rank % bins, bin = (round & 1u) ? bins - 1u - pos : pos;
            rowPos[order[rank]] = (bin / 2) * TM + (bin % 2) * half + round;
        }
    } else {
        for (uint32_t i = 0; i < P * 16; ++i) rowPos[i] = i;
    }
    for (uint32_t i = 0; i < P * 16; ++i) out.panelRows[rowPos[i]] = hd.panelRows[i];
    // 0. the slot of every column
    const uint32_t TN = NB * 16;
    std::vector<uint32_t> slotOf(hd.N);
    out.colOf.assign((size_t)S * TN, 0u);
    if (balanceColumns) {
        // HOT columns - more than twice the average number of dense entries - are dealt over the strips by descending count,
        // back and forth; the others fill the strips' remaining slots in natural order, so that what a row stores in a
        // macro-tile stays a run of P wherever the pattern has no hot columns (a full permutation was measured: scattered
        // stores cost a uniform pattern 3 - 5 us, 4096^2 K = 512: 17.8 -> 22.5 us)
        std::vector<uint32_t> degree(hd.N, 0), hot;
        for (uint64_t e = 0; e < hd.entries(); ++e) ++degree[hd.col[e]];
        // ... and only where the natural order is lopsided: a strip of natural columns with more than twice the strips' mean
        // (hot columns that lie scattered - the nips-like stand-in's do - balance the strips by themselves, and dealing them
        // would only scatter their stores: 10.9 -> 15.0 us measured on it)
        uint64_t fullest = 0;
        for (uint32_t st = 0; st < S; ++st) {
            uint64_t sum = 0;
            for (uint32_t c = st * TN; c < std::min<uint64_t>(hd.N, (uint64_t)(st + 1) * TN); ++c) sum += degree[c];
            fullest = std::max(fullest, sum);
        }
        const bool lopsided = fullest * S > 2ull * hd.entries();
        const uint64_t threshold = lopsided ? 2ull * hd.entries() / std::max<uint32_t>(1u, hd.N) + 1ull : ~0ull;
        for (uint32_t c = 0; c < hd.N; ++c)
            if (degree[c] > threshold) hot.push_back(c);
        std::stable_sort(hot.begin(), hot.end(), [&](uint32_t a, uint32_t b) { return degree[a] > degree[b]; });
        std::vector<uint32_t> filled(S, 0);
        for (uint32_t rank = 0; rank < hot.size(); ++rank) {   // round `rank / S` deals one column to every strip
            const uint32_t round = rank / S, pos = rank % S, strip = (round & 1u) ? S - 1u - pos : pos;
            slotOf[hot[rank]] = strip * TN + filled[strip]++;
        }
        uint32_t strip = 0;
        for (uint32_t c = 0; c < hd.N; ++c) {
            if (degree[c] > threshold) continue;
            while (filled[strip] == TN) ++strip;   // (S * TN >= N slots in all)
            slotOf[c] = strip * TN + filled[strip]++;
        }
    } else {
        for (uint32_t c = 0; c < hd.N; ++c) slotOf[c] = c;
    }
    for (uint32_t c = 0; c < hd.N; ++c) out.colOf[slotOf[c]] = c;
    // 1. which macro-tiles hold entries, entries per (macro-tile, wave, pass) list
    std::vector<uint32_t> itemOf((size_t)G * S, 0xFFFFFFFFu);   // (g, s) -> item, assigned in super-tile order below
    std::vector<uint8_t> used((size_t)G * S, 0);
    for (uint32_t p = 0; p < P; ++p)
        for (uint64_t e = hd.offsets[p]; e < hd.offsets[p + 1]; ++e)
            used[(size_t)(rowPos[(size_t)p * 16 + hd.row[e]] / TM) * S + slotOf[hd.col[e]] / TN] = 1;
    for (uint32_t i = 0; i < G * S; ++i) {   // the order of gemmItemPlace, macro-tiles without entries left out
        uint32_t g, st;
        gemmItemPlace(i, G, S, g, st);
        if (!used[(size_t)g * S + st]) continue;
        itemOf[(size_t)g * S + st] = (uint32_t)out.items.size();
        GemmItem it;
        it.group = g;
        it.firstBlock = st * NB;
        it.listBase = (uint32_t)(out.items.size() * (L + 1));
        it.pad = 0;
        out.items.push_back(it);
    }
    out.fullGrid = out.items.size() == (size_t)G * S;
    const size_t I = out.items.size();
    out.numTiles = (uint64_t)I * PM * NB;
    if (I * (size_t)(L + 1) > 0xFFFFFFFFull || I * (size_t)TM > 0xFFFFFFFFull) return BSMR_ERR_INVALID_ARG;
    out.lists.assign(I * (L + 1), 0);
    out.rowStart.assign(I * TM, 0xFFFFFFFFu);
    auto locate = [&](uint32_t p, uint64_t e, size_t& item, uint32_t& list, uint32_t& slot, uint32_t& rowInWave, uint32_t& rowInTile) {
        const uint32_t place = rowPos[(size_t)p * 16 + hd.row[e]];
        const uint32_t g = place / TM, pj = (place % TM) / 16u;      // row group, 16-row tile pj of the macro-tile
        const uint32_t slotCol = slotOf[hd.col[e]];
        const uint32_t cb = slotCol >> 4, s = cb / NB, bj = cb % NB;  // block bj of the macro-tile
        const uint32_t wm = pj / m, tm = pj % m, wn = bj / n, tn = bj % n;
        const uint32_t t = tm * n + tn, q = t / kGemmPassTiles, tp = t % kGemmPassTiles;
        const uint32_t r = place & 15u, c = slotCol & 15u;
        item = itemOf[(size_t)g * S + s];
        list = (wm * kGemmWavesN + wn) * Q + q;
        slot = gemmSlabSlot(tp, r, c);
        rowInWave = tm * 16 + r;
        rowInTile = pj * 16 + r;
    };
    for (uint32_t p = 0; p < P; ++p)
        for (uint64_t e = hd.offsets[p]; e < hd.offsets[p + 1]; ++e) {
            size_t item; uint32_t list, slot, rw, rt;
            locate(p, e, item, list, slot, rw, rt);
            ++out.lists[item * (L + 1) + list + 1];
            uint32_t& rs = out.rowStart[item * TM + rt];
            rs = std::min(rs, hd.idx[e]);
        }
    for (uint32_t& v : out.rowStart)
        if (v == 0xFFFFFFFFu) v = 0;
    // 2. prefix sums: the lists of the launch are one run of `words`, each padded to a multiple of 4
    uint64_t at = 0;
    for (size_t item = 0; item < I; ++item) {
        uint32_t* ls = &out.lists[item * (L + 1)];
        uint64_t run = at;
        for (uint32_t l = 0; l <= L; ++l) {
            const uint32_t cnt = ls[l];   // ls[0] is 0, ls[l + 1] held list l's count
            out.maxListWords = std::max(out.maxListWords, (cnt + 3u) & ~3u);
            run += (cnt + 3u) & ~3u;
            if (run > 0xFFFFFFF0ull) return BSMR_ERR_INVALID_ARG;
            ls[l] = (uint32_t)run;        // start of list l (ls[L]: the end of the item's words)
        }
        at = run;
    }
    if (at > 0xFFFFFFF0ull) return BSMR_ERR_INVALID_ARG;
    out.words.assign((size_t)at + kGemmWordSlack, kGemmNoEntry);
    // 3. the words; HostDense lists a panel's entries by (column, row): collect per list, then order by (row, column)
    std::vector<uint32_t> fill(out.lists.size());
    for (size_t item = 0; item < I; ++item)
        for (uint32_t l = 0; l < L; ++l) fill[item * (L + 1) + l] = out.lists[item * (L + 1) + l];
    std::vector<uint64_t> keys(hd.entries());   // (row in wave, column, word) packed for the per-list sort
    for (uint32_t p = 0; p < P; ++p)
        for (uint64_t e = hd.offsets[p]; e < hd.offsets[p + 1]; ++e) {
            size_t item; uint32_t list, slot, rw, rt;
            locate(p, e, item, list, slot, rw, rt);
            const uint32_t off = hd.idx[e] - out.rowStart[item * TM + rt];
            if (off >= kGemmMaxOffset) return BSMR_ERR_BAD_PLAN;
            const uint32_t word = slot | (rw << 12) | (off << 19);
            uint32_t& pos = fill[item * (L + 1) + list];
            // key: row in wave (7 bits) | offset in the row's run of P (13 bits) | word: a list in (row, CSR position) order
            keys[e] = ((uint64_t)rw << 45) | ((uint64_t)off << 32) | word;
            out.words[pos++] = (uint32_t)e;   // (the entry's number for now: replaced by the sorted words below)
        }
    std::vector<uint64_t> scratch;
    for (size_t item = 0; item < I; ++item)
        for (uint32_t l = 0; l < L; ++l) {
            const uint32_t b = out.lists[item * (L + 1) + l], e = fill[item * (L + 1) + l];
            scratch.resize(e - b);
            for (uint32_t i = b; i < e; ++i) scratch[i - b] = keys[out.words[i]];
            std::sort(scratch.begin(), scratch.end());
            for (uint32_t i = b; i < e; ++i) out.words[i] = (uint32_t)scratch[i - b];
        }
    return BSMR_OK;
}

}  // namespace bsmr
