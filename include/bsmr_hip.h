/*
 * bsmr_hip.h -- C ABI of libbsmr_hip.so, the MI355X (gfx950) device side of the
 * BSMR-SDDMM engine.
 *
 * This is the drop-in boundary for the reference's device layer.  What it
 * replaces in CX9898/BSMR-SDDMM (paths relative to the reference checkout):
 *
 *   bsmr_plan_create     <- the twelve h2d() uploads at the end of RPHM::RPHM
 *                           (src/BSMR.cpp:252-264); takes the same host arrays
 *                           the reference's RPHM holds (include/BSMR.hpp:133-158)
 *   bsmr_plan_create_ex  <- the same with every construction rule passed explicitly (bsmr_plan_options)
 *   bsmr_plan_destroy    <- ~RPHM / dev::vector destructors (include/devVector.cuh:54-126)
 *   bsmr_sddmm           <- sddmm_gpu(M,N,K, A_dev, B_dev, rphm, P_dev, logger) and
 *                           sddmm_gpu_k32(...) (include/sddmmKernel.cuh:25-39,
 *                           src/sddmmKernel.cu:2540-2762): device pointers in,
 *                           P written in S's CSR order
 *   bsmr_sddmm_timed     <- the 10x event-timed loop inside those launchers
 *                           (src/sddmmKernel.cu:2561-2565,2650-2653)
 *   bsmr_sddmm_host      <- sddmm_gpu(Matrix A, Matrix B, rphm, CSR P, logger)
 *                           (src/sddmmKernel.cu:2518-2538): host operands in, host P out
 *   bsmr_sddmm_batch / bsmr_batched_transpose
 *                        <- sddmm_gpu_batch / batchedMatrixTranspose (include/sddmmKernel.cuh:41-51,
 *                           src/sddmmKernel.cu:2764-2869, 2486-2515)
 *   bsmr_cluster_rows    <- bsa_rowReordering_gpu (src/rowReordering.cu:1027-1095): the row
 *                           clustering on the device, same row order and cluster count
 *   bsmr_col_reorder*    <- colReordering_cpu + the index arrays of RPHM::RPHM (src/colReordering.cu:274-404,
 *                           src/BSMR.cpp:83-265) on the device; the reference's colReordering_gpu (:146-241) is unfinished
 *   bsmr_convert_operands / bsmr_sddmm_lowp
 *                        <- no reference counterpart (the reference converts to
 *                           TF32 in registers); lets a caller that already holds
 *                           fp16/bf16 operands skip the per-call conversion
 *   bsmr_sharded_*       <- no reference counterpart (the reference is single-GPU): row-range shards over the
 *                           GPUs of one node, one RCCL gather-v of P per step
 *   bsmr_mem_info        <- cudaMemGetInfo in calculateBlockSize (src/rowReordering.cu:1010-1013)
 *   bsmr_dev_alloc / bsmr_dev_free / bsmr_memcpy_h2d / bsmr_memcpy_d2h / bsmr_dev_memset
 *                        <- dev::vector<T> ctor/dtor and h2d()/d2h() (include/devVector.cuh:54-126,
 *                           170-229): raw device buffers owned by the caller
 *
 * Only PODs and raw pointers cross.  Every function returns a status code and
 * never throws or aborts.  A plan is bound to one device; calls on one plan
 * must come from one thread at a time.  `stream` is a hipStream_t passed as
 * void* (NULL = the default stream); all device work of a call is ordered on it.
 */
#ifndef BSMR_HIP_H
#define BSMR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSMR_OK                 0
#define BSMR_ERR_INVALID_ARG    1  /* NULL pointer, inconsistent sizes            */
#define BSMR_ERR_NO_DEVICE      2  /* no usable gfx950 device / bad device index  */
#define BSMR_ERR_HIP            3  /* a HIP runtime call failed (see last error)  */
#define BSMR_ERR_UNSUPPORTED_K  4  /* K == 0 or K not a multiple of 32            */
#define BSMR_ERR_OOM            5  /* host or device allocation failed            */
#define BSMR_ERR_BAD_PLAN       6  /* index arrays violate the RPHM invariants    */

/* Arithmetic of the dense-block path (the sparse residual path is always fp32). */
#define BSMR_COMPUTE_F16   0  /* operands rounded RNE to fp16, fp32 accumulate (default;
                                 same 10-bit mantissa as the reference's TF32)       */
#define BSMR_COMPUTE_BF16  1  /* operands rounded RNE to bf16, fp32 accumulate         */
#define BSMR_COMPUTE_F32   2  /* exact fp32 (v_mfma_f32_16x16x4_f32), 1/16 MFMA rate   */

typedef struct bsmr_plan bsmr_plan;

/* Host-side index arrays of one reordered matrix: exactly the members of the
 * reference's RPHM before upload.  All arrays are read during plan_create and
 * need not outlive it. */
typedef struct bsmr_rphm_desc {
    uint32_t M, N, nnz;
    uint32_t num_row_panels;          /* ceil(num_nonzero_rows / 16)                          */
    uint32_t num_nonzero_rows;        /* length of reordered_rows                             */
    const uint32_t *reordered_rows;   /* [num_nonzero_rows] original row of each panel row    */
    const uint32_t *dense_cols;       /* [16 * block_offsets[P]] column ids, N = padding      */
    const uint32_t *block_offsets;    /* [P+1] dense 16x16 blocks before each panel           */
    const uint32_t *block_values;     /* [256 * block_offsets[P]] CSR index or 0xFFFFFFFF,
                                         row-major 16x16 per block                            */
    const uint32_t *sparse_value_offsets; /* [P+1] sparse entries before each panel           */
    const uint32_t *sparse_values;        /* CSR index of each sparse entry                   */
    const uint32_t *sparse_relative_rows; /* row inside the panel, 0..15                      */
    const uint32_t *sparse_col_indices;   /* column id                                        */
} bsmr_rphm_desc;

typedef struct bsmr_plan_stats {
    uint32_t num_row_panels;
    uint64_t num_dense_blocks;    /* 16-column blocks of the grouped device format */
    uint64_t num_dense_entries;   /* nnz covered by the dense path  */
    uint64_t num_sparse_entries;  /* nnz covered by the sparse path */
    uint64_t dense_work_items;    /* waves launched by the dense kernel        */
    uint64_t sparse_work_items;   /* workgroups launched by the sparse kernel  */
    uint64_t device_index_bytes;  /* bytes of plan metadata resident in HBM    */
    uint32_t group_size;          /* row panels whose dense columns share one B gather (1, 2, 4) */
    uint64_t num_dense_tiles;     /* non-empty 16x16 (panel, block) tiles = MFMA tiles executed   */
    uint64_t union_columns;       /* B columns gathered per SDDMM by the dense path               */
    /* second dense format (4 panels per group) kept for gather-bound calls; 0 = not built */
    uint32_t grouped_group_size;
    uint64_t grouped_dense_tiles;
    uint64_t grouped_union_columns;
    uint64_t sparse_lowp;             /* 1: in the F16/BF16 modes the residue reads the converted operands too (for every K;
                                       * bsmr_plan_sparse_choice reports calls that do so from a K upwards) */
    uint64_t folded_dense_entries;    /* entries of a small dense part (RPHM) that the plan computes with the residue */
    uint64_t free_residue;            /* 1: residue entries run in global column order, not per panel */
    uint64_t promoted_sparse_entries; /* residue entries (RPHM) that the plan computes as extra dense blocks */
} bsmr_plan_stats;

/* 1 when the promotion rule of this plan ran as kernels (bsmr_plan_options.promote_on_device), 0 when on the host. */
int bsmr_plan_promoted_on_device(const bsmr_plan *plan, int *yes);

/* Kernel timings of the last bsmr_sddmm_timed call, milliseconds per iteration. */
typedef struct bsmr_timing {
    float total_ms;    /* convert + dense + sparse, wall on the stream          */
    float convert_ms;  /* fp32 -> fp16/bf16 operand pass (0 in F32 mode)        */
    float dense_ms;    /* dense-block MFMA kernel                               */
    float sparse_ms;   /* residual sparse kernel                                */
} bsmr_timing;

const char *bsmr_strerror(int status);
/* Text of the last HIP error seen by this library on the calling thread. */
const char *bsmr_last_hip_error(void);

int bsmr_device_count(int *count);
int bsmr_mem_info(int device, size_t *free_bytes, size_t *total_bytes);
int bsmr_device_name(int device, char *buf, size_t buflen);

/* Caller-owned device buffers on `device` (synchronous copies). */
int bsmr_dev_alloc(int device, size_t bytes, void **out);
int bsmr_dev_free(void *ptr);
int bsmr_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int bsmr_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
int bsmr_dev_memset(void *dst_dev, int value, size_t bytes);
int bsmr_device_synchronize(int device);

/* Every rule of plan construction that is not a function of the RPHM arrays.  Zero-initialise, call
 * bsmr_plan_options_default, change what is needed, pass to bsmr_plan_create_ex: a plan's layout, kernels and
 * precision class are then a function of (desc, options) alone.  Fields named like the BSMR_<NAME> environment
 * variables of bsmr_plan_options_from_env (the override used by tests, the CLI and bsmr_plan_create). */
#define BSMR_ENGINE_STREAM  0  /* dense part: per-panel streaming kernels (denseStream / denseGroups), the default */
#define BSMR_ENGINE_TILES   1  /* dense part: "tiles" format, H panels per wave-private B image (denseTiles)     */
#define BSMR_ENGINE_SHARED  2  /* dense part: "tiles" format, B images shared by the 4 waves of a workgroup        */
#define BSMR_ENGINE_SWEEP   4  /* dense part computed like a GEMM: row groups x strips of B in natural column order,
                                  B streamed once by loader waves, no gather (denseSweep, csrc/sweep_kernels.hpp)  */
#define BSMR_ENGINE_GEMM    5  /* dense part as an output-stationary masked GEMM (round 4): a workgroup owns a macro-tile of C
                                  (row panels x natural 16-column blocks), accumulators stay in registers over the K loop,
                                  A rows and B columns arrive K-slice by K-slice through an LDS double buffer, every wave
                                  multiplies an m x n block of 16 x 16 tiles (denseGemm, csrc/gemm_kernels.hpp)           */
#define BSMR_ENGINE_TUNED   3  /* the plan keeps what every engine needs; calls use the streaming engine until
                                  bsmr_plan_tune has timed the three for their (K, mode) and then the fastest          */
typedef struct bsmr_plan_options {
    uint32_t struct_size;           /* sizeof(bsmr_plan_options) of the caller                                        */
    int32_t  dense_engine;          /* BSMR_ENGINE_*                                                    [DENSE_ENGINE] */
    /* which kernel computes an entry (the RPHM's own dense / sparse split is never changed) */
    int32_t  fold_dense_below;      /* a dense part with fewer entries runs with the residue (32768)  [FOLD_DENSE_BELOW] */
    int32_t  promote_average;       /* residue -> extra dense blocks when a panel averages this many
                                       entries per 16 columns (16; 0 = never)                          [PROMOTE_AVERAGE] */
    int32_t  promote_min_entries_k; /* ... plans without a dense part: only if this many thousand move (1000) [PROMOTE_MIN_ENTRIES_K] */
    int32_t  promote_column_degree; /* ... only patterns with nnz / N at least this (32)          [PROMOTE_COLUMN_DEGREE] */
    int32_t  promote_head;          /* ... leading blocks of other panels with this many entries (0 = none) [PROMOTE_HEAD] */
    /* device format of the dense part, streaming engine */
    int32_t  dense_group;           /* panels per group: 0 = both formats, chosen per call; 1, 2, 4        [DENSE_GROUP] */
    int32_t  dense_blocks_per_item; /* 0 = from the plan's shape                                  [DENSE_BLOCKS_PER_WG] */
    int32_t  stream_waves;          /* 1 or 4 waves per workgroup of denseStream (1)                       [STREAM_WAVES] */
    int32_t  output_mode;           /* 0 = 16/32-bit offsets, 1 = 8-bit windows (default), 2 = windows staged in LDS [OUTPUT_MODE] */
    int32_t  force_tile32;          /* 32-bit destination tiles                                            [FORCE_TILE32] */
    int32_t  column_order;          /* blocks / items / residue in column order (1)                        [COLUMN_ORDER] */
    int32_t  dense_stream;          /* streaming kernel for ungrouped formats (1)                          [DENSE_STREAM] */
    int32_t  dense_batch;           /* blocks per LDS batch of denseGroups at K = 128 (0 = 4)               [DENSE_BATCH] */
    /* device format of the dense part, tiles / shared engines */
    int32_t  tile_group;            /* panels per group, 0 = cost model                                      [TILE_GROUP] */
    int32_t  tile_blocks_per_item;  /* 0 = from the plan's shape                                            [TILE_BLOCKS] */
    int32_t  tile_depth;            /* ring depth of denseTiles, 0 = default                                 [TILE_DEPTH] */
    /* residue */
    int32_t  sparse_entries_per_item; /* (256)                                                [SPARSE_ENTRIES_PER_WG] */
    int32_t  sparse_lowp;           /* residue from the fp16/bf16 copies whenever they exist (1)             [SPARSE_LOWP] */
    int32_t  sparse_lpe;            /* lanes per entry: 0 = tuned shape per K; 4, 8, 16                        [SPARSE_LPE] */
    int32_t  free_residue;          /* 1: entries in global column order instead of per panel (0)          [FREE_RESIDUE] */
    /* operand conversion */
    int32_t  convert_in_kernel;     /* -1 = by size of the dense part; 0 / 1                           [CONVERT_IN_KERNEL] */
    int32_t  convert_sliced;        /* B converted by the XCD that gathers it while it fits the L2s (1)  [CONVERT_SLICED] */
    int32_t  b_only;                /* plans without a dense part may convert B alone (1)                         [B_ONLY] */
    int32_t  b_only_work_m;         /* ... from this many million residue entries x K (100)                [B_ONLY_WORK_M] */
    /* launch */
    int32_t  overlap_streams;       /* dense and residue kernels of a hybrid plan on two streams joined by events:
                                       -1 = when both parts are large enough, 0 = never, 1 = always      [OVERLAP_STREAMS] */
    int32_t  mask_tiles;            /* destination tiles as 16-bit column masks + first offsets (48 instead of 256 bytes
                                       per tile; needs consecutive entries per tile row): 1 = whenever possible, 0 = never,
                                       -1 = when the 8-bit tiles would exceed the 32 MiB of L2 (default)     [MASK_TILES] */
    /* plan build */
    int32_t  pack_on_device;        /* the dense part's device format built by kernels from the uploaded RPHM arrays
                                       (csrc/pack_device.hpp; byte-identical to the host packer, which still does every
                                       layout but the default one): 1 = whenever possible, 0 = never,
                                       -1 = from 4096 dense blocks (default)                               [PACK_ON_DEVICE] */
    /* device format of the dense part, sweep engine (fields added in round 3: older callers' struct_size leaves the defaults) */
    int32_t  sweep_panels;          /* panels per consumer wave (a row group is 4 x this many panels): 0 = by K; 1, 2, 4
                                                                                                              [SWEEP_PANELS] */
    int32_t  sweep_strip_blocks;    /* 16-column blocks of B per work item: 0 = from the plan's shape; <= 63   [SWEEP_BLOCKS] */
    int32_t  sweep_fp32;            /* sweep kernel on the caller's fp32 operands, rounded in registers (no conversion
                                       pass; a residue then runs its fp32 kernel): -1 = for K <= 128 when the plan has no
                                       residue (default), 0 = never, 1 = whenever K <= 128                     [SWEEP_FP32] */
    int32_t  sweep_waves;           /* consumer waves per workgroup: 0 = 4; 4, 8                                [SWEEP_WAVES] */
    int32_t  sweep_per_cu;          /* workgroups per CU the LDS ring is sized for: 0 = 1; 1, 2 (2: four consumer waves only)
                                                                                                              [SWEEP_PER_CU] */
    int32_t  k_hint;                /* 0 (default): the plan-time rules above (promotion of the residue, folding of a small
                                       dense part) are thresholds fitted once, K unknown.  > 0: the inner dimension the plan
                                       will mostly be called with - the plan keeps a copy of the RPHM arrays, and
                                       bsmr_plan_tune builds it under the other settings of those rules too
                                       (BSMR_VARIANT_*), times whole calls and lets the fastest serve the plan  [K_HINT] */
    int32_t  promote_on_device;     /* the promotion rule as kernels (csrc/promote_device.hpp), its blocks handed to the device
                                       packer without crossing PCIe: -1 = from 2^20 residue entries when the plan's layout
                                       and engine allow, 0 = never, 1 = whenever they allow.  Same plan, byte for byte
                                                                                                         [PROMOTE_ON_DEVICE] */
    /* device format of the dense part, GEMM engine (fields added in round 4) */
    int32_t  gemm_panels;           /* row panels per macro-tile: 0 = from the plan's shape (a model of the launch's rounds);
                                       8, 16                                                                   [GEMM_PANELS] */
    int32_t  gemm_blocks;           /* 16-column blocks of B per macro-tile: 0 = from the plan's shape; 12, 16, 20 (with 16
                                       panels), 16, 20 (with 8)                                                 [GEMM_BLOCKS] */
    int32_t  gemm_fp32;             /* GEMM kernel on the caller's fp32 operands, rounded in registers (K = 64, 128; no
                                       conversion pass; a residue then runs its fp32 kernel): -1 = when the plan has no
                                       residue (default), 0 = never, 1 = whenever K allows                        [GEMM_FP32] */
    int32_t  gemm_balance_columns;  /* 1 (default): the columns of B are dealt over the column strips by their number of dense
                                       entries, so that every macro-tile's mask epilogue holds about as many entries (hot
                                       columns - graphs, bag-of-words - otherwise gather in a few macro-tiles the launch waits
                                       for); 0: natural column order                                   [GEMM_BALANCE_COLUMNS] */
    int32_t  evict_wide_rows;       /* 1 (default): when a (block, row) of the dense part spans 255 or more entries of its row
                                       (a long row with many residue entries between two dense columns of its panel), the
                                       entries outside the densest window of 254 become residue, so that the plan keeps
                                       its 8-bit tiles; only while that moves at most 1 / 16 of the dense entries.  0: such
                                       a plan takes direct 16-bit offsets for all blocks                  [EVICT_WIDE_ROWS] */
} bsmr_plan_options;
int bsmr_plan_options_default(bsmr_plan_options *opt);
/* defaults, then every BSMR_<NAME> variable that is set */
int bsmr_plan_options_from_env(bsmr_plan_options *opt);

/* bsmr_plan_create = bsmr_plan_create_ex with bsmr_plan_options_from_env.  options == NULL: the defaults. */
int bsmr_plan_create(bsmr_plan **out, int device, const bsmr_rphm_desc *desc);
int bsmr_plan_create_ex(bsmr_plan **out, int device, const bsmr_rphm_desc *desc,
                        const bsmr_plan_options *options);
int bsmr_plan_destroy(bsmr_plan *plan);
int bsmr_plan_get_stats(const bsmr_plan *plan, bsmr_plan_stats *out);

/* Host wall time bsmr_plan_create[_ex] spent, milliseconds: the plan rules (promotion, folding), packing the
 * device format (csrc/plan_pack.hpp), allocating + uploading it, and packing + uploading the second (grouped)
 * dense format where one is kept.  The reference has no counterpart (its RPHM constructor uploads the host
 * arrays as they are, src/BSMR.cpp:236-262); reported so that callers can see what a re-plan costs. */
typedef struct bsmr_plan_build_ms {
    float rules_ms, pack_ms, upload_ms, second_format_ms, total_ms;
} bsmr_plan_build_ms;
int bsmr_plan_build_times(const bsmr_plan *plan, bsmr_plan_build_ms *out);

/* The per-call choices are measured, not modelled: bsmr_plan_tune times, on the caller's operands (HIP events on
 * `stream`, 3 warm-up + 10 timed launches each), the dense part under each engine and format, for all-sparse plans the
 * fp32 residue against the B-only conversion, for hybrid plans one stream against two - and the plan uses the fastest
 * of each for every later call with the same (K, mode); the fitted rules (grouped format from 400 MB of gathers,
 * b_only_work_m, overlap from 4 M entries a side) only serve calls that were never tuned.
 * Needs a plan created with dense_engine = BSMR_ENGINE_TUNED; P is overwritten with the (correct) result.  The
 * reference has no counterpart: it tunes (alpha, delta) per matrix by sweeping (src/sddmm.cu:62-118). */
typedef struct bsmr_tune_report {
    int32_t chosen_engine;            /* BSMR_ENGINE_STREAM / _TILES / _SHARED / _SWEEP / _GEMM */
    int32_t chosen_group;             /* panels per group of the winner (streaming engine: 1, or 4 = the grouped format;
                                         sweep engine: panels per consumer wave) */
    int32_t chosen_blocks_per_item;   /* shared-B engine: blocks per work item where that was part of the search, else 0;
                                         sweep engine: blocks per strip */
    /* best dense-kernel time per launch of each candidate family, microseconds; < 0: not measured (no such format,
     * the engine does not serve this K, or nothing to choose from) */
    float   stream_us, grouped_us, tiles_us, shared_us;
    /* plans without a dense part: whole call with the fp32 residue kernel against B converted alone + 16-bit residue
     * (replaces the b_only_work_m threshold for this (K, mode)); -1 / < 0: does not apply */
    int32_t chosen_b_only;
    float   fp32_residue_us, b_only_us;
    /* hybrid plans: whole call on one stream against the residue kernel on the side stream; -1 / < 0: does not apply */
    int32_t chosen_overlap;
    float   one_stream_us, two_streams_us;
    /* K = 32 / 64: whole call with the conversion pass against the streaming dense kernel that reads the fp32 operands and
     * rounds them in registers (same results); -1 / < 0: does not apply */
    int32_t chosen_cvt_in_kernel;
    float   convert_pass_us, fp32_dense_us;
    /* sweep engine (round 3; appended): best dense-kernel time on 16-bit operands (comparable with stream_us ...), and whole
     * calls: the best 16-bit engine behind the conversion pass against the sweep kernel on the fp32 operands */
    float   sweep_us;
    float   lowp_call_us, sweep_fp32_call_us;
    int32_t chosen_sweep_fp32;        /* 1: the chosen engine is the sweep kernel on fp32 operands */
    /* plans created with k_hint > 0 (appended): whole-call time of the plan under each setting of the plan-time rules,
     * microseconds (< 0: not built - the same split as an earlier variant, or no hint), and the one that serves the plan */
    int32_t chosen_variant;           /* BSMR_VARIANT_* */
    float   variant_us[6];
    /* GEMM engine (round 4; appended): best dense-kernel time on 16-bit operands over the macro-tile shapes tried; when it is
     * chosen, chosen_group = panels and chosen_blocks_per_item = 16-column blocks per macro-tile */
    float   gemm_us;
    /* K = 64 / 128, whole calls: the GEMM kernel on the caller's fp32 operands (no conversion pass) against the best of the
     * rest (lowp_call_us, or sweep_fp32_call_us where that had won) */
    float   gemm_fp32_call_us;
    int32_t chosen_gemm_fp32;         /* 1: the chosen engine is the GEMM kernel on fp32 operands */
} bsmr_tune_report;
#define BSMR_VARIANT_RULES        0   /* the options the plan was created with                                    */
#define BSMR_VARIANT_AS_RPHM      1   /* the RPHM's split as it is: nothing promoted, nothing folded              */
#define BSMR_VARIANT_NO_PROMOTION 2   /* promote_average = 0                                                      */
#define BSMR_VARIANT_PROMOTE_24   3   /* promote_average = 24                                                     */
#define BSMR_VARIANT_PROMOTE_ALL  4   /* every panel's residue becomes blocks                                     */
#define BSMR_VARIANT_ALL_RESIDUE  5   /* the dense part folded into the residue                                   */
int bsmr_plan_tune(bsmr_plan *plan, uint32_t K, const float *A_dev, const float *B_dev, float *P_dev, int mode, void *stream,
                   bsmr_tune_report *report /* may be NULL */);
/* The same for callers compiled against another revision of this header: bsmr_tune_report only ever grows at its end, and at
 * most report_size bytes of it are written (bsmr_plan_tune writes sizeof(bsmr_tune_report) of THIS revision: a caller built
 * against an older header must call bsmr_plan_tune_sized with its own sizeof, INTEGRATION.md "ABI revisions"). */
int bsmr_plan_tune_sized(bsmr_plan *plan, uint32_t K, const float *A_dev, const float *B_dev, float *P_dev, int mode, void *stream,
                         bsmr_tune_report *report /* may be NULL */, size_t report_size);

/* What a tuned plan keeps per (K, mode), as plain numbers: bsmr_plan_get_tuned reads the choice bsmr_plan_tune made
 * (BSMR_ERR_INVALID_ARG when that (K, mode) was never tuned), bsmr_plan_set_tuned installs one WITHOUT timing anything -
 * a second process (a profiler pass, a later run on the same matrix) replays the measured choice of the first and launches
 * exactly its kernels.  The choice is validated (the engine must serve this K and its device format must be buildable:
 * BSMR_ERR_INVALID_ARG / BSMR_ERR_BAD_PLAN otherwise, the plan unchanged).  Plans created with k_hint keep their variant:
 * these two calls address the engines of the plan that serves the calls.  The reference has no counterpart. */
typedef struct bsmr_tuned_choice {
    uint32_t struct_size;     /* sizeof(bsmr_tuned_choice) of the caller */
    int32_t  engine;          /* BSMR_ENGINE_STREAM / _TILES / _SHARED / _SWEEP / _GEMM */
    int32_t  group;           /* tiles / shared: panels per group; sweep: panels per consumer wave; gemm: macro-tile rows / 16; 0 = the engine's rule */
    int32_t  blocks_per_item; /* shared: blocks per work item; sweep: blocks per strip; gemm: macro-tile columns / 16; 0 = the engine's rule */
    int32_t  format;          /* streaming engine: 0 = one panel per group, 1 = the grouped format, -1 = the rule */
    int32_t  b_only;          /* all-sparse plans: 1 = B converted alone, 0 = fp32 residue, -1 = the rule */
    int32_t  overlap;         /* hybrid plans: 1 = residue on the side stream, 0 = one stream, -1 = as the plan was built */
    int32_t  cvt_in_kernel;   /* 1 = fp32 operands rounded inside the dense kernel, 0 = conversion pass, -1 = the rule */
    int32_t  waves;           /* sweep: consumer waves; 0 = the rule */
} bsmr_tuned_choice;
int bsmr_plan_get_tuned(const bsmr_plan *plan, uint32_t K, int mode, bsmr_tuned_choice *out);
int bsmr_plan_set_tuned(bsmr_plan *plan, uint32_t K, int mode, const bsmr_tuned_choice *choice);

/* Fingerprint of the plan's first dense format as it lies in device memory: FNV-1a of groupRows, rowBase, winLen,
 * winMask, blockCols, the destination tiles (8-bit or mask form), blockMask and the work items, then the numbers of
 * items / blocks / tiles / union columns and 1 if the tiles are in the mask form (out[0..12]).  Two plans with equal
 * fingerprints launch identical dense work; used to check the device packer against the host packer. */
int bsmr_plan_format_digest(const bsmr_plan *plan, uint64_t out[13]);
/* Fingerprint of the dense entry lists a plan keeps for the formats of the tiles / shared / sweep / GEMM engines (plans
 * created with one of these engines or BSMR_ENGINE_TUNED): FNV-1a of the panels' offsets, the entries' columns, rows in
 * panel and CSR indices (out[0..3]), the number of entries (out[4]).  All zero for a plan that keeps none.  The lists are
 * written by the device packer (csrc/pack_device.hpp: collectEmit) or by the host (csrc/tile_format.hpp: collectDense);
 * used to check the one against the other. */
int bsmr_plan_entry_lists_digest(const bsmr_plan *plan, uint64_t out[5]);
/* Which dense format a call with inner dimension K uses (any out pointer may be NULL):
 * panels per group, MFMA tiles executed, B columns gathered.  For a plan whose engines were measured (bsmr_plan_tune)
 * the answer describes the engine of the call that was prepared last - ask right after the bsmr_sddmm call in question
 * (the untuned rules depend on K alone). */
int bsmr_plan_dense_choice(const bsmr_plan *plan, uint32_t K, uint32_t *group_size,
                           uint64_t *tiles, uint64_t *union_columns);

/* Row clustering of the BSMR pipeline on the device: bsa_rowReordering_gpu
 * (src/rowReordering.cu:1027-1095 = calculateDispersion :478-501 + get_permutation_gpu
 * :893-1007 + the removal of empty rows :1082-1090).  S as host CSR; reordered_rows has
 * room for `rows` ids and receives the non-empty rows in clustered order; num_clusters is
 * the value the reference logs as bsmr_numClusters.  Results are identical to the host
 * implementation behind BSMR::rowReordering.  BSMR_ERR_OOM: the rows x bins table does not
 * fit in half of the free device memory (or one row's histogram does not fit in LDS) - the
 * caller falls back to the host implementation. */
typedef struct bsmr_cluster_stats {
    float    elapsed_ms;        /* device time, uploads and downloads included        */
    uint32_t passes;            /* speculative passes launched that did work          */
    uint32_t similarities;      /* (representative, row) pairs judged                 */
    uint32_t exact_similarities;/* ... of which within 1e-4 of alpha (exact evaluation) */
    uint32_t threads_per_pair;  /* workgroup size = the reference's clustering block  */
    uint64_t table_bytes;       /* rows x bins histogram table                        */
    uint32_t dropped_seeds;     /* clusters started ahead of the older ones' decisions whose seed an older one took */
    uint32_t passes_ahead;      /* passes in which clusters judged rows the older clusters had not decided yet    */
} bsmr_cluster_stats;
int bsmr_cluster_rows(int device, uint32_t rows, uint32_t cols, const uint32_t *row_offsets,
                      const uint32_t *col_indices, uint32_t bin_width, float alpha,
                      uint32_t *reordered_rows, uint32_t *num_reordered, int32_t *num_clusters,
                      bsmr_cluster_stats *stats);
/* ... for callers compiled against another revision of this header (bsmr_cluster_stats only grows at its end): at most
 * stats_size bytes of it are written. */
int bsmr_cluster_rows_sized(int device, uint32_t rows, uint32_t cols, const uint32_t *row_offsets,
                            const uint32_t *col_indices, uint32_t bin_width, float alpha,
                            uint32_t *reordered_rows, uint32_t *num_reordered, int32_t *num_clusters,
                            bsmr_cluster_stats *stats, size_t stats_size);

/* Column reordering, dense / sparse split and the RPHM index arrays of the BSMR pipeline on the device:
 * colReordering_cpu (src/colReordering.cu:274-404; the reference's own GPU version, :146-241, is unfinished) followed by
 * the array part of RPHM::RPHM (src/BSMR.cpp:83-265).  S as host CSR, reordered_rows as BSMR::rowReordering returns
 * them (the non-empty rows in clustered order).  The results equal the host implementation's array for array:
 * bsmr_col_reorder computes them on `device`, bsmr_col_reorder_sizes tells how much room they need, and
 * bsmr_col_reorder_fetch copies them into the caller's arrays (any pointer may be NULL). */
typedef struct bsmr_colreorder bsmr_colreorder;
typedef struct bsmr_colreorder_sizes {
    uint32_t num_row_panels;
    uint64_t num_dense_cols;      /* length of denseCols  (multiples of 16 per panel)          */
    uint64_t num_sparse_cols;     /* length of sparseCols (padding sentinels included)          */
    uint64_t num_blocks;          /* dense 16 x 16 blocks: blockValues holds 256 words of each  */
    uint64_t num_sparse_entries;  /* length of sparseValues / sparseRelativeRows / sparseColIndices */
    float    elapsed_ms;          /* device time, uploads included                              */
} bsmr_colreorder_sizes;
int bsmr_col_reorder(bsmr_colreorder **out, int device, uint32_t rows, uint32_t cols, const uint32_t *row_offsets,
                     const uint32_t *col_indices, const uint32_t *reordered_rows, uint32_t num_reordered, float delta);
int bsmr_col_reorder_sizes(const bsmr_colreorder *h, bsmr_colreorder_sizes *out);
int bsmr_col_reorder_device(const bsmr_colreorder *h, int *device);   /* where the result lies */
int bsmr_col_reorder_fetch(const bsmr_colreorder *h, uint32_t *dense_cols, uint32_t *dense_col_offsets,
                           uint32_t *sparse_cols, uint32_t *sparse_col_offsets, uint32_t *sparse_value_offsets,
                           uint32_t *block_offsets, uint32_t *block_values, uint32_t *sparse_values,
                           uint32_t *sparse_relative_rows, uint32_t *sparse_col_indices);
int bsmr_col_reorder_free(bsmr_colreorder *h);
/* The plan straight from a bsmr_col_reorder result, on its device: the RPHM's big arrays (block values, the residue) are
 * read where they lie - the promotion rule and the device packer run on them - and only what host-side steps need comes
 * down.  The plan is the one bsmr_plan_create_ex builds from the fetched arrays, byte for byte; plans that need the arrays
 * on the host (engines other than the streaming one, layouts of the host packer, k_hint) fetch them and take that road.
 * reordered_rows as given to bsmr_col_reorder; options == NULL: bsmr_plan_options_from_env. */
int bsmr_plan_create_from_colreorder(bsmr_plan **out, const bsmr_colreorder *h, uint32_t M, uint32_t N, uint32_t nnz,
                                     const uint32_t *reordered_rows, uint32_t num_nonzero_rows,
                                     const bsmr_plan_options *options);

/* How the sparse residue of a call (K, compute_mode) will run: lanes that share one entry's
 * K-long dot product (the fp32 summation order of the residue depends on it; the CPU twin in
 * oracle/sddmm_oracle.c takes the same number) and whether it reads the fp16/bf16 copies. */
int bsmr_plan_sparse_choice(const bsmr_plan *plan, uint32_t K, int compute_mode,
                            uint32_t *lanes_per_entry, uint32_t *low_precision);

/* Which kernel computes each stored entry: flags_host[i] (i = CSR index, nnz bytes, host memory) becomes 1
 * when the plan computes entry i on the dense (MFMA) path and 0 when the residue kernel does.  This is the
 * reference's dense / sparse split (RPHM::blockValues_ / sparseValues_, reference src/BSMR.cpp:83-265) after the plan's own moves
 * (folded_dense_entries, promoted_sparse_entries); the accuracy contract of an entry follows its path. */
int bsmr_plan_dense_flags(const bsmr_plan *plan, uint8_t *flags_host);

/* Grow the plan's operand workspace for inner dimension K now (otherwise it
 * grows on first use, which allocates and therefore must not happen inside a
 * stream capture). */
int bsmr_plan_reserve(bsmr_plan *plan, uint32_t K);

/* One SDDMM: A_dev is M x K row-major fp32, B_dev is K x N column-major fp32
 * (column j = K contiguous floats at B_dev + j*K), P_dev receives nnz floats in
 * S's CSR order.  Every element of P_dev is overwritten. */
int bsmr_sddmm(bsmr_plan *plan, uint32_t K, const float *A_dev, const float *B_dev,
               float *P_dev, int compute_mode, void *stream);

/* warmup + iters repetitions of bsmr_sddmm bracketed by HIP events on `stream`;
 * per-kernel times are measured in a second pass of `iters` repetitions with
 * events around each kernel. */
int bsmr_sddmm_timed(bsmr_plan *plan, uint32_t K, const float *A_dev, const float *B_dev,
                     float *P_dev, int compute_mode, void *stream, int warmup, int iters,
                     bsmr_timing *out);

/* sddmm_gpu_batch (include/sddmmKernel.cuh:41-47, src/sddmmKernel.cu:2764-2869): num_batches
 * independent problems over one sparsity plan, stored back to back - A: [b][M][K] row-major,
 * B: [b][N][K] (each K x N column-major), P: [b][nnz].  One conversion pass over all batches, then
 * the dense and residue kernels with the batch index in the grid's y dimension. */
int bsmr_sddmm_batch(bsmr_plan *plan, uint32_t K, const float *A_dev, const float *B_dev, float *P_dev,
                     uint32_t num_batches, int compute_mode, void *stream);

/* batchedMatrixTranspose (include/sddmmKernel.cuh:49-51, src/sddmmKernel.cu:2486-2515):
 * out[b] = transpose(in[b]) for num_batches row-major height x width matrices. */
int bsmr_batched_transpose(uint32_t width, uint32_t height, uint32_t num_batches, const float *in_dev,
                           float *out_dev, void *stream);

/* Convert fp32 operands once (mode F16 or BF16); A16_dev / B16_dev are M*K and
 * N*K 16-bit elements in the same layouts. */
int bsmr_convert_operands(bsmr_plan *plan, uint32_t K, const float *A_dev, const float *B_dev,
                          void *A16_dev, void *B16_dev, int compute_mode, void *stream);

/* SDDMM on pre-converted operands.  The dense path reads A16/B16.  The sparse residue reads
 * them too when bsmr_plan_stats.sparse_lowp is 1 (A_dev/B_dev may then be NULL); otherwise it
 * needs the fp32 A_dev/B_dev. */
int bsmr_sddmm_lowp(bsmr_plan *plan, uint32_t K, const void *A16_dev, const void *B16_dev,
                    const float *A_dev, const float *B_dev, float *P_dev, int compute_mode,
                    void *stream);

/* ---- one SDDMM over several GPUs of one node, from one process (the reference is single-GPU) ----
 * Row panels are independent, so the rows of S are cut into contiguous ranges (the caller's cost partition) and
 * every range is a problem of its own: shard_descs[i] is the RPHM of rows [row_begin[i], row_begin[i+1]) with LOCAL
 * row ids (built by the host pipeline on that slice), devices[i] the GPU it lives on.  A step runs bsmr_sddmm on
 * every device and gathers the compact outputs to devices[0] with one RCCL send/recv group (communicators are
 * created once, in bsmr_sharded_create, over the distinct devices of the list); P in S's CSR order is the concatenation
 * of the shards' outputs.  A device may be listed more than once (more shards than GPUs): a shard on the root's device
 * hands its part over with a device-to-device copy.  A range without rows or entries takes no part in a step. */
typedef struct bsmr_sharded bsmr_sharded;
typedef struct bsmr_sharded_timing {
    float    step_ms;      /* device time of one step (SDDMM on every device + gather), max over devices; the steps are
                              pipelined over two sets of output buffers: the gather of step i runs behind the SDDMM of i + 1 */
    float    wall_ms;      /* host wall time of one step                                                  */
    uint32_t num_devices;
    /* (round 4; appended - ABI revision 4) one un-pipelined step taken apart: */
    float    compute_ms;   /* SDDMM alone, max over devices                                               */
    float    gather_ms;    /* the gather-v alone: first part leaving to last part delivered               */
} bsmr_sharded_timing;
/* Revision of this header's struct layouts (4 = round 4: bsmr_sharded_timing, bsmr_tune_report, bsmr_cluster_stats and
 * bsmr_plan_options grew at their ends).  A caller built against an older revision uses the *_sized entry points for the
 * structs the library writes, or is rebuilt. */
#define BSMR_ABI_REVISION 4
int bsmr_abi_revision(void);
int bsmr_sharded_create(bsmr_sharded **out, const int *devices, uint32_t num_devices,
                        const bsmr_rphm_desc *const *shard_descs, const uint32_t *row_begin,
                        const bsmr_plan_options *options);
int bsmr_sharded_destroy(bsmr_sharded *s);
/* total stored entries (length of P) and, if per_shard != NULL, every shard's share */
int bsmr_sharded_num_entries(const bsmr_sharded *s, uint64_t *total, uint64_t *per_shard);
/* Host operands in (A: all rows, row-major M x K; B: K x N column-major), host P out (S's CSR order): uploads every
 * shard's rows of A and a replica of B, one warm-up step, `iters` timed steps, download from the root. */
int bsmr_sharded_sddmm_host(bsmr_sharded *s, uint32_t K, const float *A_host, const float *B_host, float *P_host,
                            int compute_mode, int iters, bsmr_sharded_timing *timing);

/* Host operands in, host P out (upload, `iters` timed repetitions after one
 * warm-up, download).  ms_per_iter may be NULL. */
int bsmr_sddmm_host(bsmr_plan *plan, uint32_t K, const float *A_host, const float *B_host,
                    float *P_host, int compute_mode, int iters, float *ms_per_iter);

#ifdef __cplusplus
}
#endif
#endif /* BSMR_HIP_H */
