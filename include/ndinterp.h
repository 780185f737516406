/* include/ndinterp.h -- C ABI of libndinterp_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the batched `interp_array` hot path of the Rust crate
 * ndarray-interp v0.6.0 (1D Linear, 1D CubicSpline, 2D Bilinear).  The reference
 * has no FFI of its own: its boundary is the strategy trait pair plus the inherent
 * methods of Interp1D / Interp2D.  Each entry point below names the reference
 * interface it replaces (paths relative to the reference tree).  A Rust
 * `extern "C"` block binding exactly these symbols, and the strategy overrides that
 * call them, are shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C types only; no exceptions, no panics cross the ABI; every call
 *    returns an ndi_status (ndi_last_error_string() has the text for the calling
 *    thread).
 *  - element type T is selected per handle by ndi_dtype (f32 / f64 only: the GPU
 *    path covers the float types; integer element types stay on the host's generic
 *    per-query path).
 *  - arrays are C-order and contiguous: data[n][lanes], data2d[nx][ny][lanes];
 *    "lanes" = product of the trailing (non-interpolated) axes.
 *  - every pointer argument carries a memory space (host or device).  Device
 *    pointers must belong to the handle's device.
 *  - the product has NO CPU fallback: without a usable HIP device every compute
 *    entry point fails with NDI_HIP_ERROR.
 */
#ifndef NDINTERP_H
#define NDINTERP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDI_VERSION_MAJOR 0
#define NDI_VERSION_MINOR 5

/* BuilderError / InterpolateError (src/lib.rs:127-146) + ABI-only codes. */
typedef enum ndi_status {
  NDI_OK = 0,
  NDI_NOT_ENOUGH_DATA = 1, /* BuilderError::NotEnoughData                                  */
  NDI_MONOTONIC = 2,       /* BuilderError::Monotonic                                      */
  NDI_SHAPE = 3,           /* BuilderError::ShapeError                                     */
  NDI_VALUE = 4,           /* BuilderError::ValueError (periodic y[0] != y[n-1])           */
  NDI_OUT_OF_BOUNDS = 5,   /* InterpolateError::OutOfBounds                                */
  NDI_NAN_QUERY = 6,       /* reference panics: vector_extensions.rs:83-84                 */
  NDI_HIP_ERROR = 7,       /* no device / HIP runtime failure                              */
  NDI_BAD_ARG = 8,
  NDI_UNSUPPORTED = 9
} ndi_status;

typedef enum ndi_dtype { NDI_F32 = 0, NDI_F64 = 1 } ndi_dtype;
typedef enum ndi_memspace { NDI_MEM_HOST = 0, NDI_MEM_DEVICE = 1 } ndi_memspace;

/* 1D strategies: Linear (src/interp1d/strategies/linear.rs),
 * CubicSpline (src/interp1d/strategies/cubic_spline.rs). */
typedef enum ndi_strategy1d { NDI_LINEAR = 0, NDI_CUBIC_SPLINE = 1 } ndi_strategy1d;

/* SingleBoundary (cubic_spline.rs:204-217).  Natural == SecondDeriv(0),
 * Clamped == FirstDeriv(0) (:287-296). */
typedef enum ndi_bc_kind {
  NDI_BC_NOT_A_KNOT = 0,
  NDI_BC_NATURAL = 1,
  NDI_BC_CLAMPED = 2,
  NDI_BC_FIRST_DERIV = 3,
  NDI_BC_SECOND_DERIV = 4
} ndi_bc_kind;

typedef struct ndi_boundary {
  int32_t kind; /* ndi_bc_kind */
  double value; /* derivative value for FIRST_DERIV / SECOND_DERIV */
} ndi_boundary;

/* Monotonic (src/vector_extensions.rs:25-29). */
typedef enum ndi_monotonic {
  NDI_MONO_NOT = 0,
  NDI_MONO_RISING_STRICT = 1,
  NDI_MONO_RISING = 2,
  NDI_MONO_FALLING_STRICT = 3,
  NDI_MONO_FALLING = 4
} ndi_monotonic;

/* Evaluation formulation (results are identical):
 *  GATHER   one coalesced row gather per query (4 operand rows in, 1 row out);
 *  BUCKETED queries are grouped by interval on the device so each table row is
 *           read once per group and the kernel becomes a pure output stream;
 *  AUTO     BUCKETED when the batch has enough queries per interval (>= 5) to pay for
 *           the grouping pass, else GATHER. */
typedef enum ndi_path { NDI_PATH_AUTO = 0, NDI_PATH_GATHER = 1, NDI_PATH_BUCKETED = 2 } ndi_path;

/* CubicSpline::build (cubic_spline.rs:310-368, 409-721) -- numerical contract of the coefficient tables.
 * The serial per-lane kernels evaluate thomas (:678-721) in the reference's operation order without contraction: the
 * a / b tables are BIT-IDENTICAL to the reference's.  For narrow trailing axes on many knots (n >= 2048 and
 * lanes <= 256: scalar data on 1e5-1e6 knots, 8 lanes on 4096) that would be one or two wavefronts doing 2n dependent
 * steps, so by default such builds -- on axes whose neighbouring knot spacings differ by less than 1e3 (f32) / 1e9
 * (f64); wilder axes keep the serial kernels -- take blocked sweeps: both first-order recurrences are cut into blocks and
 * re-associated (back substitution as r'/mid' + (-up/mid') k).  Every coefficient then agrees with the reference's to
 * within 1e-12 (f64) / 1e-5 (f32) of the larger of: the magnitudes of the table entries within 32 rows of it, and the
 * interval's |dy| -- errors do not travel (the recurrences' multipliers are <= 1/2 in magnitude) -- and evaluated
 * rows meet the crate's own assertion form at 1e-10 / 1e-5 (tests/test_gpu_spline_blocked.py, incl. geometric and
 * clustered knots).  NDI_BUILD_REFERENCE_ORDER keeps the serial kernels for the handle: bit-identical tables at the
 * serial kernels' speed. */
typedef enum ndi_build_flags {
  NDI_BUILD_DEFAULT = 0,
  NDI_BUILD_REFERENCE_ORDER = 1 /* never re-associate the Thomas sweeps: tables bit-identical to the reference's */
} ndi_build_flags;

/* Replaces Interp1DBuilder::{new,x,strategy,build} (src/interp1d/mod.rs:399-476)
 * + Interp1DStrategyBuilder::build (src/interp1d/strategies/mod.rs:12-40):
 * Linear::build (linear.rs:54-63) / CubicSpline::build (cubic_spline.rs:754-771). */
typedef struct ndi_interp1d_desc {
  int32_t dtype;       /* ndi_dtype */
  int32_t strategy;    /* ndi_strategy1d */
  int32_t extrapolate; /* Linear::extrapolate / CubicSpline::extrapolate (bool) */
  int32_t device;      /* HIP device ordinal */
  uint64_t n;          /* data.shape()[0] */
  uint64_t lanes;      /* product of data.shape()[1..] (1 for 1-D data) */
  uint64_t x_len;      /* x.len(); checked against n exactly as build() does (:465-471) */
  const void* x;       /* T[x_len] knots, or NULL for the default axis 0..n (:402-406) */
  const void* data;    /* T[n * lanes] */
  int32_t memspace;    /* ndi_memspace of x and data */
  int32_t validate;    /* != 0: run Interp1DBuilder::build's checks (:449-471) here */
  /* CubicSpline boundary (BoundaryCondition, cubic_spline.rs:153-168) */
  int32_t periodic;    /* BoundaryCondition::Periodic */
  int32_t build_flags; /* ndi_build_flags (0 = default); occupies what was alignment padding in v0.3: same layout */
  ndi_boundary left;   /* applied to every lane unless lane_* are given */
  ndi_boundary right;
  /* BoundaryCondition::Individual (per trailing element, cubic_spline.rs:332-347):
   * arrays of `lanes` entries, host memory; all four NULL for a global boundary. */
  const int32_t* lane_left_kind;
  const double* lane_left_value;
  const int32_t* lane_right_kind;
  const double* lane_right_value;
} ndi_interp1d_desc;

/* Replaces Interp2DBuilder::{new,x,y,strategy,build} (src/interp2d/mod.rs:382-519)
 * + Bilinear::build (src/interp2d/strategies/bilinear.rs:45-52).
 * Device memory: one copy of the grid; for short trailing axes (lanes * sizeof(T) <= 64 bytes) the copy
 * is kept pair-packed instead ({z[xi][yi], z[xi][yi+1]} per cell), which takes twice the grid size. */
typedef struct ndi_interp2d_desc {
  int32_t dtype;
  int32_t extrapolate; /* Bilinear::extrapolate */
  int32_t device;
  int32_t memspace;    /* of x, y, data */
  uint64_t nx, ny;     /* data.shape()[0], data.shape()[1] */
  uint64_t lanes;      /* product of data.shape()[2..] */
  uint64_t x_len, y_len;
  const void* x;       /* T[x_len] or NULL for 0..nx (:389-393) */
  const void* y;       /* T[y_len] or NULL for 0..ny (:394-398) */
  const void* data;    /* T[nx * ny * lanes] */
  int32_t validate;    /* != 0: run Interp2DBuilder::build's checks (:477-509) here */
  int32_t reserved;
} ndi_interp2d_desc;

typedef struct ndi_interp1d ndi_interp1d; /* owns device copies of x, data (and a, b) */
typedef struct ndi_interp2d ndi_interp2d;

/* First failing query of a batch: the reference's query loop stops at the first Err
 * (src/interp1d/mod.rs:334-342, src/interp2d/mod.rs:297-306).  `index` is the lowest
 * flat query index that failed, `value` the offending coordinate and `axis` 0 for x,
 * 1 for y (x is tested before y for the same query: bilinear.rs:71-80), so the host
 * can format the reference's message ("x = {x:#?} is not in range"). */
typedef struct ndi_oob_info {
  uint64_t index;
  double value;
  int32_t axis;
  int32_t status; /* NDI_OUT_OF_BOUNDS or NDI_NAN_QUERY */
} ndi_oob_info;

typedef struct ndi_eval_opts {
  int32_t q_memspace;   /* ndi_memspace of the query array(s) */
  int32_t out_memspace; /* ndi_memspace of the output buffer */
  void* stream;         /* hipStream_t the kernels are enqueued on; NULL = the HIP default stream
                           (hipStreamPerThread is accepted like any other handle) */
  int32_t path;         /* ndi_path */
  int32_t async_launch; /* != 0 (device out only): enqueue and return; fetch the batch
                           status later with ndi_interp{1,2}d_finish on the same stream (from the
                           same host thread); the query array(s) must stay valid until then */
  int32_t flags;        /* ndi_eval_flags */
  int32_t reserved;     /* 0 */
} ndi_eval_opts;

/* NDI_EVAL_FRESH_OUTPUT: the output buffer was allocated for this call and is dropped if the call fails -- what
 * Interp1D::interp_array / Interp2D::interp_array do (src/interp1d/mod.rs:197-211: `zeros(..)`, `?` on Err).  The
 * library may then write rows at / after the first failing query; status, index and value of the failure are reported as
 * always.  Without the flag (interp_array_into semantics: a caller-owned buffer) rows at / after the first failing query
 * are left untouched, as the reference's serial loop leaves them (:334-342) -- which costs the short-row kernels a
 * pre-pass over the queries (8-16 bytes per query, a quarter of the time of a scalar batch). */
/* NDI_EVAL_ROWS_AFTER_ERROR_UNSPECIFIED (v0.5): an interp_array_into caller's opt-in to the same kernels on a buffer it
 * OWNS -- if the call fails, rows at / after the first failing query hold unspecified values (rows before it are the
 * reference's, the failure report is unchanged).  The reference leaves those rows untouched (:334-342); a caller that
 * discards or overwrites the buffer on Err anyway gets the pre-pass back: scalar f64 data 200 -> 280 Gqueries/s.
 * Unknown flag bits and a non-zero `reserved` are refused with NDI_BAD_ARG (v0.5): a binary built against an older,
 * shorter ndi_eval_opts cannot silently select an option. */
typedef enum ndi_eval_flags {
  NDI_EVAL_DEFAULT = 0,
  NDI_EVAL_FRESH_OUTPUT = 1,
  NDI_EVAL_ROWS_AFTER_ERROR_UNSPECIFIED = 2
} ndi_eval_flags;

/* ---- build ------------------------------------------------------------------ */
ndi_status ndi_interp1d_create(const ndi_interp1d_desc* desc, ndi_interp1d** out);
void ndi_interp1d_destroy(ndi_interp1d* h);
ndi_status ndi_interp2d_create(const ndi_interp2d_desc* desc, ndi_interp2d** out);
void ndi_interp2d_destroy(ndi_interp2d* h);

/* A replica of a built interpolator on `device` (any device, the handle's own included): the device-resident knots /
 * data / spline tables are copied device to device (between two GPUs: over xGMI) -- the caller's arrays are not
 * uploaded again and CubicSpline::build's Thomas solve (cubic_spline.rs:754-771) is not repeated.  What the sharded
 * calls below take as `handles`: knots / coefficients replicated per device.  The replica is independent of the
 * original (destroy each with ndi_interp{1,2}d_destroy). */
ndi_status ndi_interp1d_clone(const ndi_interp1d* h, int32_t device, ndi_interp1d** out);
ndi_status ndi_interp2d_clone(const ndi_interp2d* h, int32_t device, ndi_interp2d** out);

/* CubicSplineStrategy{a, b} (cubic_spline.rs:94-102): copies the coefficient tables,
 * each T[(n-1) * lanes], to `a_out` / `b_out` (either may be NULL). */
ndi_status ndi_interp1d_coefficients(const ndi_interp1d* h, void* a_out, void* b_out,
                                     int32_t memspace);

/* ---- evaluate ----------------------------------------------------------------
 * Replaces Interp1D::interp_array_into for a flattened query array
 * (src/interp1d/mod.rs:272-343) with Linear::interp_into (linear.rs:73-98) or
 * CubicSplineStrategy::interp_into (cubic_spline.rs:791-830) as the per-query body:
 *   out[i * out_row_stride + l] = strategy(q[i])[l],  i < nq, l < lanes.
 * out_row_stride is in elements (>= lanes).  On NDI_OUT_OF_BOUNDS / NDI_NAN_QUERY
 * rows before info->index are written and later rows are untouched, as in the
 * reference.  General-rank queries are a flatten on the caller's side. */
ndi_status ndi_interp1d_eval(const ndi_interp1d* h, const void* q, uint64_t nq, void* out,
                             uint64_t out_row_stride, const ndi_eval_opts* opts,
                             ndi_oob_info* info);

/* Replaces Interp2D::interp_array_into (src/interp2d/mod.rs:215-307) with
 * Bilinear::interp_into (bilinear.rs:64-99) as the per-query body. */
ndi_status ndi_interp2d_eval(const ndi_interp2d* h, const void* qx, const void* qy, uint64_t nq,
                             void* out, uint64_t out_row_stride, const ndi_eval_opts* opts,
                             ndi_oob_info* info);

/* Completes async_launch evaluations on `stream`: synchronises it and reports the
 * status of the last batch enqueued there. */
ndi_status ndi_interp1d_finish(const ndi_interp1d* h, void* stream, ndi_oob_info* info);
ndi_status ndi_interp2d_finish(const ndi_interp2d* h, void* stream, ndi_oob_info* info);

/* ---- chunked evaluation through a device-output ring ---------------------------------
 * Replaces Interp1D::interp_array (src/interp1d/mod.rs:197-211: allocate zeros(xs.shape ++ lanes), then the
 * query loop :326-343) for batches whose whole output does not fit, or need not stay, in device memory
 * (4096 lanes x 1e7 queries of f64 = 327.7 GB > 288 GB HBM).  The flattened queries are evaluated in chunks of
 * `chunk_queries` rows into a ring of `n_slots` device buffers; after a chunk's kernels are enqueued the
 * `consume` callback is called on the calling host thread with the chunk's location, and the slot is reused
 * n_slots chunks later.  Nothing is copied to the host.
 *
 * Slot hand-off is stream-ordered: a consumer that enqueues its work on chunk->stream needs nothing else and
 * returns NULL; a consumer that works on another stream records a hipEvent_t there when it is done with the slot
 * and returns it -- the library makes its stream wait for that event before the slot is overwritten.
 *
 * First-error semantics are the reference's: a range pre-pass over all queries finds the lowest failing flat
 * index F before any chunk is produced; exactly the rows [0, F) are produced (the last chunk is cut short),
 * then NDI_OUT_OF_BOUNDS / NDI_NAN_QUERY is returned with info filled in.  The call returns after the last
 * chunk's kernels (and the consumer's stream-ordered work) have completed.
 *
 * Internally the producer is a two-stream pipeline: the search + grouping of chunk k+1 run on a side stream the
 * handle owns while chunk k is evaluated on opts->stream (event-ordered; two scratch sets).  Everything the
 * consumer can observe -- the chunk's rows and chunk->stream -- is on opts->stream. */
typedef struct ndi_ring_chunk {
  uint64_t index;      /* chunk number, 0, 1, ... */
  uint64_t q_begin;    /* flat index of the chunk's first query */
  uint64_t q_count;    /* rows produced in this chunk */
  void* out;           /* device pointer, T[q_count][row_stride] */
  uint64_t row_stride; /* elements */
  uint32_t slot;       /* ring slot the chunk lives in */
  uint32_t shard;      /* sharded evaluation: index of the handle (device) that produced the chunk; else 0 */
  void* stream;        /* hipStream_t the chunk's kernels were enqueued on */
} ndi_ring_chunk;

typedef void* (*ndi_ring_consumer)(void* user, const ndi_ring_chunk* chunk);

/* Ring layout.  slots[i] points at row 0 of slot i; row r of a chunk is at slots[i] + r * row_stride elements.
 * Slots may be separate buffers (row_stride >= lanes), but on MI355X the recommended layout is ONE allocation with
 * the slots interleaved row by row: slots[i] = base + i * lanes, row_stride = n_slots * lanes.  The rate at which
 * a kernel streams into a 32.8 GB extent depends on where that extent lies in physical memory (up to 27 %
 * between the slots of one allocation); striping every chunk over the whole ring removes the dependence
 * (DESIGN.md 4.3).  With slots == NULL the library owns the ring (allocated once per handle, kept until trim /
 * destroy) and uses exactly that layout: chunk->row_stride is then n_slots * max(row_stride, lanes). */
typedef struct ndi_ring_desc {
  void* const* slots;     /* n_slots device pointers, or NULL for a library-owned ring */
  uint32_t n_slots;       /* >= 1 */
  uint32_t reserved;
  uint64_t chunk_queries; /* rows per chunk */
  uint64_t row_stride;    /* elements between consecutive rows of a slot (>= lanes; 0 = lanes) */
} ndi_ring_desc;

ndi_status ndi_interp1d_eval_ring(const ndi_interp1d* h, const void* q, uint64_t nq,
                                  const ndi_ring_desc* ring, ndi_ring_consumer consume, void* user,
                                  const ndi_eval_opts* opts, ndi_oob_info* info);
/* Same for Interp2D::interp_array (src/interp2d/mod.rs:175-196, 287-307). */
ndi_status ndi_interp2d_eval_ring(const ndi_interp2d* h, const void* qx, const void* qy, uint64_t nq,
                                  const ndi_ring_desc* ring, ndi_ring_consumer consume, void* user,
                                  const ndi_eval_opts* opts, ndi_oob_info* info);

/* ---- sharded evaluation: one call, several devices -------------------------------------------------
 * The reference is single-threaded; its multi-worker shape is one interpolator driven from many threads over
 * contiguous blocks of the query array (benches/bench_interp1d.rs:49-79; rayon support "planned", README.md:16-18).
 * These calls are that shape below the host language: `handles` are n_shards replicas of one interpolator (same
 * knots / data / strategy; normally one per device, built with ndi_interp{1,2}d_create and desc.device = d), the
 * flattened query array is split into contiguous blocks -- shard i owns [lo_i, hi_i) = ndi_shard_bounds(nq, i,
 * n_shards), block sizes differ by at most one -- and every shard is evaluated by its own host thread on its
 * handle's device: shard 0 on the calling thread, shard i > 0 on the i-th persistent worker thread the library keeps
 * for that calling thread (so a handle's per-thread scratch is reused from call to call, and concurrent sharded calls
 * from different host threads do not share workers).  No device-to-device traffic, no collective.
 *
 * First-error semantics are the reference's serial loop (src/interp1d/mod.rs:326-343, src/interp2d/mod.rs:287-307)
 * over the WHOLE batch: every shard range-checks its block first, the shards agree on the minimum failing flat
 * index F at a host barrier, and exactly the rows [0, F) are produced -- rows at or after F are never written, in
 * any shard.  info->index is the global flat index.
 *
 * ndi_shard_io (one per shard):
 *   q / qy  NULL: the shard reads its block of the flattened q (qx, qy) -- these are then host arrays or memory
 *           every device can read; non-NULL: the shard's own block (hi_i - lo_i queries), e.g. already resident on
 *           the shard's device.  Memory space of either form: opts->q_memspace.
 *   out     the shard's rows, T[hi_i - lo_i][out_row_stride], memory space opts->out_memspace (device pointers
 *           belong to the shard's device).  For one host output array: out = base + lo_i * out_row_stride.
 *   stream  hipStream_t on the shard's device (NULL = its default stream); opts->stream is ignored.
 * The handles must be distinct.  opts->async_launch is ignored (the call returns when every shard has finished). */
typedef struct ndi_shard_io {
  const void* q;
  const void* qy;
  void* out;
  void* stream;
} ndi_shard_io;

void ndi_shard_bounds(uint64_t nq, uint32_t shard, uint32_t n_shards, uint64_t* lo, uint64_t* hi);

ndi_status ndi_interp1d_eval_sharded(const ndi_interp1d* const* handles, uint32_t n_shards, const void* q,
                                     uint64_t nq, const ndi_shard_io* io, uint64_t out_row_stride,
                                     const ndi_eval_opts* opts, ndi_oob_info* info);
ndi_status ndi_interp2d_eval_sharded(const ndi_interp2d* const* handles, uint32_t n_shards, const void* qx,
                                     const void* qy, uint64_t nq, const ndi_shard_io* io, uint64_t out_row_stride,
                                     const ndi_eval_opts* opts, ndi_oob_info* info);

/* The same through one device-output ring per shard (rings[i] on handles[i]'s device; slots == NULL: the handle
 * owns it): what ndi_interp{1,2}d_eval_ring is to one device.  `io` may be NULL (out is unused).  `consume` is
 * called from the shards' host threads CONCURRENTLY (once per chunk, in order within a shard); chunk->shard names
 * the shard, chunk->q_begin is the global flat index of the chunk's first query, chunk->index counts within the
 * shard.  Exactly the rows [0, F) are handed out. */
ndi_status ndi_interp1d_eval_ring_sharded(const ndi_interp1d* const* handles, uint32_t n_shards, const void* q,
                                          uint64_t nq, const ndi_shard_io* io, const ndi_ring_desc* rings,
                                          ndi_ring_consumer consume, void* user, const ndi_eval_opts* opts,
                                          ndi_oob_info* info);
ndi_status ndi_interp2d_eval_ring_sharded(const ndi_interp2d* const* handles, uint32_t n_shards, const void* qx,
                                          const void* qy, uint64_t nq, const ndi_shard_io* io,
                                          const ndi_ring_desc* rings, ndi_ring_consumer consume, void* user,
                                          const ndi_eval_opts* opts, ndi_oob_info* info);

/* Per-(stream, host thread) scratch is cached on the handle (at most 16 idle sets are kept; the least recently
 * used idle set is freed beyond that).  trim frees every idle set and a library-owned ring now. */
ndi_status ndi_interp1d_trim(const ndi_interp1d* h);
ndi_status ndi_interp2d_trim(const ndi_interp2d* h);
/* Diagnostic: number of scratch sets currently cached on the handle. */
uint64_t ndi_interp1d_scratch_sets(const ndi_interp1d* h);

/* ---- helpers on the path ------------------------------------------------------ */
/* VectorExtensions::get_lower_index (src/vector_extensions.rs:55-111), batched:
 * out_idx[i] = the unique j with knots[j] <= q[i] < knots[j+1], clamped to [0, n-2];
 * -1 for a NaN query.  Device search (wavefront-cooperative, knots in LDS). */
ndi_status ndi_get_lower_index_batch(int32_t dtype, int32_t device, const void* knots, uint64_t n,
                                     const void* q, uint64_t nq, int64_t* out_idx,
                                     int32_t memspace);

/* The same search with the knot pyramid resident on the device (no per-call allocation or knot upload):
 * what Interp1D::get_index_left_of (src/interp1d/mod.rs:380-382) is to a built interpolator.
 * `stream` as in ndi_eval_opts; the call returns when out_idx is complete. */
typedef struct ndi_locator ndi_locator;
ndi_status ndi_locator_create(int32_t dtype, int32_t device, const void* knots, uint64_t n,
                              int32_t memspace, ndi_locator** out);
ndi_status ndi_locator_eval(const ndi_locator* h, const void* q, uint64_t nq, int64_t* out_idx,
                            int32_t memspace, void* stream);
void ndi_locator_destroy(ndi_locator* h);

/* VectorExtensions::monotonic_prop (src/vector_extensions.rs:40-53, 116-198).
 * Host-side O(n) validation; returns an ndi_monotonic. */
int32_t ndi_monotonic_prop(int32_t dtype, const void* host_v, uint64_t n);

/* Interp1DBuilder::build / Interp2DBuilder::build checks alone (host). */
ndi_status ndi_validate1d(int32_t dtype, const void* host_x, uint64_t x_len, uint64_t n,
                          int32_t strategy);
ndi_status ndi_validate2d(int32_t dtype, const void* host_x, uint64_t x_len, const void* host_y,
                          uint64_t y_len, uint64_t nx, uint64_t ny);

/* ---- library-owned output buffers ---------------------------------------------------------------------------
 * Replaces the `Array::zeros(..)` of Interp1D::interp_array / Interp2D::interp_array (src/interp1d/mod.rs:204-209,
 * src/interp2d/mod.rs:183-188): a device buffer of `bytes` zero bytes for a batch's output rows.  On MI355X the rate at
 * which rows stream into a multi-gigabyte buffer depends on the physical pages behind it (4.6 .. 6.1 ms per 1e6 queries of
 * BASELINE configs[1] into buffers allocated one after the other; a property of the buffer, not of order or warm-up), so
 * the zero fill -- done the way the evaluation writes, whole 32 KiB rows at scattered positions: a sequential fill is blind
 * to the difference -- is timed and a slowly filling buffer is set aside for another candidate, up to `max_tries` (0 = as many as
 * fit into half of the free memory, 2 .. 8; buffers under 1 GiB: 1), while the device has room; the best candidate is returned, the others are freed.  `info` (may be NULL)
 * reports what happened.  Free with ndi_output_free (NDI_BAD_ARG for any other pointer).  The mirrors' interp_array uses
 * it for device outputs of >= 1 GiB; interp_array_into never allocates.
 * ndi_output_free keeps up to two freed buffers of >= 1 GiB (together at most a third of the device's memory) for the next
 * request of the same size on the same device -- a fresh 32.8 GB allocation costs the allocator 0.7-1.4 s, a caller that
 * evaluates batch after batch would pay it on every call -- and ndi_output_alloc hands such a buffer out again, zeroed
 * (info->tries == 0).  ndi_output_trim releases the kept buffers.
 * `flags` = NDI_OUTPUT_UNINITIALIZED: the contents are unspecified -- for a caller that overwrites every row it will read, as
 * interp_array does (the buffer is dropped on Err, src/interp1d/mod.rs:210): a kept buffer then goes out without the refill
 * (zeroing 32.8 GB costs what evaluating into it costs); new candidates are still filled, the fill being the measurement.
 * ndi_output_free waits for the device before it keeps a buffer, as hipFree does before it releases one. */
typedef enum ndi_output_flags {
  NDI_OUTPUT_ZEROED = 0,
  NDI_OUTPUT_UNINITIALIZED = 1
} ndi_output_flags;
typedef struct ndi_output_info {
  uint32_t tries;          /* candidates allocated and filled (0: a buffer kept by ndi_output_free was reused) */
  uint32_t reserved;
  double fill_tbps;        /* zero-fill rate of the buffer returned, TB/s */
  double worst_fill_tbps;  /* ... of the slowest candidate seen */
  double alloc_ms;         /* wall time of the whole call */
} ndi_output_info;
ndi_status ndi_output_alloc(int32_t device, uint64_t bytes, uint32_t max_tries, uint32_t flags, void** out,
                            ndi_output_info* info);
ndi_status ndi_output_free(void* p);
ndi_status ndi_output_trim(void);

/* ---- runtime ------------------------------------------------------------------ */
int32_t ndi_device_count(void);
const char* ndi_last_error_string(void);
uint32_t ndi_version(void); /* (major << 16) | minor */

/* Per-kernel HIP-event timing of the evaluation stages, recorded on the stream the
 * kernels run on.  Enable, run evaluations, then read (read synchronises the events). */
typedef struct ndi_profile {
  uint64_t eval_launches;   /* dominant kernel: gather / bucketed evaluation */
  double eval_ms;           /* summed duration of those launches */
  uint64_t locate_launches; /* per-query search kernel */
  double locate_ms;
  uint64_t group_launches;  /* bucketed path only: count + scan + scatter kernels */
  double group_ms;
  int32_t last_path;        /* ndi_path actually taken by the last evaluation */
  int32_t reserved;
} ndi_profile;
void ndi_profile_enable(int32_t on);
ndi_status ndi_profile_read(ndi_profile* out, int32_t reset);

/* Measurement aid for the 2-D gather (Bilinear::interp_into, bilinear.rs:83-97): runs the evaluation kernel's memory
 * access mix alone -- per query one uniformly random cell of the handle's own grid, the four corner vectors with
 * the kernel's lane mapping, the output row stored; no searches, knots or query values -- `reps` times and returns
 * the median launch duration in *ms.  `out`: device buffer T[nq][out_row_stride] (overwritten with meaningless
 * values).  The ceiling the memory system sets for this gather on this box: bench.py reports the evaluation
 * kernel against it next to the fraction of the HBM spec peak. */
ndi_status ndi_interp2d_probe_ceiling(const ndi_interp2d* h, uint64_t nq, void* out, uint64_t out_row_stride,
                                      void* stream, int32_t reps, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* NDINTERP_H */
