/*
 * gmhip.h -- C-ABI of libgmhip.so: the MI355X (gfx950) per-tile raster compute path
 * behind dask-geomodeling's Block.process() bodies.
 *
 * The reference (nens/dask-geomodeling) is pure Python: there is no FFI to replace.
 * The drop-in boundary is the `process` staticmethod of each hot Block; every entry
 * point below is what the Python `process` body binds through ctypes *instead of*
 * its NumPy sweeps.  Reference paths are relative to /root/reference/dask_geomodeling.
 *
 * Conventions
 *   - plain C types only (pointers, sizes, PODs); no exceptions cross the ABI
 *   - every function returns 0 on success, non-zero (a hipError_t value or a
 *     GM_ERR_* code) on failure; gm_last_error() gives the message of the last
 *     failure on the calling thread
 *   - device pointers come from gm_malloc(); all work is enqueued on the calling
 *     thread's current device's compute stream and is asynchronous with respect
 *     to the host unless stated otherwise (gm_sync / gm_d2h are the sync points)
 *   - tiles are C-contiguous (bands, H, W); element-wise kernels see n = bands*H*W
 */
#ifndef GMHIP_H
#define GMHIP_H

#ifndef __HIPCC_RTC__ /* (hiprtc has no host headers: gm_vm.h declares the fixed-width types there) */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define GM_VERSION "0.1.0"

/* ---- error codes (besides hipError_t values, which are < 2000) ---------------- */
#define GM_OK 0
#define GM_ERR_INVALID 9001     /* bad argument / malformed program */
#define GM_ERR_UNSUPPORTED 9002 /* valid request the kernels do not cover */
#define GM_ERR_NO_DEVICE 9003
#define GM_ERR_GUARD 9005  /* GM_DEBUG_GUARD=1: a kernel wrote outside a block of gm_malloc (details on stderr) */
#define GM_ERR_FORKED 9004 /* this process is a fork() of one that had entered the library: see gm_init */

/* ---- element types of tiles in HBM ------------------------------------------- */
enum gm_dtype {
  GM_BOOL = 0, /* 1 byte, 0/1 */
  GM_U8 = 1,
  GM_I8 = 2,
  GM_U16 = 3,
  GM_I16 = 4,
  GM_U32 = 5,
  GM_I32 = 6,
  GM_U64 = 7,
  GM_I64 = 8,
  GM_F32 = 9,
  GM_F64 = 10,
  GM_DTYPE_COUNT = 11
};

/* 8-byte scalar slot; which member is meaningful is stated per use.  Functions take it as a
 * raw `uint64_t *_bits` payload (the 8 bytes of this union; float tiles: .f64, integer and
 * bool tiles: .i64/.u64) because unions cannot be passed by value through libffi/ctypes. */
typedef union gm_scalar {
  double f64;
  float f32;
  int64_t i64;
  uint64_t u64;
  int32_t i32;
} gm_scalar;

/* =============================================================================
 * Runtime: devices, memory, streams, events
 * ========================================================================== */

/* Initialise HIP; GM_ERR_NO_DEVICE when no GPU is visible (the library still loads).
 * fork(): a HIP runtime does not survive it.  A child forked after its parent probed for devices (any call
 * below but gm_last_error / gm_version) gets GM_ERR_FORKED -- "forked after gm_init: start workers with spawn" --
 * from every runtime call instead of a hang: dask's `processes` scheduler (core/graphs.py:42-49) must be given
 * spawned workers, as dask_geomodeling_amd.shard starts them. */
int gm_init(void);
/* Number of visible devices, or 0 when HIP cannot initialise. Never fails. */
int gm_device_count(void);
/* Select the PROCESS's device (one worker process per GPU; also creates its stream/pool): every
 * thread that enters the library afterwards -- e.g. the pool threads of dask's threaded scheduler
 * (core/graphs.py:42-49) -- is pointed at this device, not at HIP's per-thread default 0. */
int gm_set_device(int device);
int gm_get_device(int* device);
/* "domain:bus:device.function" of a visible device, without binding it: what a worker process of the one-worker-per-GPU
 * scheduler looks up under /sys/bus/pci/devices/<id>/local_cpulist to pin itself to the cores next to ITS GPU before it
 * binds the device (the reference leaves placement to dask's workers: core/graphs.py:42-49 only picks the scheduler). */
int gm_device_pci_bus_id(int device, char* buf, size_t len);
/* Device name / arch string ("gfx950...") of the current device into buf. */
int gm_device_name(char* buf, size_t len);

/* Caching device allocator (per device, mutex-guarded; blocks are recycled without
 * hipFree, safe because all work of a device is ordered on one stream). */
int gm_malloc(void** ptr, size_t bytes);
int gm_free(void* ptr);
/* n blocks of `bytes` each in one call (the result tiles of one batched launch: a boundary crossing per launch instead of
 * one per tile); all or nothing. */
int gm_malloc_many(void** ptrs, int n, size_t bytes);
int gm_pool_trim(void);                      /* hipFree all cached blocks */
/* Debug canaries.  With GM_DEBUG_GUARD=1 in the environment every block of gm_malloc is a device allocation of
 * its own between two poisoned red zones (4 KiB in front, 4 KiB behind the last byte asked for), comes filled with 0xFF bytes
 * (NaN / -1: reading what nobody wrote shows), and is never recycled.  The zones are checked at gm_free, at every gm_sync and
 * here: a violation is described on stderr, counted, and fails the call with GM_ERR_GUARD.  Slow (a synchronisation per free):
 * for the stress drivers, never for measurements.  gm_guard_check: violations counted so far (also without the mode: 0). */
int gm_guard_enabled(void);
int gm_guard_check(unsigned long long* violations_so_far);
int gm_mem_info(size_t* free_bytes, size_t* total_bytes);
int gm_pool_stats(size_t* in_use_bytes, size_t* cached_bytes);

/* Pinned host buffers (recycled by size) for results handed back to the host: gm_d2h into one runs
 * at PCIe speed, into pageable memory at a small fraction of it. */
int gm_host_malloc(void** ptr, size_t bytes);
int gm_host_free(void* ptr);
/* Pin a host range somebody else owns -- the shared-memory mosaic the one-worker-per-GPU scheduler hands to every worker
 * (raster/parallelize.py:108-124 assembles it on the host): device-to-host copies into it then run at PCIe speed and
 * straight to their place.  The range stays the caller's; unregister before unmapping. */
int gm_host_register(void* ptr, size_t bytes);
int gm_host_unregister(void* ptr);
int gm_h2d(void* dst_dev, const void* src_host, size_t bytes); /* returns after the copy */
int gm_d2h(void* dst_host, const void* src_dev, size_t bytes); /* returns after the copy */
/* `rows` rows of width_bytes from a device tile (rows src_pitch_bytes apart) to a host window (rows dst_pitch_bytes
 * apart): a result tile pasted at its place of a host mosaic (`values[:, rows, columns] = tile`, parallelize.py:118-124).
 * Asynchronous: gm_sync() is the sync point (the host range should be pinned: gm_host_malloc / gm_host_register). */
int gm_d2h_2d(void* dst_host, size_t dst_pitch_bytes, const void* src_dev, size_t src_pitch_bytes, size_t width_bytes,
              size_t rows);
int gm_d2d(void* dst_dev, const void* src_dev, size_t bytes);  /* async */
int gm_memset(void* dst_dev, int byte_value, size_t bytes);    /* async */
int gm_sync(void);                                             /* drain the current device's stream */

/* Timing on the compute stream (HIP events). */
int gm_event_create(void** event);
int gm_event_destroy(void* event);
int gm_event_record(void* event);
int gm_event_sync(void* event);
int gm_event_elapsed_ms(void* start, void* stop, float* ms);

/* Route introspection.  Several entry points choose between kernels by the shape of their input (gm_zonal_statistic among
 * six routes, the bilinear resample among three, tables and medians among two or three each).  Every choice counts itself:
 * gm_route_stats copies the first n counters of a family (process-wide, monotonic since the library was loaded; returns the
 * number of slots the family has, or -1 for an unknown family), gm_route_name names a slot (NULL past the end).  The tests
 * that force a route assert that its counter moved.  Slot 0 of every family counts the calls of its entry point(s). */
#define GM_ROUTE_ZONAL 0
#define GM_ROUTE_BILINEAR 1
#define GM_ROUTE_TABLES 2
#define GM_ROUTE_MEDIANS 3
#define GM_ROUTE_STENCILS 4
#define GM_ROUTE_FAMILIES 5
int gm_route_stats(int family, uint64_t* counters, int n);
const char* gm_route_name(int family, int slot);

/* zonal slots */
#define GM_RZ_CALLS 0
#define GM_RZ_DIRECT 1           /* count / min / max: run-compressed atomics, one pass */
#define GM_RZ_CHUNK_ATTEMPTS 2   /* sums: passes of the chunk-record kernel (a frame may take two) */
#define GM_RZ_CHUNK_ANSWERED 3   /* sums: frames the chunk records answered */
#define GM_RZ_CHUNK_GAVE_UP 4    /* sums: frames handed on by the chunk records */
#define GM_RZ_EXACT_ALL 5        /* sums: every label certified exact, parallel sums are the answer */
#define GM_RZ_EXACT_PARTIAL 6    /* sums: a few labels sorted and added in order, the others parallel */
#define GM_RZ_SORT_PARTITION 7   /* sorts by label through the counting partition */
#define GM_RZ_SORT_RADIX 8       /* sorts by label through the radix sort */
#define GM_RZ_SELECT 9           /* medians / percentiles: partition + radix select */
#define GM_RZ_LEXSORT 10         /* medians / percentiles: sort by value, then by label */
#define GM_RZ_BLOCK_PIPELINE 11  /* sums: long labels composed block-parallel */
#define GM_RZ_ORDERED_SUMS 12    /* sums: the wave-per-label kernel over sorted cells ran */
#define GM_RZ_LDS_TABLES 13      /* count / min / max: per-workgroup LDS tables (label rasters without runs) */
#define GM_RZ_SEGMENT_SORT 14    /* medians / percentiles under many labels: sort by label, every label's segment sorted in LDS */
#define GM_RZ_LDS_EXACT 15       /* sums of scattered labels that cannot round: certificates and sums through workgroup LDS tables */
#define GM_RZ_LDS_LAZY 16        /* sums of scattered labels answered by the float32 certificate over workgroup LDS tables */
#define GM_RZ_EXACT_LAZY 17      /* sums: LABELS the exponents turned away that the float32 certificate answered from the parallel sums */
#define GM_RZ_PARALLEL_LAZY 18   /* sums of rasters of thousands of small labels: parallel sums + the float32 certificate on every label, one pass */
#define GM_RZ_RAW_SELECT 19      /* medians / percentiles under a few labels: radix select straight off the raster, no partition */
#define GM_RZ_SLOTS 20
/* bilinear slots */
#define GM_RB_CALLS 0
#define GM_RB_TWO_X 1            /* the dedicated 2 x kernel */
#define GM_RB_STAGED 2           /* general kernel, source window staged in LDS */
#define GM_RB_DIRECT 3           /* general kernel, taps through the caches */
#define GM_RB_UNSTAGED_FALLBACK 4 /* general kernel, window too large for LDS */
#define GM_RB_SLOTS 5
/* table slots */
#define GM_RT_CALLS 0
#define GM_RT_LOOKUP32_W4 1      /* LOOKUP(int32) -> 1-byte tile, LDS window + ballot fast path */
#define GM_RT_LOOKUP32_OTHER 2   /* the other instantiations of the int32 lookup kernel */
#define GM_RT_LUT8 3             /* programs over one 1-byte tile as a 256-entry table */
#define GM_RT_LUT8_ROWS 4        /* ... over windows that do not sit on 16 bytes */
#define GM_RT_CLASSIFY_REGISTERS 5
#define GM_RT_CLASSIFY_LDS 6
#define GM_RT_CLASSIFY_WINDOW 7
#define GM_RT_RECLASSIFY 8
#define GM_RT_SLOTS 9
/* median slots (temporal / stack medians) */
#define GM_RM_CALLS 0
#define GM_RM_NETWORK 1
#define GM_RM_LDS_SORT 2
#define GM_RM_SELECT 3
#define GM_RM_SLOTS 4
/* stencil slots */
#define GM_RS_CALLS 0
#define GM_RS_HILLSHADE_PAIRS 1
#define GM_RS_HILLSHADE_SINGLE 2
#define GM_RS_MOVINGMAX_3X3 3
#define GM_RS_MOVINGMAX_STACK 4   /* sizes 5-15: nested horizontal maxima in registers */
#define GM_RS_MOVINGMAX_COLUMNS 5 /* larger disks, column by column */
#define GM_RS_MOVINGMAX_KEYS 6    /* reserved (the integer-key walk for windows holding a -0.0 is chosen inside the kernel: nothing on the host counts it) */
#define GM_RS_DILATE8_WIDE 7
#define GM_RS_DILATE_OTHER 8
#define GM_RS_GAUSS_ONE_LAUNCH 9
#define GM_RS_GAUSS_TWO_PASS 10
#define GM_RS_SLOTS 11

const char* gm_last_error(void);
const char* gm_version(void);

/* =============================================================================
 * K1: fused element-wise program ("VM") -- one launch per tile
 *
 * Replaces the body of wrap_math_process_func (raster/elemwise.py:235-299) for
 * Add/Subtract/Multiply/Divide/Power (:302-406), the comparison/logic/exp/log
 * members of the same wrapper (:409-548, 641-697, 784-832), Invert/IsData/IsNoData
 * (:551-638), FillNoData (:700-757), and Clip.process / Mask.process / MaskBelow /
 * Step (raster/misc.py:98-123, 208-222, 245-251, 308-328) -- and chains of them.
 *
 * A program is a straight-line accumulator program: every step computes
 *     acc = op(a, b)
 * where each operand is the accumulator, an input tile, a literal or a saved temp.
 * Every step re-applies the reference's per-op rule literally:
 *     nodata = OR over raster operands (operand == operand_ndv)   [exact ==]
 *     r = op(a, b) computed in the step's dtype
 *     r = fill where !isfinite(r) or nodata
 * so a fused chain is bit-identical to the unfused one.
 * ========================================================================== */

#define GM_MAX_INPUTS 16
#define GM_MAX_STEPS 24
#define GM_MAX_TEMPS 4
#define GM_MAX_TABLES 4
#define GM_MAX_LINEAR_EDGES 32

/* register class: the type all VM registers are held in */
enum gm_reg_class { GM_CLASS_F32 = 0, GM_CLASS_F64 = 1, GM_CLASS_I32 = 2, GM_CLASS_I64 = 3 };

enum gm_operand_kind { GM_OPD_NONE = 0, GM_OPD_ACC = 1, GM_OPD_INPUT = 2, GM_OPD_SCALAR = 3, GM_OPD_TEMP = 4 };

enum gm_opcode {
  GM_OP_NOP = 0,
  /* wrap_math_process_func family: r = ufunc(a, b) */
  GM_OP_ADD = 1, GM_OP_SUB = 2, GM_OP_MUL = 3, GM_OP_DIV = 4, GM_OP_POW = 5,
  GM_OP_EQ = 6, GM_OP_NE = 7, GM_OP_GT = 8, GM_OP_GE = 9, GM_OP_LT = 10, GM_OP_LE = 11,
  GM_OP_AND = 12, GM_OP_OR = 13, GM_OP_XOR = 14,
  GM_OP_EXP = 15, GM_OP_LOG = 16, GM_OP_LOG10 = 17,
  /* stand-alone element-wise blocks */
  GM_OP_INVERT = 18,   /* ~a (bool) */
  GM_OP_ISDATA = 19,   /* a != ndv_a */
  GM_OP_ISNODATA = 20, /* a == ndv_a */
  GM_OP_CLIP = 21,     /* a where b has data (b bool: where b), else ndv_a */
  GM_OP_MASK = 22,     /* aux where a is data (isclose rule for float tiles) else fill */
  GM_OP_MASKBELOW = 23,/* ndv_a where a < aux else a */
  GM_OP_STEP = 24,     /* left/at/right by comparing a with aux; no-data kept */
  GM_OP_FILLNODATA = 25,/* b where b is data (isclose rule) else a */
  GM_OP_COPY = 26,     /* acc = a (cast only) */
  /* table lookups inside a program (tables travel with the program: gm_program.tables) */
  GM_OP_DIGITIZE = 27, /* Classify.process (raster/misc.py:387-399): acc = number of edges e with e <= a
                          (right: e < a); NaN -> nbins; a == ndv_a -> fill.  <= GM_MAX_LINEAR_EDGES edges */
  GM_OP_LOOKUP = 28,   /* Reclassify.process (raster/misc.py:482-515): acc = LUT[a - kmin] inside the dense
                          window, else the matching exception key's target, else select ? fill : a */
  /* Max = reduce_rasters(stack, "max") (raster/reduction.py:38-119): like NumPy, carried out in a float
     class with "no data" coded as NaN */
  GM_OP_NANIFNODATA = 29, /* a where a is data (utils.get_index: isclose rule for float tiles, flags / aux1 /
                             aux2 as MASK) else NaN; flags & 8: the data value rounded to float16 (the work dtype
                             of a 1-byte output, reduction.py:99-107, over wider rasters) */
  GM_OP_FMAX = 30,        /* np.fmax(a, b) = (a > b || isnan(b)) ? a : b  -- what np.nanmax reduces with (ties: b) */
  GM_OP_NANTOFILL = 31,   /* fill where a is NaN else a */
  GM_OP_COUNT
};

/* Descriptors are made of 32/64-bit words only so that a wave reads them with scalar loads
 * (s_load_dword*) straight from the kernarg segment. */
typedef struct gm_operand {
  uint32_t bits;    /* kind (gm_operand_kind) | index << 8 | has_ndv << 16 | is_bool << 17
                       index:   input / temp index
                       has_ndv: INPUT/ACC/TEMP: the operand is a raster whose cells equal to
                                `value` are no-data (exact ==)
                       is_bool: operand is a boolean raster (never carries no-data) */
  uint32_t pad_;
  gm_scalar value;  /* SCALAR: the literal; otherwise the ndv -- in register-class representation */
} gm_operand;
#define GM_OPERAND_BITS(kind, index, has_ndv, is_bool) \
  ((uint32_t)(kind) | ((uint32_t)(index) << 8) | ((uint32_t)(has_ndv) << 16) | ((uint32_t)(is_bool) << 17))

typedef struct gm_step {
  uint32_t bits;      /* opcode (gm_opcode) | save_temp << 8 | flags << 16 | round_f32 << 24 |
                         wrap_i32 << 25 | check_finite << 26
                         save_temp:    0 = no, else 1 + temp index: also keep the result in a temp
                         flags:        MASK/FILLNODATA: 1 = isclose evaluated in f64 (else f32),
                                       2 = ndv finite, 4 = integer tile (exact != instead of isclose)
                         round_f32:    F64 class: step dtype is float32 -> round result to f32
                         wrap_i32:     I64 class: step dtype is int32 -> wrap result to 32 bits
                         check_finite: apply r[~isfinite(r)] = fill (float classes) */
  uint32_t pad_;
  gm_operand a, b;
  gm_scalar fill;     /* value written where no-data / non-finite (register-class repr) */
  gm_scalar aux0;     /* MASK: value (register-class repr); MASKBELOW/STEP: threshold */
  gm_scalar aux1;     /* MASK/FILLNODATA: .f64 isclose tolerance atol + rtol*|ndv|; STEP: left */
  gm_scalar aux2;     /* MASK/FILLNODATA: .f64 the ndv isclose compares with; STEP: at */
  gm_scalar aux3;     /* STEP: right */
  /* DIGITIZE: aux0.u64 = table index, aux1.u64 = number of edges; flags: 1 = right, 2 = the table
               holds int64 edges (integer tile, integer edges) instead of float64 edges
     LOOKUP:   aux0.u64 = table index, aux1.i64 = kmin, aux2.u64 = slots of the dense window,
               aux3.u64 = number of exception keys; flags: 1 = select.  Table: `slots` targets in the
               register class' own representation and width (unmapped slots already resolved),
               padded to 8 bytes, then one 16-byte (int64 key, target) pair per exception.  The
               class must hold every target exactly (plain Reclassify: I64 / F64; a table folded
               with a following Mask holds small integers and runs in any class).  The window is
               staged in LDS: at most 64 KiB, one LOOKUP table per program. */
} gm_step;
#define GM_STEP_BITS(opcode, save_temp, flags, round_f32, wrap_i32, check_finite)                     \
  ((uint32_t)(opcode) | ((uint32_t)(save_temp) << 8) | ((uint32_t)(flags) << 16) |                    \
   ((uint32_t)(round_f32) << 24) | ((uint32_t)(wrap_i32) << 25) | ((uint32_t)(check_finite) << 26))

typedef struct gm_input_desc {
  uint32_t bits;      /* dtype (gm_dtype of the tile in HBM) | broadcast << 8 | native_ndv << 9
                         broadcast:  single element, broadcast to every pixel
                         native_ndv: 64-bit integer tile read into the F64 class (NumPy casts it to
                                     float64 for the op, e.g. Divide(int64 raster, x)): cells equal
                                     to `ndv` -- compared as integers -- are loaded as the sentinel
                                     0.5, a value no converted integer can take; operands of this
                                     input then test against 0.5 */
  uint32_t pad_;
  gm_scalar ndv;      /* .i64 / .u64, used when native_ndv is set */
} gm_input_desc;
#define GM_INPUT_BITS(dtype, broadcast, native_ndv) \
  ((uint32_t)(dtype) | ((uint32_t)(broadcast) << 8) | ((uint32_t)(native_ndv) << 9))

/* A lookup table of a program: n_words 8-byte words in HOST memory (copied, uploaded and cached
 * together with the program on its first use). */
typedef struct gm_table {
  const void* data;
  uint64_t n_words;
} gm_table;

typedef struct gm_program {
  uint32_t n_steps;
  uint32_t n_inputs;
  uint32_t reg_class;  /* gm_reg_class */
  uint32_t out_dtype;  /* gm_dtype the final accumulator is stored as */
  gm_input_desc inputs[GM_MAX_INPUTS];
  gm_step steps[GM_MAX_STEPS];
  uint32_t n_tables;
  uint32_t pad_;
  gm_table tables[GM_MAX_TABLES];
} gm_program;

/* Run `prog` over n elements.  inputs[k] are device pointers (n elements each, or 1
 * when broadcast); out is a device pointer to n elements of prog->out_dtype.
 * The program is validated, lowered and uploaded on first use and cached by content. */
int gm_elemwise_run(const gm_program* prog, const void* const* inputs, void* out, size_t n);

/* The same program over n_tiles independent tiles of n elements each in as few launches as
 * possible (16 tiles per launch): inputs[t * prog->n_inputs + k] is input k of tile t, outs[t]
 * its result.  This is how the tiles a RasterTiler fans out (raster/parallelize.py:82-89) -- one
 * identical sub-graph per tile -- reach the GPU: kernel boundaries between tiles disappear. */
int gm_elemwise_run_batch(const gm_program* prog, int n_tiles, const void* const* inputs, void* const* outs, size_t n);

/* The same with inputs that are zero-copy WINDOWS of larger rasters resident in HBM -- what a tile
 * request into a MemorySource hands on (raster/sources.py:119-149 does a warp + copy there): every
 * tile is rows x cols elements; pitches[t * prog->n_inputs + k] is the row pitch, in elements, of
 * input k of tile t (0 or cols: contiguous); out_pitches[t] likewise for the result of tile t, so that
 * a tile can be written straight into its place in a mosaic (raster/parallelize.py:118-124 pastes
 * there).  Either array may be NULL.  16-byte vector access needs cols % 4 == 0 and 16-byte aligned
 * rows; other windows take the scalar path (correct, slower). */
int gm_elemwise_run_batch_2d(const gm_program* prog, int n_tiles, const void* const* inputs, const uint32_t* pitches,
                             void* const* outs, const uint32_t* out_pitches, size_t rows, size_t cols);
/* Device-to-device copy of a 2-D window into another pitch (compaction of a window), asynchronous: a kernel of this library
 * (16-byte moves, element-aligned on both sides), any alignment and width at the same speed. */
int gm_copy_2d(void* dst, size_t dst_pitch_bytes, const void* src, size_t src_pitch_bytes, size_t width_bytes, size_t rows);
/* ... of a window of several planes -- the frames of a series inside a tile of its raster (sources.py:119-149 warps every
 * frame): `planes` windows of `rows` rows; a plane of the source is src_plane_rows rows of src_pitch_bytes, one of the
 * destination dst_plane_rows rows of dst_pitch_bytes.  One copy. */
int gm_copy_3d(void* dst, size_t dst_pitch_bytes, size_t dst_plane_rows, const void* src, size_t src_pitch_bytes,
               size_t src_plane_rows, size_t width_bytes, size_t rows, size_t planes);
/* gm_elemwise_run_batch_2d over tiles of several bands that do not lie back to back (elemwise.py:243-297 computes on
 * (bands, y, x) arrays): input k of tile t is `bands` windows of rows x cols, band_pitches[t * prog->n_inputs + k] elements
 * apart (0: one operand for every band), the result of tile t likewise with out_band_pitches[t].  The bands run as tiles of
 * their own. */
int gm_elemwise_run_batch_bands(const gm_program* prog, int n_tiles, const void* const* inputs, const uint32_t* pitches,
                                const uint64_t* band_pitches, void* const* outs, const uint32_t* out_pitches,
                                const uint64_t* out_band_pitches, int bands, size_t rows, size_t cols);

/* Tuning knobs of the VM launch (for measurements; the defaults are the measured optimum). */
typedef struct gm_vm_tuning {
  int32_t vec;          /* elements per lane: 0 = auto (4; 8 for programs of more than 3 micro-ops), 4, 8 */
  int32_t max_batch;    /* tiles per launch of gm_elemwise_run_batch: 0 = 16 */
  int32_t reserved[2];  /* (tile loads/stores always carry the non-temporal hint: it is worth +10 % and
                           has to be a compile-time property of the kernel) */
} gm_vm_tuning;
int gm_vm_set_tuning(const gm_vm_tuning* t);
int gm_vm_get_tuning(gm_vm_tuning* t);
/* Program specialiser (csrc/gm_spec.hip).  The kernels behind gm_elemwise_run* interpret a program; a
 * program that turns out to be hot is compiled once, at run time (hiprtc, gfx950), into a kernel of its
 * own -- the same templates with the program as literals, hence the same results bit for bit -- and
 * cached per device and on disk ($GM_SPEC_CACHE, default ~/.cache/gmhip).  mode: 0 = off, 1 = programs
 * asked for at least hot_elements pixels (default; 0 keeps the current threshold, initially 2^25),
 * 2 = every program at its first launch (tests; a failing compilation is then an error).  Environment:
 * GM_SPEC, GM_SPEC_HOT, GM_SPEC_CACHE, GM_SPEC_VERBOSE, GM_SPEC_SYNC.  Without hiprtc everything keeps the interpreter.
 * In mode 1 the compilation (about a second) runs on a worker thread of the library: the launch that finds a
 * program hot -- and every launch until the kernel is loaded -- stays on the interpreter, no caller waits for the
 * compiler (GM_SPEC_SYNC=1: compile on the calling thread instead).  The disk cache is used only when its
 * directory is a private directory of this user (no link, mode 0700); its key covers the hiprtc / HIP versions. */
int gm_spec_set_mode(int mode, uint64_t hot_elements);
int gm_spec_get_mode(int* mode, uint64_t* hot_elements);
int gm_spec_stats(uint64_t* compiled, uint64_t* disk_hits, uint64_t* launches, uint64_t* failed);
int gm_spec_available(void);
/* Kernels being compiled in the background: their number / wait until there are none (a benchmark calls this after
 * its warm-up so that the timed region runs the kernels it is meant to time).  A host that may exit while a program has
 * just become hot calls gm_spec_wait() before it does: the compiler stack registers exit handlers of its own while it
 * works, and a process that runs them under the worker thread can crash or hang.  The library stops its worker from an
 * exit handler (no new compilation starts, no module is loaded once the exit has begun), which covers everything but an
 * exit in the middle of a compilation; the Python package calls gm_spec_wait() from the interpreter's exit hook. */
int gm_spec_pending(uint64_t* pending);
int gm_spec_wait(void);
/* Calls of gm_elemwise_run* so far and the longest time any of them kept its caller (milliseconds) -- what a caller
 * that had to wait for a compilation would show; reset != 0 clears both. */
int gm_elemwise_call_stats(uint64_t* calls, double* longest_ms, int reset);
/* Kernel-only timing of specialised launches: while enabled every such launch is bracketed by events stamped with
 * the kernel's own begin and end; gm_spec_timing_read waits for them, returns the sum of their durations in
 * microseconds and their number, and forgets them. */
int gm_spec_timing(int enable);
int gm_spec_timing_read(double* total_us, uint64_t* launches);
/* Name of the specialised kernel resolved last ("gm_spec_<hash of its program>", as profilers show it). */
int gm_spec_last_kernel(char* buf, size_t len);
/* The generated HIP source of a program's kernel / a compile check of it; neither needs a device.
 * group: 1 or 2 (4 or 8 elements per lane); strided: tiles are zero-copy windows. */
int gm_spec_source(const gm_program* prog, int strided, int group, char* buf, size_t len, size_t* needed);
int gm_spec_compile_check(const gm_program* prog, int strided, int group, size_t* code_bytes);

/* Drop every cached program (lowered code, tables and byte-LUTs go back to the pool).  Programs are cached by
 * content for the life of the process; a long-lived worker that has seen very many distinct programs -- or a
 * leak checker -- calls this while no launch is in flight. */
int gm_program_cache_clear(void);

/* sizeof of the ABI structs (0 gm_program, 1 gm_step, 2 gm_operand, 3 gm_input_desc): lets a
 * binding verify its struct layout at load time. */
size_t gm_sizeof(int which);

/* =============================================================================
 * K4: Classify -- np.digitize + no-data.  raster/misc.py:387-399
 *   out[i] = (in[i] == ndv && has_ndv) ? fill : digitize(in[i], bins, right)
 * Comparisons are made in float64 (NumPy promotes value and edges); NaN -> nbins.
 * out_dtype is GM_U8/GM_U16/GM_U32; bins is a HOST pointer (staged into LDS).
 * ========================================================================== */
int gm_classify(const void* in, int in_dtype, int has_ndv, uint64_t ndv_bits /* gm_scalar payload: .f64 for float tiles, .i64 for ints */,
                const double* bins_host, int nbins, int right,
                void* out, int out_dtype, uint64_t fill, size_t n);
/* Same with integer edges: when tile and edges are both integers NumPy compares in an integer
 * type; bins_are_int = 1 makes bins_host an array of int64 and the comparison exact int64. */
int gm_classify2(const void* in, int in_dtype, int has_ndv, uint64_t ndv_bits, const void* bins_host, int nbins,
                 int bins_are_int, int right, void* out, int out_dtype, uint64_t fill, size_t n);
/* The tile as a window of a wider resident raster: rows x cols cells, rows `pitch` cells apart; out is compact.  Served for
 * float32 tiles whose edges take the grid kernel (more than a handful of float edges; cols a multiple of 4 -- the window may
 * start on any cell and lie in rows of any pitch: a caller with cells to the right of an odd width asks for the next multiple);
 * GM_ERR_UNSUPPORTED = compact the window and call gm_classify2 (what the window path of the reference's Classify.process,
 * raster/misc.py:387-399, amounts to: it sees the array the source cut out). */
int gm_classify2_window(const void* in, int in_dtype, int has_ndv, uint64_t ndv_bits, const void* bins_host, int nbins,
                        int bins_are_int, int right, void* out, int out_dtype, uint64_t fill, int rows, int cols, int pitch);

/* =============================================================================
 * K5: Reclassify -- np.isin + np.searchsorted lookup.  raster/misc.py:482-515
 *   keys_host: ntab sorted unique int64 keys (the no-data key already merged in by
 *   the caller, misc.py:495-502); targets_host: ntab values of out_dtype's class
 *   (int64 or float64 bit patterns).  select: unmapped -> fill, else value cast.
 *   in_dtype: bool or any integer; out_dtype: GM_I64 or GM_F64.
 * ========================================================================== */
int gm_reclassify(const void* in, int in_dtype,
                  const int64_t* keys_host, const gm_scalar* targets_host, int ntab,
                  int select, uint64_t fill_bits, void* out, int out_dtype, size_t n);

/* =============================================================================
 * K6/K7: resampling of (bands, Hs, Ws) -> (bands, Hd, Wd)
 * ========================================================================== */

/* Nearest neighbour with separable destination->source index maps evaluated in float64.
 * Per axis (rows: row_params, columns: col_params), for destination index k:
 *   mode 0 (warp):   src = floor(((p[0] + (k + 0.5) * p[1]) - p[2]) / p[3] + eps)
 *                    p = {dst origin, dst step, src origin, src step} in world units -- the
 *                    GDAL nearest warp of an axis-aligned same-CRS request
 *                    (raster/sources.py:119-149; GDAL's kernel uses eps = 1e-10)
 *   mode 1 (affine): src = floor(k * p[0] + p[1] + 0.5)
 *                    SciPy affine_transform(order=0, matrix=diag(z), offset=o) as called by
 *                    Smooth's zoom branch (raster/spatial.py:296-305); p = {z, o, -, -}
 * src outside [0, n_src) -> fill.  finite_to_fill: non-finite source values -> fill
 * (sources.py:148).  row_params / col_params are HOST pointers to 4 doubles. */
int gm_resample_nearest(const void* src, int dtype, int bands, int hs, int ws,
                        void* dst, int hd, int wd, int mode,
                        const double* row_params, const double* col_params, double eps,
                        uint64_t fill_bits, int finite_to_fill);

/* Bilinear (our spec; no reference implementation exists): pixel-centre
 * aligned, edge-clamped, evaluated in float64, rounded once; any tap == ndv or
 * non-finite -> ndv.  dtype: GM_F32 or GM_F64.  Source tile + 1 px apron staged in LDS. */
int gm_resample_bilinear(const void* src, int dtype, int bands, int hs, int ws,
                         void* dst, int hd, int wd, int has_ndv, uint64_t ndv_bits);
/* The same on a zero-copy window of a wider resident raster: source rows src_pitch elements apart (>= ws), one band. */
int gm_resample_bilinear_window(const void* src, int dtype, int bands, int hs, int ws, int src_pitch, void* dst, int hd, int wd,
                                int has_ndv, uint64_t ndv_bits);

/* =============================================================================
 * N3: small-radius stencils (raster/spatial.py).  `in` is the (bands, h, w) tile INCLUDING the
 * margin the Block requested (expand_request_pixels, spatial.py:27-47); `out` is cropped.
 * ========================================================================== */

/* Dilate.process (spatial.py:143-154): out = (bands, h-2, w-2).  For every value, in order, a cell
 * takes it when the value occurs in its 6-neighbourhood over (band, y, x) of the ORIGINAL tile
 * (scipy.ndimage.binary_dilation's default structure).  values_bits: <= 16 gm_scalar payloads. */
int gm_dilate(const void* in, int dtype, int bands, int h, int w, const uint64_t* values_bits, int n_values, void* out);
/* The same on a zero-copy window of a wider resident raster: rows in_pitch elements apart (in_pitch >= w), one band.
 * Served by the LDS-tile kernels; a tile shape / dtype that would need another kernel returns GM_ERR_UNSUPPORTED and
 * the caller hands over a compact copy instead.  The plain entry point is this one with in_pitch = w. */
int gm_dilate_window(const void* in, int dtype, int bands, int h, int w, int in_pitch, const uint64_t* values_bits, int n_values,
                     void* out);
/* MovingMax.process (spatial.py:190-213): maximum over the circular footprint of diameter `size`
 * (odd, 3..63; utils.py:536-547) with no-data lowered to the dtype's minimum; a result that is
 * still the minimum on a no-data centre stays no-data.  out = (bands, h-2r, w-2r), r = size/2.
 * The (32 + 2r) x (64 + 2r) window of a workgroup is staged in LDS. */
int gm_moving_max(const void* in, int dtype, int bands, int h, int w, int size, int has_ndv, uint64_t ndv_bits, void* out);
int gm_moving_max_window(const void* in, int dtype, int bands, int h, int w, int in_pitch, int size, int has_ndv,
                         uint64_t ndv_bits, void* out); /* (window rules of gm_dilate_window) */
/* HillShade.process (spatial.py:354-417): out = uint8 (bands, h-2, w-2).  no-data -> fill (already
 * cast to the tile's dtype), gradients in the tile's dtype (integers wrap like NumPy), then
 * float32 exactly as NumPy orders it; arctan2 / sin are the float32 roundings of the float64
 * functions (NumPy's own float32 arctan2 / sin are not correctly rounded and differ between hosts), so
 * about 1-5 cells per million differ by 1 from a given NumPy build, at truncation boundaries. */
int gm_hillshade(const void* in, int dtype, int bands, int h, int w, int has_ndv, uint64_t ndv_bits, uint64_t fill_bits,
                 double xres, double yres, double altitude_deg, double azimuth_deg, void* out);
int gm_hillshade_window(const void* in, int dtype, int bands, int h, int w, int in_pitch, int has_ndv, uint64_t ndv_bits,
                        uint64_t fill_bits, double xres, double yres, double altitude_deg, double azimuth_deg,
                        void* out);
/* gm_dilate_window / gm_moving_max_window / gm_gauss_2d_window with the result written at a row pitch of its own
 * (out_pitch cells between the rows of `out`: the tile's window of a mosaic, RasterTiler.process parallelize.py:113-124
 * without the paste).  One band; GM_ERR_UNSUPPORTED for the dtypes / footprints whose kernels write compact tiles only
 * (then: the plain call + gm_paste_tile). */
int gm_dilate_into(const void* in, int dtype, int bands, int h, int w, int in_pitch, const uint64_t* values_bits, int n_values,
                   void* out, int out_pitch);
int gm_moving_max_into(const void* in, int dtype, int bands, int h, int w, int in_pitch, int size, int has_ndv,
                       uint64_t ndv_bits, void* out, int out_pitch);
int gm_gauss_2d_into(const void* in, int dtype, int bands, int h, int w, int in_pitch, const double* weights_y,
                     const double* weights_x, int radius, double cval, int replace_ndv, uint64_t ndv_bits, uint64_t fill_bits,
                     int y0, int x0, int oh, int ow, void* out, int out_pitch);

/* HillShade over the tiles of one mosaic in ONE launch (RasterTiler over HillShade: parallelize.py:82-89 fans the tiles out,
 * spatial.py:354-417 shades each): tile t is a single band of h x w cells at in[t] with row pitch in_pitch[t] (cells), its
 * (h - 2) x (w - 2) shades go to out[t] with row pitch out_pitch[t] (bytes = cells) -- e.g. straight into its window of the
 * mosaic.  float32 tiles, and integer tiles of up to 4 bytes at power-of-two resolutions; GM_ERR_UNSUPPORTED otherwise (run
 * gm_hillshade_window tile by tile). */
int gm_hillshade_batch(int n_tiles, const void* const* in, int dtype, int h, int w, const int* in_pitch, int has_ndv,
                       uint64_t ndv_bits, uint64_t fill_bits, double xres, double yres, double altitude_deg, double azimuth_deg,
                       void* const* out, const int* out_pitch); /* (window rules of gm_dilate_window) */

/* One axis (1 = y, 2 = x) of scipy.ndimage.gaussian_filter as Smooth.process calls it
 * (spatial.py:276-290; mode="constant"): weights_host[j] = weight at distance j (radius + 1 float64
 * values), accumulation in float64 in SciPy's order, result cast to the tile's dtype (float32 /
 * float64; integer tiles are truncated like SciPy's C cast).  replace_ndv: cells equal to ndv are read as fill (the first axis of Smooth does
 * values[values == ndv] = fill).  out = the window [y0, y0+oh) x [x0, x0+ow) of the filtered tile.  Any radius (a zoomed
 * Smooth request of N pixels can ask for up to 2 N / 3): the weights are kept in device memory, keyed by content. */
int gm_gauss_axis(const void* in, int dtype, int bands, int h, int w, int axis, const double* weights_host, int radius,
                  double cval, int replace_ndv, uint64_t ndv_bits, uint64_t fill_bits, int y0, int x0, int oh, int ow,
                  void* out);
/* ... of a single-band tile that is a window of a wider allocation: rows in_pitch cells apart (window rules of
 * gm_dilate_window). */
int gm_gauss_axis_window(const void* in, int dtype, int bands, int h, int w, int in_pitch, int axis,
                         const double* weights_host, int radius, double cval, int replace_ndv, uint64_t ndv_bits,
                         uint64_t fill_bits, int y0, int x0, int oh, int ow, void* out);

/* Both axes of the same filter in one launch for float32 tiles whose two axes have the same radius
 * (1..16): exactly gm_gauss_axis(axis 1) followed by gm_gauss_axis(axis 2) -- the first axis's result is
 * rounded to float32 before the second reads it, as SciPy stores it -- without the round trip of the
 * intermediate tile through HBM. */
int gm_gauss_2d(const void* in, int dtype, int bands, int h, int w, const double* weights_y, const double* weights_x,
                int radius, double cval, int replace_ndv, uint64_t ndv_bits, uint64_t fill_bits, int y0, int x0, int oh,
                int ow, void* out);
int gm_gauss_2d_window(const void* in, int dtype, int bands, int h, int w, int in_pitch, const double* weights_y,
                       const double* weights_x, int radius, double cval, int replace_ndv, uint64_t ndv_bits, uint64_t fill_bits,
                       int y0, int x0, int oh, int ow, void* out); /* (window rules of gm_dilate_window) */

/* =============================================================================
 * N4: statistics over the band (time) axis (raster/temporal.py).  `frames` is a C-contiguous
 * (n_frames, pixels) stack; a bin is a run of consecutive frames (the resampler bins of an
 * ascending time axis, temporal.py:717-722).
 * ========================================================================== */

enum gm_statistic {
  GM_STAT_SUM = 0,   /* np.nansum   */
  GM_STAT_COUNT = 1, /* frames that are data */
  GM_STAT_MIN = 2,   /* np.nanmin   */
  GM_STAT_MAX = 3,   /* np.nanmax   */
  GM_STAT_MEAN = 4,  /* np.nanmean  */
  GM_STAT_VAR = 5,   /* np.nanvar   */
  GM_STAT_STD = 6,   /* np.nanstd   */
  GM_STAT_MEDIAN = 7 /* np.nanmedian (gm_temporal_aggregate only; bins of any length: sorted in registers up to 32 frames,
                        in LDS up to GM_MAX_MEDIAN_FRAMES, selected in global memory beyond) */
};
#define GM_MAX_MEDIAN_FRAMES 64       /* per bin, sorted by 256-lane workgroups */
#define GM_MAX_MEDIAN_FRAMES_LONG 599 /* what the 64-lane sort reaches in the 160 KiB of LDS (640 float32 / 320 float64 values
                                         per lane): kept for GM_MEDIAN_LONG_SORT=1 -- since round 5 longer bins are selected,
                                         40 x faster; np.nanmedian itself takes another route from 600 frames on */

typedef struct gm_frame_bin {
  int32_t first; /* first source frame */
  int32_t count; /* frames in the bin; 0 = the label has no frame (result: fill) */
  int32_t out;   /* gm_temporal_aggregate: output frame written by this bin */
  int32_t reserved;
} gm_frame_bin;

/* dtype_for_statistic (utils.py:826-845) as a gm_dtype code; -1 for an unknown statistic. */
int gm_temporal_out_dtype(int dtype, int statistic);

/* TemporalAggregate.process, vals branch (temporal.py:724-768): out = (n_out, pixels) of
 * gm_temporal_out_dtype(dtype, statistic).  Cells equal to ndv (and NaN) are skipped; a result that
 * is not finite, or a bin without frames, becomes fill_bits (0 for sum / count, else the out
 * dtype's maximum).  Output frames no bin writes to are left untouched (gm_fill them first).
 * Accumulation runs frame by frame in result_type(float32, out dtype), like NumPy's axis-0
 * reductions; the mean divides in float64 and rounds once.  The median sorts the data frames of a
 * cell in LDS (one column per lane) and returns (low + high) / 2 as np.ma.median does.
 * has_ndv: bit 0 = cells equal to ndv are no data; for GM_STAT_MEDIAN and gm_temporal_percentile bit 1
 * (GM_TEMPORAL_KEEP_NONFINITE) leaves a result that is not finite as it is and fills only the cells without a
 * data frame -- what reduce_rasters does with its nan-functions (reduction.py:113-115), where TemporalAggregate
 * fills both (temporal.py:760-766). */
#define GM_TEMPORAL_KEEP_NONFINITE 2
int gm_temporal_aggregate(const void* frames, int dtype, int has_ndv, uint64_t ndv_bits, int64_t n_frames, int64_t pixels,
                          const gm_frame_bin* bins, int n_bins, int n_out, int statistic, void* out, uint64_t fill_bits);

/* np.nanpercentile (method "linear") of the data frames of a cell, for TemporalAggregate's
 * "p<number>" statistics.  plan[n - 1] tells what a cell with n data frames returns from its sorted
 * values v: a = v[lower], b = v[upper], d = b - a, result = gamma >= 0.5 ? b - d * one_minus_gamma
 * : a + d * gamma (numpy.lib._function_base_impl._lerp), all in result_type(float32, dtype).  The
 * host derives the plan with NumPy's own scalar arithmetic ((n - 1) * (q / 100) in that dtype), so
 * the rounding of the virtual index is NumPy's.  n_plan >= the longest bin. */
typedef struct gm_percentile_step {
  int32_t lower, upper;
  double gamma, one_minus_gamma;
} gm_percentile_step;
int gm_temporal_percentile(const void* frames, int dtype, int has_ndv, uint64_t ndv_bits, int64_t n_frames, int64_t pixels,
                           const gm_frame_bin* bins, int n_bins, int n_out, const gm_percentile_step* plan, int n_plan,
                           void* out, uint64_t fill_bits);

/* Snap.process / Resample.process (raster/temporal.py:165-182, 1196-1214): `data["values"][nearest]` -- output frame
 * j is source frame index[j] (0 <= index[j] < n_frames; index on the host, n_out <= 65535).  frames: (n_frames,
 * frame_bytes) C-contiguous on the device, out: (n_out, frame_bytes). */
int gm_frames_take(const void* frames, size_t frame_bytes, int64_t n_frames, const int32_t* index, int n_out, void* out);

/* TemporalSum.process (raster/temporal.py:239-257): `data["values"].sum(axis=0)` -- np.sum over the frame axis of a
 * (n_frames, pixels) stack, no-data values added like any other value, in np.sum's accumulator type
 * (gm_frames_sum_dtype: int64 for bool and signed integers, uint64 for unsigned ones, the type itself for floats),
 * which is the type of `out` (pixels).  Floats add frame by frame like NumPy's reduction over the outer axis; a
 * raster of one cell is NumPy's pairwise sum in pieces of 8192 frames. */
int gm_frames_sum_dtype(int dtype);
int gm_frames_sum(const void* frames, int dtype, int64_t n_frames, int64_t pixels, void* out);

/* reduce_rasters (raster/reduction.py:38-119), the part behind NumPy's nan-functions: `stack` is the
 * (n, pixels) C-contiguous stack the reference builds in its work dtype result_type(dtype, float16) with NaN
 * for no data (reduction.py:99-107) -- float32 or float64 storage; half = 1: the work dtype is float16
 * (1-byte rasters), stored as float32 and rounded to float16 after every operation like NumPy's float16
 * loops.  out = (pixels) in the same storage type: NumPy's func(stack, axis=0) per cell, `fill` (.f64
 * payload) where every raster is NaN.  MEDIAN / PERCENTILE: at most 64 rasters; PERCENTILE takes the plan of
 * gm_temporal_percentile (one step per number of data rasters, n steps). */
enum gm_reduce_statistic {
  GM_RED_SUM = 0, GM_RED_MIN = 2, GM_RED_MAX = 3, GM_RED_MEAN = 4, GM_RED_VAR = 5, GM_RED_STD = 6, GM_RED_MEDIAN = 7,
  GM_RED_PRODUCT = 8, GM_RED_ARGMIN = 9, GM_RED_ARGMAX = 10, GM_RED_PERCENTILE = 11
};
int gm_reduce_stack(const void* stack, int work_dtype, int half, int64_t n, int64_t pixels, int statistic,
                    const gm_percentile_step* plan, void* out, uint64_t fill_bits);
/* The same statistics over n SEPARATE rasters of the work dtype read in place (MEDIAN / PERCENTILE: up to 32 rasters, 64 of
 * float32 -- sorted in registers; `plan` as for gm_reduce_stack, NULL otherwise): the coding of the
 * stack -- `stack_array[i, index] = values[index]` with index = utils.get_index(values, no_data_value), reduction.py:99-107,
 * utils.py:61-64 -- happens in the load.  Per raster: its cells (device, `pixels` of them, C-contiguous), and the parameters of
 * ~np.isclose(values, ndv) as NumPy's promotion rules give them for that raster: the no-data value as compared, the tolerance
 * 1e-8 + 1e-5 |ndv|, flags bit 0: compare in float64, bit 1: the no-data value is finite.  float32 rasters with work dtype
 * float32, float64 with float64; anything else (other raster dtypes, a float16 work dtype) goes through gm_reduce_stack. */
#define GM_RED_COUNT 1  /* gm_reduce_rasters only: the number of rasters with data per cell (reduction.py:88-91), as the work dtype */
#define GM_RED_FIRST 12 /* gm_reduce_rasters only: the first / last raster's cell that is data (reduction.py:78-87); a NaN cell IS data */
#define GM_RED_LAST 13
typedef struct gm_raster_ref {
  const void* values;
  double ndv_compared, tolerance;
  uint32_t flags, reserved;
} gm_raster_ref;
int gm_reduce_rasters(const gm_raster_ref* rasters, int work_dtype, int64_t n, int64_t pixels, int statistic,
                      const gm_percentile_step* plan, void* out, uint64_t fill_bits);

/* ==========================================================================
 * N4: labelled (zonal) statistics -- the numeric core of geometry.AggregateRaster
 * (geometry/aggregate.py:160-203, the loop over frames inside aggregate_polygons).
 *
 *   values  : frames x cells raster values (device), dtype any gm_dtype
 *   labels  : cells int32 (device): the index of the geometry painted on each cell by
 *             utils.rasterize_geoseries; cells outside [0, n_labels) are unlabelled (the
 *             reference's rasterize no-data label is INT32_MAX)
 *   thresholds : n_labels float32 (device) or NULL: a cell only counts when its label's threshold is
 *             not NaN and value >= threshold (aggregate.py:181-184)
 *   statistic : gm_zonal_statistic_code; percentile in [0, 100] for GM_ZONAL_PERCENTILE
 *   out     : frames x n_labels float32 (device): what the reference assigns into `agg` -- scipy.ndimage
 *             sum / sum of ones / minimum / maximum / mean / median resp. measurements.percentile
 *             (measurements.py:24-134) of the label's active cells, cast to float32; NaN for a label
 *             without an active cell
 *   counts  : frames x n_labels int64 (device): active cells per label (the caller assigns only labels
 *             with counts > 0, like `agg[frame_no][select_and_active] = ...`)
 *   present : n_labels bytes (device) or NULL: 1 where the label occurs in `labels` at all
 *             (set(select) - unique_labels are the reference's geometries_no_cells)
 * sum and mean are the float32 roundings of np.bincount's sum: the cells of a label added in raster order in float64.
 * A label whose float64 sum in ANY order lies provably clear of every float32 rounding boundary is answered without that
 * order (the bound is derived in csrc/gm_zonal.hip: zonal_chunk_combine_kernel); every other label is added in order.
 * The float32 bits are bincount's either way.  GM_ZONAL_LAZY=0 in the environment: every sum in order.
 * ========================================================================== */
enum gm_zonal_statistic_code {
  GM_ZONAL_SUM = 0, GM_ZONAL_COUNT = 1, GM_ZONAL_MIN = 2, GM_ZONAL_MAX = 3, GM_ZONAL_MEAN = 4, GM_ZONAL_MEDIAN = 5,
  GM_ZONAL_PERCENTILE = 6
};
int gm_zonal_statistic(const void* values, int dtype, int64_t frames, int64_t cells, int has_ndv, uint64_t ndv_bits,
                       const int32_t* labels, int32_t n_labels, const float* thresholds, int statistic, double percentile,
                       float* out, int64_t* counts, uint8_t* present);

/* The same with the three results in HOST memory (what aggregate_polygons needs them for: `agg[...] = ...` is NumPy on the host):
 * out_host frames x n_labels float32, counts_host frames x n_labels int64, present_host n_labels bytes or NULL.  One wait of the
 * host per call where gm_zonal_statistic + a copy are two: the copy is queued behind the kernels of a route before that route's
 * verdict is waited for, and trusted only when the verdict says the route answered. */
int gm_zonal_statistic_host(const void* values, int dtype, int64_t frames, int64_t cells, int has_ndv, uint64_t ndv_bits,
                            const int32_t* labels, int32_t n_labels, const float* thresholds, int statistic, double percentile,
                            float* out_host, int64_t* counts_host, uint8_t* present_host);


/* Cumulative.process, vals branch (temporal.py:957-1003): running sum / count inside every bin;
 * source frame j goes to output frame j - emit_first for emit_first <= j < emit_stop.
 * statistic: GM_STAT_SUM or GM_STAT_COUNT. */
int gm_temporal_cumulative(const void* frames, int dtype, int has_ndv, uint64_t ndv_bits, int64_t n_frames, int64_t pixels,
                           const gm_frame_bin* bins, int n_bins, int emit_first, int emit_stop, int statistic, void* out,
                           uint64_t fill_bits);

/* =============================================================================
 * K8: helpers
 * ========================================================================== */

/* *result = 1 iff every element equals `value` (Clip early-out, misc.py:108). Synchronous. */
int gm_all_equal(const void* in, int dtype, uint64_t value_bits, size_t n, int* result);
/* *result = 1 iff every element is finite (always 1 for integer tiles). Synchronous.  Lets a
 * source skip the non-finite -> no-data rewrite of raster/sources.py:148 when nothing needs it. */
int gm_all_finite(const void* in, int dtype, size_t n, int* result);
/* out[i] = value */
int gm_fill(void* out, int dtype, uint64_t value_bits, size_t n);
/* Synthetic tile: the window [row0, row0+h) x [col0, col0+w) of a procedural H x W mosaic.
 * Pixel (r, c) of the mosaic = lo + u * (hi - lo) cast to dtype, u = splitmix64-uniform in [0, 1)
 * keyed by (seed, r * W + c); `ndv` where r % lattice_y == 0 && c % lattice_x == 0
 * (lattice_x == 0 disables) and outside the mosaic.  Tiles of any size therefore agree. */
int gm_generate(void* out, int dtype, uint64_t seed, int64_t row0, int64_t col0, int h, int w,
                int64_t H, int64_t W, double lo, double hi, uint32_t lattice_y, uint32_t lattice_x,
                uint64_t ndv_bits);
/* 64-bit order-independent checksum (sum of mixed 32-bit words) of a device buffer. Synchronous. */
int gm_checksum(const void* in, size_t bytes, uint64_t* result);
/* Paste a (bands, th, tw) tile into a (bands, H, W) mosaic at (row0, col0): the device side
 * of RasterTiler.process (raster/parallelize.py:118-124). itemsize in bytes. */
int gm_paste_tile(void* mosaic, int H, int W, const void* tile, int bands, int th, int tw,
                  int row0, int col0, int itemsize);
/* The same for all tiles of a mosaic in one launch (RasterTiler.process, parallelize.py:113-124): tile t of th[t] x tw[t]
 * cells per band goes to rows row0[t] .., columns col0[t] .. of every band of the mosaic (host arrays of n_tiles). */
int gm_paste_tiles(void* mosaic, int H, int W, int n_tiles, const void* const* tiles, const int* th, const int* tw,
                   const int* row0, const int* col0, int bands, int itemsize);

#ifdef __cplusplus
}
#endif
#endif /* GMHIP_H */
