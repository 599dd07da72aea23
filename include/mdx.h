/*
 * mdx.h — C-ABI of libmdx.so, the MI355X-native trajectory-analysis core.
 *
 * The reference (bbye98/mdhelper) has no FFI on this path: its hot loops sit
 * behind Python functions.  Each entry point below therefore cites the Python
 * function (reference file:line) whose work it carries; INTEGRATION.md shows
 * the ctypes stub a maintainer adds at that call site.
 *
 * Conventions
 *   - every function returns MDX_OK (0) or a negative error class;
 *     mdx_last_error() returns the thread-local message of the last failure;
 *   - plain pointers and sizes only; the caller owns every host buffer and the
 *     library never keeps a host pointer past return;
 *   - "_device" variants take pointers obtained from mdx_malloc() (HBM-resident
 *     input, no PCIe traffic inside the call);
 *   - a handle owns its device memory and stream, is bound to one device and
 *     is not thread-safe (one handle per device per host thread);
 *   - positions are float32[n_frames][n][3] (MDAnalysis' native layout),
 *     boxes float32[n_frames][6] = (lx, ly, lz, alpha, beta, gamma), or NULL
 *     for no periodic boundaries.  Orthorhombic and triclinic cells are both
 *     handled by the RDF (a batch may mix them); the Fourier-space engines take
 *     positions only.
 */
#ifndef MDX_H
#define MDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDX_VERSION 100 /* 0.1.0 */

#define MDX_OK 0
#define MDX_ERR_INVALID_VALUE (-1) /* -> ValueError          */
#define MDX_ERR_UNSUPPORTED (-2)   /* -> NotImplementedError */
#define MDX_ERR_NO_DEVICE (-3)     /* -> RuntimeError        */
#define MDX_ERR_HIP (-4)           /* -> RuntimeError        */
#define MDX_ERR_ROCFFT (-5)        /* -> RuntimeError        */
#define MDX_ERR_RCCL (-6)          /* -> RuntimeError        */
#define MDX_ERR_OUT_OF_MEMORY (-7) /* -> MemoryError         */
#define MDX_ERR_STATE (-8)         /* -> RuntimeError        */
#define MDX_ERR_IO (-9)            /* -> OSError             */

/* ------------------------------------------------------------------ runtime
 * Loading the library sets GPU_PINNED_MIN_XFER_SIZE=1048576 (MiB) in the process unless the variable is set already:
 * the HIP runtime then stages pageable copies through its own buffers instead of page-locking caller memory and
 * KEEPING the registration (a kept registration of memory that is freed and handed out again, or of a file mapping
 * that is truncated, aborts or blocks the process: NOTES.md round 5).  Copies of >= 1 MiB through mdx_memcpy_* /
 * mdx_upload* use the library's pinned ring or pages it locks and unlocks itself.  MDX_ABORT_TRACE=<fd>: native
 * call stack on SIGABRT. */

const char *mdx_last_error(void);
int mdx_version(void);
int mdx_device_count(int *count);
/* name: caller buffer of name_len bytes; any out pointer may be NULL. */
int mdx_device_info(int dev, char *name, size_t name_len, int *compute_units,
                    size_t *hbm_bytes, size_t *hbm_free_bytes);
int mdx_malloc(int dev, size_t bytes, void **dptr);
int mdx_free(int dev, void *dptr);
int mdx_memcpy_h2d(int dev, void *dst, const void *src, size_t bytes);
int mdx_memcpy_d2h(int dev, void *dst, const void *src, size_t bytes);
int mdx_memset(int dev, void *dst, int value, size_t bytes);
/* Host memory -> HBM at the rate of the host-buffer entry points: page-locked / registered memory by one DMA
 * where it lies, pageable memory through the library's pinned ring with its copy threads (so does mdx_memcpy_h2d
 * from 1 MiB on; below that it is the runtime's staging).  Returns when the data is in HBM. */
int mdx_upload(int dev, void *d_dst, const void *src, size_t bytes);
/* ... for n_rows rows of row_bytes that lie src_stride bytes apart on the host and end up contiguous in HBM: a
 * range of particles out of every frame of float32[T][N][3] (row_bytes = 12 n_range, src_stride = 12 N) — what
 * Onsager streams per group while the group before is being transformed (transport.py:976-992).  Pageable rows of
 * >= 4 KB that lie >= 2 pages apart in anonymous memory are read by the DMA engine where they lie: slices of ~128 MB
 * are page-locked, copied by one 2-D DMA and unlocked again, several side by side; every slice is unlocked before
 * the call returns.  Short or nearly contiguous rows and rows of a file mapping are gathered into the pinned ring. */
int mdx_upload_rows(int dev, void *d_dst, const void *src, size_t row_bytes, size_t src_stride, size_t n_rows);
/* Destroyed handles and mdx_free leave their device blocks (of any size) in a per-device, per-process cache so
 * that an analysis object per call does not pay hipMalloc / hipFree each time (at C4 size: seconds inside
 * hipMalloc every other analysis).  The cache holds at most MDX_CACHE_GB GiB when that variable is set, else a
 * quarter of the memory that was FREE on the device when the cache was first used (at least 4 GiB).  It is given
 * back automatically when an allocation of this library fails, and here on request: call mdx_trim_cache before
 * handing the device to another library or process (rocFFT, another rank sharing the GPU) — nobody else can
 * reclaim it.  mdx_device_info reports free memory WITHOUT the cached bytes (they are reported separately by
 * mdx_cached_bytes).  freed_bytes may be NULL. */
int mdx_trim_cache(int dev, size_t *freed_bytes);
int mdx_cached_bytes(int dev, size_t *bytes);
/* Which user-mode ROCm libraries this library's calls are bound to in THIS process — the files the dynamic
 * linker resolved hipGetDeviceCount, rocfft_setup and ncclGetVersion to — with their versions and the ROCm root
 * the library was built for, as "key=value\n" text (keys: rocm_root, libamdhip64, librocfft, librccl, libmdx,
 * hip_runtime_version, hip_driver_version, rccl_version, rocfft_version).  Works without a device.  A process
 * that loaded another copy of these libraries first (e.g. `import torch`, whose wheel bundles its own ROCm)
 * binds libmdx.so to THAT copy; mdhelper_amd._lib refuses to run so (one process, one runtime). */
int mdx_runtime_info(char *buf, size_t bytes);
int mdx_device_synchronize(int dev);
/* Page-lock a caller buffer (e.g. the float32[n_frames][n][3] array an MDAnalysis memory reader
 * holds — what `universe.trajectory[frame].positions` views, reference structure.py:753,796) so
 * that the host-buffer entry points below hand it to the DMA engine where it lies, without the
 * staging copy through the library's own pinned ring.  Optional: unregistered memory works, one
 * host copy slower.  The caller unregisters before freeing the buffer. */
/* Whole pages are locked (the range is widened to page boundaries); a range that shares a page with one still
 * registered is refused (MDX_ERR_STATE); mdx_host_unregister takes the pointer that was registered, waits for
 * the device, and fails for a pointer it does not know.  A buffer only partly covered by a registration is
 * treated as pageable by every entry point. */
int mdx_host_register(int dev, void *ptr, size_t bytes);
int mdx_host_unregister(int dev, void *ptr);

/* Synthetic wrapped Gaussian random walk, generated in HBM (bench / tests):
 * frame 0 uniform in [0, L)^3, then x += sigma * N(0,1) wrapped into the box;
 * counter-based Philox-4x32-10 keyed by (seed, atom, frame) so any frame can
 * be regenerated anywhere.  out: float32[n_frames][n_atoms][3] on the device. */
int mdx_synth_random_walk(int dev, float *d_out, int64_t n_frames, int64_t n_atoms,
                          const float box_lengths[3], float sigma, uint64_t seed,
                          int wrap);
/* Unwrapped walk in float64[n_frames][n_atoms][3] (the Onsager position store,
 * transport.py:932), same generator. */
int mdx_synth_random_walk_f64(int dev, double *d_out, int64_t n_frames, int64_t n_atoms,
                              const float box_lengths[3], float sigma, uint64_t seed);

/* ---------------------------------------------------------------- collectives
 * One process per GPU; the communicator is RCCL over xGMI.  Rank 0 calls
 * mdx_comm_unique_id() and ships the 128-byte id to the other ranks through
 * whatever rendezvous the launcher offers (mdhelper_amd/launch.py: a node-local
 * unix socket); then every rank calls mdx_comm_init_rank(). */
typedef struct mdx_comm *mdx_comm_t;
#define MDX_COMM_ID_BYTES 128
int mdx_comm_unique_id(unsigned char id[MDX_COMM_ID_BYTES]);
int mdx_comm_init_rank(mdx_comm_t *comm, int dev, const unsigned char id[MDX_COMM_ID_BYTES],
                       int rank, int world_size);
int mdx_comm_destroy(mdx_comm_t comm);
/* what RCCL itself reports for this communicator (ncclCommCount / UserRank / CuDevice);
 * any out pointer may be NULL */
int mdx_comm_count(mdx_comm_t comm, int *count, int *rank, int *device);
int mdx_comm_barrier(mdx_comm_t comm);
/* in-place sum / max all-reduce of small host vectors (staged through HBM) */
int mdx_comm_allreduce_f64(mdx_comm_t comm, double *host_inout, int64_t n, int op_max);
int mdx_comm_allreduce_i64(mdx_comm_t comm, int64_t *host_inout, int64_t n);

/* ------------------------------------------------------------------------ RDF
 * Replaces the per-frame body of RadialDistributionFunction._single_frame
 * (reference src/mdhelper/analysis/structure.py:750-791), i.e. the call
 *     counts += radial_histogram(pos1, pos2, n_bins, range, dims, exclusion)
 * (structure.py:32-104, :788-791) for a batch of frames at once. */
typedef struct mdx_rdf *mdx_rdf_t;

#define MDX_RDF_ALGO_AUTO 0
#define MDX_RDF_ALGO_EXACT_F64 1 /* contract arithmetic on every pair            */
#define MDX_RDF_ALGO_FILTER_F32 2 /* f32 filter + exact f64 re-evaluation near edges */
#define MDX_RDF_ALGO_CELL 3      /* cell-sorted tiles, culled tile pairs, f32 filter */

/* edges: the n_bins+1 float64 bin edges exactly as numpy.linspace(r_min, r_max,
 * n_bins+1) yields them (structure.py:737; numpy.histogram builds the same
 * array).  excl1/excl2: the reference's `exclusion` tuple, 0/0 for None. */
int mdx_rdf_create(mdx_rdf_t *out, int dev, int n_bins, const double *edges,
                   int64_t excl1, int64_t excl2, int algo);
int mdx_rdf_destroy(mdx_rdf_t h);
int mdx_rdf_reset(mdx_rdf_t h);
/* Host buffers.  pos2 == NULL (or == pos1 with n2 == n1) means ag2 is ag1: the
 * kernel then evaluates each unordered pair once and counts it twice, which is
 * bit-identical because the contract arithmetic is exactly antisymmetric. */
int mdx_rdf_accumulate(mdx_rdf_t h, const float *pos1, int64_t n1, const float *pos2,
                       int64_t n2, const float *boxes, int64_t n_frames);
/* 2-D mode, RadialDistributionFunction(drop_axis=...) (structure.py:761-770): coordinate `axis`
 * (0, 1, 2; -1 switches the mode off) of both sets is set to zero after the centre-of-mass stage
 * and the cell length along it becomes max(lx, ly, lz), on the device, for every entry point. */
int mdx_rdf_set_drop_axis(mdx_rdf_t h, int axis);
/* Centres of mass on the device for groupings="residues"/"segments" (structure.py:753-759 via
 * algorithm/molecule.py:300-306): the rows of set `which` (1 or 2) handed to any accumulate
 * call are then PARTICLES, molecule g owning rows [offsets[g], offsets[g+1]); the histogram is
 * taken over the float32 centres sum_a m_a x_a / M_g.  offsets: int64[n_groups+1], masses:
 * float64[offsets[n_groups]], both on the host; n_groups <= 0 removes the grouping. */
int mdx_rdf_set_grouping(mdx_rdf_t h, int which, int64_t n_groups, const int64_t *offsets,
                         const double *masses);
/* Same, all pointers in HBM (from mdx_malloc); asynchronous on the handle's stream. */
int mdx_rdf_accumulate_device(mdx_rdf_t h, const float *d_pos1, int64_t n1,
                              const float *d_pos2, int64_t n2, const float *d_boxes,
                              int64_t n_frames);
/* Waits for the stream, then copies int64[n_bins] counts (np.intp, structure.py:740). */
int mdx_rdf_counts(mdx_rdf_t h, int64_t *counts);
int mdx_rdf_synchronize(mdx_rdf_t h);
/* Sum the counts of every rank's handle (one RCCL all-reduce, uint64 sum). */
int mdx_rdf_allreduce(mdx_rdf_t h, mdx_comm_t comm);
/* Timing / statistics of the pair kernel, measured with HIP events on the
 * handle's stream: launches since reset, total milliseconds, ordered pairs
 * covered (frames * n1 * n2), pairs re-evaluated by the exact path, and distance
 * evaluations actually executed (after symmetry and tile culling, padding
 * included).  Any pointer may be NULL. */
int mdx_rdf_stats(mdx_rdf_t h, int64_t *launches, double *kernel_ms,
                  int64_t *pairs_evaluated, int64_t *pairs_exact, int64_t *pairs_computed);
int mdx_rdf_enable_timing(mdx_rdf_t h, int on);
/* Raw device counters since reset: [0] exact re-evaluations, [1] (64 x 16)-pair units run by
 * the cell kernel, [2] units on its per-pair image-search path, [3] brute-force evaluations. */
int mdx_rdf_debug_counters(mdx_rdf_t h, int64_t out[4]);
/* Timing runs (mdx_rdf_enable_timing): the engine clock the cell-sorted pair kernel actually ran
 * at since reset, in Hz — every block adds its span in s_memtime and in 100 MHz s_memrealtime
 * ticks; 0 when no such kernel has run. */
int mdx_rdf_kernel_clock(mdx_rdf_t h, double *hz);
/* Cell path: the sorted copies (wrapped and original float4 rows, n_pad rows) of one frame of
 * the most recent slab — for debugging the tile logic on the host.  With exclusion 0 or 1 the
 * sorted originals are not materialised: pw is filled and MDX_ERR_STATE returned. */
int mdx_rdf_debug_sorted(mdx_rdf_t h, int64_t frame, int64_t n_pad, float *pw, float *po);

/* Function-level drop-in for structure.radial_histogram (structure.py:32-104):
 * one frame, host buffers, counts overwritten. */
int mdx_radial_histogram(int dev, const float *pos1, int64_t n1, const float *pos2,
                         int64_t n2, int n_bins, const double *edges, const float dims[6],
                         int64_t excl1, int64_t excl2, int64_t *counts);

/* --------------------------------------------------------------- structure factor
 * Replaces StructureFactor._single_frame (structure.py:1481-1527) with its
 * Numba kernels delta_fourier_transform_sum_2d_2d / inner_2d_2d /
 * pythagorean_trigonometric_identity_* (src/mdhelper/algorithm/accelerated.py:81-321):
 *   rho_g(q) = sum_{j in group g} exp(i q.r_j)  (fp64), then per pair (j,k)
 *   ssf += |rho_j|^2  (j == k)   or   2 Re(rho_j rho_k*)  (j != k).
 * Both `form`s of the reference map onto this one fused kernel. */
typedef struct mdx_sq *mdx_sq_t;

/* wavevectors: float64[n_q][3].  group_offsets: int64[n_groups+1] into the
 * concatenated position array (structure.py:1427-1431).  pairs: int32[n_pairs][2]
 * group indices; (-1,-1) = all particles as one group (mode=None). */
int mdx_sq_create(mdx_sq_t *out, int dev, const double *wavevectors, int64_t n_q,
                  const int64_t *group_offsets, int n_groups, const int32_t *pairs,
                  int n_pairs);
int mdx_sq_destroy(mdx_sq_t h);
int mdx_sq_reset(mdx_sq_t h);
/* groupings="residues" / "segments" (structure.py:1484-1486 with center_of_mass): the rows handed to
 * every accumulate entry point are then particles sorted molecule by molecule — molecule g = rows
 * [offsets[g], offsets[g+1]), masses float64[offsets[n_molecules]] — and the Fourier sums run over
 * the float32 centres of mass formed on the device; group_offsets of mdx_sq_create index the
 * molecules.  Groups with groupings="atoms" enter as molecules of one particle and mass 1.
 * n_molecules <= 0 removes the grouping. */
int mdx_sq_set_grouping(mdx_sq_t h, int64_t n_molecules, const int64_t *offsets, const double *masses);
int mdx_sq_accumulate(mdx_sq_t h, const float *pos, int64_t n, int64_t n_frames);
/* Positions already in HBM.  ASYNCHRONOUS on the handle's stream: the call may return while kernels still read
 * d_pos; a producer that rewrites the buffer (an MD engine feeding batches) calls mdx_sq_synchronize first.
 * mdx_sq_result and mdx_sq_destroy wait by themselves. */
int mdx_sq_accumulate_device(mdx_sq_t h, const float *d_pos, int64_t n, int64_t n_frames);
int mdx_sq_synchronize(mdx_sq_t h);
/* float64[n_pairs][n_q] un-normalised sums over frames (structure.py:1494-1508). */
int mdx_sq_result(mdx_sq_t h, double *ssf);
int mdx_sq_allreduce(mdx_sq_t h, mdx_comm_t comm);
int mdx_sq_stats(mdx_sq_t h, int64_t *launches, double *kernel_ms);
int mdx_sq_enable_timing(mdx_sq_t h, int on);
/* Function-level drop-in for accelerated.delta_fourier_transform_sum_2d_2d
 * (accelerated.py:81-122): out = complex128[n_q] as (re, im) pairs; float64 positions. */
int mdx_fourier_sum(int dev, const double *wavevectors, int64_t n_q, const double *positions,
                    int64_t n, double *out_re_im);
/* Function-level drop-ins for the trigonometric forms (accelerated.py:167-247, :249-321, :323-627; the public
 * static methods StructureFactor.ssf_trigonometric_2d / psf_trigonometric_2d_2d, structure.py:1238-1317):
 *   mdx_inner         out[i][j] = q_i . r_j, float64[n_q][n]                      (inner_2d_2d)
 *   mdx_trig_rowsums  cos_out[i] = sum_j cos(x[i][j]), sin_out[i] = sum_j sin(x[i][j]) for a caller-supplied
 *                     float64[n_rows][n_cols] (either output may be NULL)         (cosine_sum_*, sine_sum_*,
 *                     and, squared and added on the host, pythagorean_trigonometric_identity_*)
 * x streams from host memory in row slabs (pinned ring / DMA) beside the kernels; the _device variant takes
 * x and the outputs in HBM and returns when they are written. */
int mdx_inner(int dev, const double *wavevectors, int64_t n_q, const double *positions, int64_t n, double *out);
int mdx_trig_rowsums(int dev, const double *x, int64_t n_rows, int64_t n_cols, double *cos_out, double *sin_out);
int mdx_trig_rowsums_device(int dev, const double *d_x, int64_t n_rows, int64_t n_cols, double *d_cos,
                            double *d_sin);

/* ------------------------------------------------- intermediate scattering function
 * Replaces IntermediateScatteringFunction._single_frame (structure.py:1956-2083): the
 * coherent part from lagged products of rho_g(q, f) and, optionally, the incoherent
 * part sum_j cos(q . (r_j(f) - r_j(f - lag))) for lags 0 .. n_lags-1.  Arguments as for
 * mdx_sq_create; frames must be fed in analysis order (consecutive calls continue). */
typedef struct mdx_isf *mdx_isf_t;
int mdx_isf_create(mdx_isf_t *out, int dev, const double *wavevectors, int64_t n_q,
                   const int64_t *group_offsets, int n_groups, const int32_t *pairs, int n_pairs,
                   int n_lags, int incoherent);
int mdx_isf_destroy(mdx_isf_t h);
int mdx_isf_reset(mdx_isf_t h);
int mdx_isf_accumulate(mdx_isf_t h, const float *pos, int64_t n, int64_t n_frames);
/* Same, positions already in HBM on the engine's device (float32[n_frames][n][3]).  ASYNCHRONOUS on the
 * handle's stream: the frames are copied into the engine's position ring by a device-to-device copy queued on
 * that stream, and the call may return before the copy has run; a producer that rewrites d_pos calls
 * mdx_isf_synchronize first (mdx_isf_result / mdx_isf_stats / mdx_isf_destroy wait by themselves). */
int mdx_isf_accumulate_device(mdx_isf_t h, const float *d_pos, int64_t n, int64_t n_frames);
int mdx_isf_synchronize(mdx_isf_t h);
/* As mdx_sq_set_grouping; only before the first frame of a series. */
int mdx_isf_set_grouping(mdx_isf_t h, int64_t n_molecules, const int64_t *offsets, const double *masses);
/* cisf: float64[n_lags][n_pairs][n_q]; iisf (may be NULL): float64[n_lags][n_slots][n_q] with
 * n_slots = 1 for mode=None (pairs = (-1,-1)) and n_groups otherwise; un-normalised sums. */
int mdx_isf_result(mdx_isf_t h, double *cisf, double *iisf);
int mdx_isf_stats(mdx_isf_t h, int64_t *launches, double *kernel_ms, int64_t *frames);
int mdx_isf_enable_timing(mdx_isf_t h, int on);

/* ------------------------------------------------------------- time correlation
 * Replaces algorithm.correlation.correlation_fft / msd_fft
 * (src/mdhelper/algorithm/correlation.py:17-226, :461-668) as called from
 * Onsager._conclude (src/mdhelper/analysis/transport.py:1016-1059). */
typedef struct mdx_msd *mdx_msd_t;

/* One engine per (n_frames_block, n_blocks).  The transform length n_fft >= 2 n_frames_block - 1
 * (any such length gives the same linear correlation; the reference pads to
 * 2*next_fast_len(n_frames_block), correlation.py:176-178): the shortest length of the engine's own
 * two-pass kernels (400 x 2^k or a power of two) for blocks of 201 .. 524 288 frames, else the reference
 * length or the next power of two through rocFFT (mdx_msd_n_fft / mdx_msd_transform report it). */
int mdx_msd_create(mdx_msd_t *out, int dev, int64_t n_frames_block, int n_blocks,
                   int n_groups);
int mdx_msd_destroy(mdx_msd_t h);
int mdx_msd_reset(mdx_msd_t h);
int mdx_msd_n_fft(mdx_msd_t h, int64_t *n_fft);
/* Which forward transform the engine runs: *own = 1 and n_fft = r1 x r2 for the engine's own two-pass kernels
 * (csrc/mdx_msd_fft.hpp), *own = 0 (r1 = r2 = 0) for the rocFFT pipeline.  No reference counterpart: the
 * reference has one path (scipy.fft, correlation.py:176-197); benchmarks and tests name the path they measured. */
int mdx_msd_transform(mdx_msd_t h, int *own, int *r1, int *r2);
/* Feed particles of one group: pos float64[n_blocks*n_frames_block][n_total][3]
 * (transport.py:932 layout), of which particles [first, first+count) belong to
 * `group`.  Accumulates sum_particles of the per-particle self MSD numerators
 * and sum_particles r(t) for the collective terms.  zero_dims: bit k set ->
 * dimension k is zeroed (transport.py:1025,1033). */
int mdx_msd_push(mdx_msd_t h, int group, const double *pos, int64_t n_total, int64_t first,
                 int64_t count, int zero_dims);
int mdx_msd_push_device(mdx_msd_t h, int group, const double *d_pos, int64_t n_total,
                        int64_t first, int64_t count, int zero_dims);
/* The same for float32 positions resident in HBM — what a trajectory reader or a GPU MD engine holds —, a plain
 * particle range with nothing to prepare (no unwrapping, no molecule centres, no shift): pass A widens them as it
 * stages them, so no float64 copy is made (C4: 18 GB of traffic per group less).  Only the engine's transforms
 * with a 400- or 64-point first factor read float32 in place (mdx_msd_transform: r1 == 400 or 64 — the single-pass
 * kernel, the 400 x R2 family and 2^13 .. 2^16: every block length up to 204 800 frames); other engines (2^18 .. 2^20,
 * rocFFT) return MDX_ERR_UNSUPPORTED
 * and the caller goes through mdx_msd_push_frames_device.  Results equal those of the widened
 * frames bit for bit. */
int mdx_msd_push_device_f32(mdx_msd_t h, int group, const float *d_pos, int64_t n_total, int64_t first,
                            int64_t count, int zero_dims);
/* msd_self[g][b][t] = sum_particles MSD_particle / N_g  (transport.py:1036-1039, before /2D)
 * sum_traj[g][b][t][3] = sum_particles r(t)            (input of :1034 and :1044-1052) */
int mdx_msd_result(mdx_msd_t h, double *msd_self_sum, double *sum_traj);
/* acf_sum[g][b][m] = sum_particles sum_d sum_k x(k) x(k+m), m < n_frames_block, not normalised:
 * the vector ACF numerator of correlation_fft(..., average=True, vector=True) as called by
 * EndToEndVector._conclude (src/mdhelper/analysis/polymer.py:765-781) = acf_sum / (N (T - m)). */
int mdx_msd_result_acf(mdx_msd_t h, double *acf_sum);
int mdx_msd_allreduce(mdx_msd_t h, mdx_comm_t comm);
int mdx_msd_stats(mdx_msd_t h, int64_t *launches, double *kernel_ms, int64_t *bytes_moved);
int mdx_msd_enable_timing(mdx_msd_t h, int on);

/* Function-level drop-in for correlation.correlation_fft on real input
 * (correlation.py:17-226): n_series independent series of length n_t, laid out
 * [n_series][n_t] contiguous; b == NULL -> ACF.  out[n_series][n_t] =
 * sum_k a[k] b[k+m] (un-normalised, lags m = 0..n_t-1); with neg_out != NULL the
 * negative lags sum_k a[k+m] b[k] are written there too. */
int mdx_correlate(int dev, const double *a, const double *b, int64_t n_series, int64_t n_t,
                  double *out, double *neg_out);

/* ------------------------------------------------------- trajectory ingest */

/* Native readers for the files the reference's users analyse: AMBER NetCDF trajectories
 * (the container and variables its own writer produces,
 * src/mdhelper/openmm/file.py:49-52 `NETCDF3_64BIT_OFFSET`, :160-188 `coordinates`,
 * `cell_lengths`, `cell_angles`, `time`) and CHARMM/NAMD DCD.  They replace the per-frame
 * Python reader behind `universe.trajectory[frame]` (structure.py:796, transport.py:976-985)
 * on the way to the GPU: raw records go through pinned buffers to HBM and are byte-swapped /
 * transposed / gathered there.  format: 1 NetCDF, 2 DCD. */
typedef struct mdx_traj *mdx_traj_t;
int mdx_traj_open(mdx_traj_t *out, const char *path);
int mdx_traj_close(mdx_traj_t h);
int mdx_traj_info(mdx_traj_t h, int64_t *n_frames, int64_t *n_atoms, int *has_box, int *has_time,
                  int *format);
/* Host reads of the listed frames: float32[n][n_atoms][3] (ts.positions), float32[n][6]
 * (ts.dimensions: lx ly lz alpha beta gamma), float64[n] times in ps. */
int mdx_traj_read_positions(mdx_traj_t h, const int64_t *frames, int64_t n, float *out);
int mdx_traj_read_boxes(mdx_traj_t h, const int64_t *frames, int64_t n, float *boxes6);
int mdx_traj_read_times(mdx_traj_t h, const int64_t *frames, int64_t n, double *times);
/* The listed frames into HBM: d_out float32[n][n_sel][3].  d_index: device int32[n_sel]
 * particle indices, or NULL for the first n_sel particles (n_sel <= 0: all). */
int mdx_traj_load_device(mdx_traj_t h, int dev, const int64_t *frames, int64_t n,
                         const int32_t *d_index, int64_t n_sel, float *d_out);
/* RDF over frames of a trajectory file.  boxes: host float32[n_frames][6] of those frames
 * (mdx_traj_read_boxes) or NULL; index1/index2: host int32 particle selections (ag.indices),
 * NULL = all particles; index2 == NULL with n2 == 0 = the same group twice. */
int mdx_rdf_accumulate_traj(mdx_rdf_t h, mdx_traj_t traj, const int64_t *frames,
                            int64_t n_frames, const float *boxes, const int32_t *index1,
                            int64_t n1, const int32_t *index2, int64_t n2);
/* S(q) / ISF over frames of a trajectory file.  index: host int32[n_index] particle indices
 * in the order of the concatenated groups (the `self._positions[s] = g.positions` fill of
 * structure.py:1484-1486), or NULL for the file's first n_index particles (<= 0: all).
 * The ISF consumes the frames in the order listed. */
int mdx_sq_accumulate_traj(mdx_sq_t h, mdx_traj_t traj, const int64_t *frames, int64_t n_frames,
                           const int32_t *index, int64_t n_index);
int mdx_isf_accumulate_traj(mdx_isf_t h, mdx_traj_t traj, const int64_t *frames,
                            int64_t n_frames, const int32_t *index, int64_t n_index);

/* MSD / Onsager: the first n_blocks * n_frames_block listed frames of a trajectory file as
 * one group's positions (the `self._positions[frame, slice] = g.positions` fill of
 * transport.py:976-992 for groupings="atoms").  index: host int32[n_index] particle indices or
 * NULL.  unwrap != 0 applies the reference's global unwrap (algorithm/topology.py:366-376,
 * thresholds dims/2 as passed by transport.py:933-937) on the device, with dims = the three
 * box lengths; zero_dims as for mdx_msd_push. */
int mdx_msd_push_traj(mdx_msd_t h, int group, mdx_traj_t traj, const int64_t *frames,
                      int64_t n_frames, const int32_t *index, int64_t n_index, int unwrap,
                      const double *dims, int zero_dims, const double *shift);
/* groupings="residues" / "segments" for the trajectory-file path (transport.py:983-992 with
 * center_of_mass(g, gr, images=...)): the rows of the following mdx_msd_push_traj calls are particles
 * sorted molecule by molecule — molecule g = rows [offsets[g], offsets[g+1]), masses
 * float64[offsets[n_molecules]] — and the engine receives the float64 centres of mass of the
 * unwrapped particles, minus `shift`.  n_molecules <= 0 removes the grouping. */
int mdx_msd_set_grouping(mdx_msd_t h, int64_t n_molecules, const int64_t *offsets, const double *masses);
/* System centre of mass of every listed frame, out float64[n_frames][3] (host), over the
 * listed particles with the given masses (transport.py:993-1014: all atoms for center_atom,
 * else the groups' particles); unwrap as above; wrap != 0 brings coordinates outside [0, L]
 * back first (center_wrap).  Its output is the `shift` of mdx_msd_push_traj (NULL: none):
 * shift[t] is subtracted from every position of frame t. */
int mdx_msd_system_com_traj(mdx_msd_t h, mdx_traj_t traj, const int64_t *frames, int64_t n_frames,
                            const int32_t *index, int64_t n_index, const double *masses, int unwrap,
                            const double *dims, int wrap, double *out);
/* Molecules made whole in the first analysed frame (Onsager(unwrap=True), reference
 * transport.py:936-941: make_whole over universe.atoms.fragments before the starting positions are
 * stored): images int32[n_sel][3] = the periodic image each row starts in, i.e. the flags the
 * reference's first unwrap call derives from x - x_whole.  They apply to every following unwrapped
 * push / system-COM call, whose selection must have n_sel rows in this order; n_sel = 0 clears. */
int mdx_msd_set_initial_images(mdx_msd_t h, const int32_t *images, int64_t n_sel);
/* The same frame preparation for frames in host memory (an in-memory trajectory): pos
 * float32[n_frames][n_sel][3] holds the selection, already gathered, in analysis order (rows sorted
 * molecule by molecule when mdx_msd_set_grouping is active); arguments otherwise as for
 * mdx_msd_push_traj / mdx_msd_system_com_traj. */
int mdx_msd_push_f32(mdx_msd_t h, int group, const float *pos, int64_t n_frames, int64_t n_sel,
                     int unwrap, const double *dims, int zero_dims, const double *shift);
int mdx_msd_system_com_f32(mdx_msd_t h, const float *pos, int64_t n_frames, int64_t n_sel,
                           const double *masses, int unwrap, const double *dims, int wrap, double *out);
/* ... and for float64 frames (an in-memory float64 trajectory: the reference's positions are the
 * float64 copies of transport.py:978); same stages, the unwrap state kept in float64. */
int mdx_msd_push_f64(mdx_msd_t h, int group, const double *pos, int64_t n_frames, int64_t n_sel,
                     int unwrap, const double *dims, int zero_dims, const double *shift);
int mdx_msd_system_com_f64(mdx_msd_t h, const double *pos, int64_t n_frames, int64_t n_sel,
                           const double *masses, int unwrap, const double *dims, int wrap, double *out);
/* Cross displacements between the groups' summed trajectories, out float64[n_pairs][n_blocks][n_frames_block]:
 * msd_fft(sum_i r, sum_j r) of reference correlation.py:461-668 as Onsager._conclude calls it for every pair
 * (transport.py:1034, 1052; pairs int32[n_pairs][2], i == j gives the collective MSD of a group), computed
 * from the summed trajectories the pushes have left in HBM (after mdx_msd_allreduce: of all ranks). */
int mdx_msd_cross(mdx_msd_t h, const int32_t *pairs, int64_t n_pairs, double *out);
/* ... and for frames already resident in HBM (a GPU MD engine's output, or a host / file trajectory uploaded
 * once for several groups: mdx_upload, mdx_traj_load_device): d_pos float32 (elem_bytes 4) or float64 (8)
 * [n_frames][n_total][3]; index: host int32[n_index] rows of the selection in analysis order, or NULL for
 * the first n_index rows (<= 0: all).  Nothing is staged or copied: the frame-preparation kernels gather. */
int mdx_msd_push_frames_device(mdx_msd_t h, int group, const void *d_pos, int elem_bytes, int64_t n_frames,
                               int64_t n_total, const int32_t *index, int64_t n_index, int unwrap,
                               const double *dims, int zero_dims, const double *shift);
int mdx_msd_system_com_device(mdx_msd_t h, const void *d_pos, int elem_bytes, int64_t n_frames,
                              int64_t n_total, const int32_t *index, int64_t n_index, const double *masses,
                              int unwrap, const double *dims, int wrap, double *out);
/* With a grouping declared (mdx_msd_set_grouping) both mdx_msd_system_com_* variants return the
 * centre of mass of the molecules' CENTRES (wrapped into the box first when wrap != 0), each
 * weighted with its molecule's mass: Onsager(center=True, center_atom=False, center_wrap=True)
 * with residue / segment groupings (transport.py:1004-1014). */

#ifdef __cplusplus
}
#endif
#endif /* MDX_H */
