/* include/edt_hip.h -- C ABI of the MI355X-native multi-label anisotropic EDT.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  The entry
 * points below are exactly what the reference's FFI for this path binds
 * (`cdef extern from "edt.hpp" namespace "pyedt"`, /root/reference/src/edt.pyx:62-113),
 * with the template parameter T replaced by a run-time `dtype` code.  INTEGRATION.md shows
 * the reference-side stubs (Cython / C++ header) a maintainer would add.
 *
 * Conventions (identical to the reference, src/edt.hpp:411-484):
 *   - x is the fastest axis: idx = x + sx * (y + sy * z);
 *   - label 0 is background; any change of label is a boundary;
 *   - wx, wy, wz are the physical voxel sizes (anisotropy) as fp32; every size of an axis the call uses
 *     must be finite and non-zero, wx positive, or the call fails with EDT_ERR_BAD_ARG before any device
 *     work.  A negative wy / wz is taken as |w|: along y and z a size enters only as its square
 *     (src/edt.hpp:181, :258), which is what the reference computes with it.  (Stated deviation for the
 *     rest: the reference does not validate -- a negative wx makes its pass 1 cross label boundaries,
 *     src/edt.hpp:107-109, NaN / inf / 0 give NaN or all-zero fields);
 *   - black_border != 0 treats the outside of the volume as background;
 *   - output is fp32, squared distances unless the entry point says otherwise;
 *     with black_border == 0 voxels that see no boundary are +INF.
 *   - `parallel` is accepted for signature compatibility and ignored (the GPU grid
 *     replaces the reference's ThreadPool, src/threadpool.h:46-140).
 *
 * Every function returns EDT_OK (0) or a negative error code and never throws;
 * edt_hip_last_error() describes the last failure on the calling thread.  There is NO
 * CPU fallback inside this library: without a usable gfx950 device every compute entry
 * point fails with EDT_ERR_NO_DEVICE.
 */
#ifndef EDT_HIP_H
#define EDT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* label element types (the seven instantiations of src/edt.pyx:670-732) */
enum {
  EDT_U8 = 0,
  EDT_U16 = 1,
  EDT_U32 = 2,
  EDT_U64 = 3,
  EDT_F32 = 4,
  EDT_F64 = 5,
  EDT_BOOL = 6 /* one byte per voxel, any non-zero byte is foreground */
};

enum {
  EDT_OK = 0,
  EDT_ERR_NO_DEVICE = -1,
  EDT_ERR_BAD_ARG = -2,
  EDT_ERR_HIP = -3,
  EDT_ERR_UNSUPPORTED = -4,
  EDT_ERR_NOMEM = -5
};

/* flags for the device-resident entry points */
enum {
  EDT_FLAG_BLACK_BORDER = 1, /* treat the outside of the volume as background          */
  EDT_FLAG_SQRT = 2,         /* return distances instead of squared distances          */
  EDT_FLAG_FORCE_GENERIC = 4, /* use the size-agnostic fallback kernels (test hook)     */
  EDT_FLAG_BATCH_2D = 8,     /* edt_hip_edtsq_device with ndim = 3: the volume is a STACK of sz independent
                                2-D images (sx x sy each) -- x and y passes only, one launch for all images */
  EDT_FLAG_BINARY_YZ = 32,   /* the reference's *binary* route for a multi-valued label type (pyedt::_binary_edt{2,3}dsq<T>,
                                src/edt.hpp:487-576, :681-755): labels split runs in pass X only; passes Y and Z treat every
                                column as one envelope from its first non-zero value on, background voxels as height-0
                                sites.  Identical to the ordinary transform on 0/1 input. */
  EDT_FLAG_SIGNED = 64,      /* edt_hip_edtsq_device: the SIGNED transform -- sdf / sdfsq of the reference's Python layer
                                (src/edt.pyx:121-202: edt(x) - edt(x == 0)) as ONE transform: label 0 is measured like every
                                other label and its voxels come out negated.  Bit-identical to the two-transform definition
                                (the two fields have disjoint supports, and a voxel's value depends only on the voxels of its
                                own label runs).  Shapes: edt_hip_signed_supported; others: EDT_ERR_UNSUPPORTED. */
  EDT_FLAG_SMALL_WORKSPACE = 16 /* scratch = the four bit planes only (1/2 byte per voxel): passes X and Y then
                                exchange fp32 values instead of 16-bit distance indices (no 256 MiB index slab,
                                about 7 % slower at 512^3); pass it to edt_hip_workspace_bytes_flags as well */
};

/* ---- introspection -------------------------------------------------------------- */
int edt_hip_device_count(void);         /* number of visible HIP devices (0 if none)  */
/* 1 if every multiple k * wx, k <= sx + 1, is exactly representable in fp32 -- the reference's sequential fp32 sums
 * of the voxel size (src/edt.hpp:97, :113) then ARE the multiples, and pass X may hand pass Y 16-bit distance indices
 * instead of fp32 values (needs no device; exported so that tests can hold the criterion against its property). */
int edt_hip_index_form_exact(float wx, int64_t sx);
/* 1 if edt_hip_edtsq_device serves EDT_FLAG_SIGNED for this shape and these flags (needs no device) */
int edt_hip_signed_supported(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, int flags);
const char *edt_hip_last_error(void);   /* thread-local, never NULL                   */
const char *edt_hip_version(void);

/* ---- host-buffer entry points (numpy in / numpy out) ------------------------------
 * Pointers are HOST pointers; the call stages through device memory (H2D, kernels, D2H)
 * and is synchronous.  `output` must hold one float per voxel and is fully overwritten.
 * `output` must NOT overlap `labels`: for results of 32 MiB and more, helper threads touch (write zeros into) the
 * pages of `output` while the labels are still travelling to the device, so that the copy back runs at PCIe speed.
 * For the same reason the contents of `output` are unspecified after a call that FAILED.
 * (EDT_HIP_NO_PREFAULT=1 in the environment switches the helper threads off.)
 */

/* replaces pyedt::squared_edt_1d_multi_seg<T>  (src/edt.hpp:70-119; bound at
 * src/edt.pyx:63-70).  stride must be 1 (every reference caller passes 1). */
int edt_hip_squared_edt_1d_multi_seg(const void *labels, int dtype, float *dest, int64_t n,
                                     int64_t stride, float anisotropy, int black_border);

/* replaces pyedt::_edt2dsq<T>  (src/edt.hpp:632-678, bool: :758-772; bound at
 * src/edt.pyx:72-78) */
int edt_hip_edt2dsq(const void *labels, int dtype, int64_t sx, int64_t sy, float wx, float wy,
                    int black_border, int parallel, float *output);

/* replaces pyedt::_edt3dsq<T>  (src/edt.hpp:411-484, bool: :580-587; bound at
 * src/edt.pyx:80-86) -- THE hot path. */
int edt_hip_edt3dsq(const void *labels, int dtype, int64_t sx, int64_t sy, int64_t sz, float wx,
                    float wy, float wz, int black_border, int parallel, float *output);

/* replace pyedt::_edt2d<T> / _edt3d<T>  (src/edt.hpp:776-797, :591-604): the same plus a
 * correctly rounded sqrt fused into the last pass. */
int edt_hip_edt2d(const void *labels, int dtype, int64_t sx, int64_t sy, float wx, float wy,
                  int black_border, int parallel, float *output);
int edt_hip_edt3d(const void *labels, int dtype, int64_t sx, int64_t sy, int64_t sz, float wx,
                  float wy, float wz, int black_border, int parallel, float *output);

/* replace pyedt::_edt2dsq_voxel_graph<T,uint8_t> / _edt3dsq_voxel_graph<T,uint8_t>
 * (src/edt_voxel_graph.hpp:54-117, :120-214; bound at src/edt.pyx:89-100).  `graph` holds
 * one byte per voxel; only bits 0x01 (+x), 0x04 (+y), 0x10 (+z) are read. */
int edt_hip_edt2dsq_voxel_graph(const void *labels, int dtype, const uint8_t *graph, int64_t sx,
                                int64_t sy, float wx, float wy, int black_border,
                                float *workspace);
int edt_hip_edt3dsq_voxel_graph(const void *labels, int dtype, const uint8_t *graph, int64_t sx,
                                int64_t sy, int64_t sz, float wx, float wy, float wz,
                                int black_border, float *workspace);

/* replaces pyedt::_binary_edt2dsq<T> / _binary_edt2d<T> / _binary_edt3dsq<T> / _binary_edt3d<T> (src/edt.hpp:681-755,
 * :487-576, :607-629 -- what edt::binary_edt / edt::binary_edtsq instantiate for ANY label type): pass X splits runs
 * at label changes like the multi-label transform, passes Y and Z do not (one envelope per column, zeros as
 * height-0 sites).  ndim in {2,3} (sz = 1 for 2-D); take_sqrt != 0: distances instead of squared distances. */
int edt_hip_binary_edtsq(const void *labels, int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, float wx,
                         float wy, float wz, int black_border, int take_sqrt, float *output);

/* ---- one process, several GPUs ---------------------------------------------------------------------
 * pyedt::_edt3dsq / _edt3d on host buffers, Z-sharded over the listed devices of this process (a host thread per
 * device; X and Y passes per Z-slab in z-chunks, ONE exchange of slab records as peer-to-peer copies over xGMI --
 * the copies of a chunk run under the kernels of the next --, Z pass per Y-slab, each device copies its rows of the
 * result straight into `output`).  Same result, bit for bit, as the single-device entry points.  Device buffers and
 * streams are kept between calls (edt_hip_release_cache frees them).  The same ordinal may be listed more than once
 * (virtual devices: tests on one GPU).  Distinct ordinals must have peer access to each other: a pair without it is
 * EDT_ERR_UNSUPPORTED naming the pair (EDT_HIP_ALLOW_STAGED_PEER=1 in the environment accepts staging through host
 * memory).  A volume the slab-record form cannot cut n_devices ways (edt_hip_multi_supported == 0: sx, sy or
 * sz > 2048, fewer z-slices or 32-row words of y than devices) is EDT_ERR_UNSUPPORTED too; n_devices = 1 runs on
 * that device.
 * edt_hip_set_devices makes the ordinary host-buffer 3-D entry points (edt_hip_edt3dsq / edt_hip_edt3d, and with
 * them edt::edt<T>() and the Python / Cython front ends) take this route; every OTHER host-buffer call -- volumes that
 * cannot be cut that way (one note on stderr), 1-D and 2-D transforms, stacks of images, the binary route, the forced
 * generic kernels, sdf and the voxel-graph transforms -- runs on the FIRST listed device (the caller's current device is
 * restored afterwards); the *_device entry points are never redirected (their buffers live where the caller put them);
 * n_devices = 0 restores the current-device behaviour.  The environment variable EDT_HIP_DEVICES="0,1,2,..." presets the list. */
int edt_hip_edt3dsq_multi(const void *labels, int dtype, int64_t sx, int64_t sy, int64_t sz, float wx, float wy,
                          float wz, int black_border, int take_sqrt, float *output, const int *devices,
                          int n_devices);
int edt_hip_multi_supported(int dtype, int64_t sx, int64_t sy, int64_t sz, int n_devices);
int edt_hip_set_devices(const int *devices, int n_devices);

/* sdf / sdfsq of the reference's Python layer (src/edt.pyx:121-158, :161-202): edt(labels) - edt(labels == 0)
 * (squared != 0: edtsq(labels) - edtsq(labels == 0)) on host buffers in one round trip -- labels up once, both
 * transforms and the subtraction on the device, the difference down once.  ndim in {1,2,3}, unused extents 1. */
int edt_hip_sdf(const void *labels, int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, float wx, float wy,
                float wz, int black_border, int squared, float *output);

/* The host-buffer entry points keep their device buffers (labels, output, scratch) between calls --
 * allocating gigabytes per call costs more than moving them over PCIe.  This frees them.
 * (EDT_HIP_NO_CACHE=1 in the environment disables the cache altogether.) */
int edt_hip_release_cache(void);

/* ---- device-resident entry points --------------------------------------------------
 * All pointers are DEVICE pointers on the current HIP device; `stream` is a hipStream_t
 * (NULL = the null stream).  Calls only enqueue work (no host synchronisation, no
 * allocation), so they can be timed with events and captured into a hipGraph.
 */

/* bytes of scratch edt_hip_edtsq_device needs for a volume of this shape: the four bit planes of the column
 * passes (1/8 of a byte per voxel each: 0.5 GiB for 1024^3) plus one slab of 16-bit distance indices between
 * passes X and Y (2 bytes per voxel, never more than 256 MiB: larger volumes run those passes slab by slab).
 * Only a call that has to use the size-agnostic column kernels (an axis longer than 32735 voxels, or
 * EDT_FLAG_FORCE_GENERIC) needs a second fp32 volume and the hull stacks as well (+ 8 bytes per voxel): ask
 * with the flags of the call. */
size_t edt_hip_workspace_bytes(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz);
size_t edt_hip_workspace_bytes_flags(int dtype, int ndim, int64_t sx, int64_t sy, int64_t sz, int flags);

/* A stack of `count` independent 2-D images, images[k][y][x] (pyedt::_edt2dsq per image, src/edt.hpp:632-678):
 * one launch per pass for the whole stack, so that small images fill the chip (a single 512 x 512 image is 16
 * workgroups on 256 compute units).  Host buffers; take_sqrt != 0 gives edt instead of edtsq.  The device-resident
 * form is edt_hip_edtsq_device(ndim = 3, sz = count, flags | EDT_FLAG_BATCH_2D). */
int edt_hip_edt2dsq_batch(const void *labels, int dtype, int64_t sx, int64_t sy, int64_t count, float wx, float wy,
                          int black_border, int take_sqrt, float *output);

/* ndim in {1,2,3}; unused extents must be 1.  flags: EDT_FLAG_*.  d_output may not alias
 * d_labels.  Implements _edt3dsq / _edt2dsq / squared_edt_1d_multi_seg (+ optional sqrt).
 * ndim = 1: with edt_hip_workspace_bytes() of scratch the line runs through the parallel pipeline (a thread per
 * voxel); with d_workspace = NULL (or too small) it is served by one thread walking the line -- correct, and slow
 * for long lines. */
int edt_hip_edtsq_device(const void *d_labels, int dtype, int ndim, int64_t sx, int64_t sy,
                         int64_t sz, float wx, float wy, float wz, int flags, float *d_output,
                         void *d_workspace, size_t workspace_bytes, void *stream);

/* Per-pass timing hook for bench.py: when enabled, edt_hip_edtsq_device brackets every
 * kernel it launches with hipEvents on `stream`.  After synchronising the stream,
 * edt_hip_get_pass_times copies the durations (ms) of the last call, in launch order,
 * into `ms` and returns how many there were (names via edt_hip_get_pass_name). */
int edt_hip_set_profiling(int enabled);
/* Diagnostics / test hook.  The mode belongs to the CALLING THREAD (EDT_HIP_DEBUG_MODE in the environment presets
 * every thread's).  The shipped library honours only bits that choose between result-preserving forms of the same
 * computation -- e.g. 0x2000: no tile of the column pass takes the windowed path, 0x4000: every tile does, 0x8000:
 * fp64 candidates there (0x2000000: only where fp32 fma candidates would serve), 0x100000: fp32 instead of 16-bit indices between passes X and Y, 0x20000: up-sampled
 * voxel-graph formulation, 32 / 64: workgroup-phased row / column kernels (csrc/edt_common.h lists them); results
 * are bit-identical under every one of them, which is what the GPU test tier uses them for.  Bits that switch
 * phases off (wrong results, for cost measurements) exist only in a build with -DEDT_DIAG and are ignored here.
 * edt_hip_get_debug_mode returns the effective (masked) mode of the calling thread. */
int edt_hip_set_debug_mode(int mode);
int edt_hip_get_debug_mode(void);
/* Diagnostics (pure host arithmetic, no device needed): the library's own proof that the 16-bit integer column kernel
 * (csrc/edt_colq16.hip) can refuse no tile of pass Y / pass Z of a call of these extents and voxel sizes whose pass X hands
 * over 16-bit distance indices -- where it holds, the fp32 launch over the hand-over list is not made.  Returns 0 if the
 * voxel sizes share no quantum (the integer kernel does not apply), 1 otherwise with *pass_y / *pass_z set to 0 / 1 (pass_z:
 * behind a pass Y that holds; always 0 for ndim == 2).  The CPU test tier plays both passes through the kernel's lane logic
 * and holds this answer against what the tiles do (tests/test_q16_logic.py).  No counterpart in the reference. */
int edt_hip_q16_no_refusals(int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz, int ndim, int black_border,
                            int *pass_y, int *pass_z);
int edt_hip_get_pass_times(float *ms, int capacity);
const char *edt_hip_get_pass_name(int index);

/* ---- Z-sharded (multi-GPU) building blocks ------------------------------------------
 * One process per GPU holds a contiguous Z-slab.  Phase 1 runs the X and Y passes on the
 * slab (no data dependence across z) and emits, per voxel, the fp32 partial result plus
 * one flag byte (bit0 = foreground, bit1 = label differs from the voxel below in z).
 * The host layer then re-partitions both arrays from Z-slabs to Y-slabs with ONE
 * all-to-all (RCCL over xGMI) and phase 2 runs the Z pass on whole z-columns.
 *   d_halo: the last xy-slice of the previous rank's labels (NULL on the first rank) --
 *           the one-slab halo that decides run continuity across the cut.
 */
size_t edt_hip_shard_workspace_bytes(int dtype, int64_t sx, int64_t sy, int64_t sz);
int edt_hip_shard_xy_device(const void *d_labels, const void *d_halo, int dtype, int64_t sx,
                            int64_t sy, int64_t sz_local, float wx, float wy, int flags,
                            float *d_partial, uint8_t *d_zflags, void *d_workspace,
                            size_t workspace_bytes, void *stream);
int edt_hip_shard_z_device(float *d_partial, const uint8_t *d_zflags, int64_t sx,
                           int64_t sy_local, int64_t sz, float wz, int flags,
                           void *d_workspace, size_t workspace_bytes, void *stream);
/* The same with `field_floor`: a lower bound of the NON-ZERO values of the partial field, which the caller of both
 * phases knows -- after edt_hip_shard_xy_* every non-zero value is at least min(fl32(wx*wx), fl32(wy*wy))
 * (edt_hip_field_floor).  It lets the Z pass form its candidates in fp32 for voxel sizes whose multiples are not exact
 * in fp32; results are the same bits either way.  0 (or anything not positive) = unknown, which is
 * what the entry points without the argument pass.  A floor ABOVE a non-zero value of the field is a contract
 * violation (results may differ from the reference's in the last bit). */
float edt_hip_field_floor(float wx, float wy);
int edt_hip_shard_z_device_ex(float *d_partial, const uint8_t *d_zflags, int64_t sx,
                              int64_t sy_local, int64_t sz, float wz, float field_floor, int flags,
                              void *d_workspace, size_t workspace_bytes, void *stream);

/* Slab records: the fast form of the same two phases (sx, sy and sz <= 2048; query with
 * edt_hip_shard_records_supported, otherwise use the pair above).  The y axis is cut into `nparts`
 * destination ranges at multiples of 32 rows (y_splits[0] = 0 ... y_splits[nparts] = sy, HOST array).
 * For destination h and every xy-slice of the slab the XY phase writes ONE contiguous record of
 * edt_hip_shard_record_floats(sx, ylen_h) 4-byte elements straight into d_blocks[h] (device pointer
 * to sz_local consecutive records; HOST array of nparts pointers):
 *     ylen_h * sx floats   the slice after the X and Y passes, rows of that destination only
 *     words_h * sx uint32  foreground bits, 32 rows of y per word      (words_h = ceil(ylen_h / 32))
 *     words_h * sx uint32  "label differs from the voxel below in z" bits, same packing
 * i.e. 4.25 bytes per voxel travel instead of the labels, each peer's message is contiguous on
 * both sides (no pack / unpack copies), and a rank may hand in its OWN part of the receive buffer
 * as d_blocks[rank].  A slab can be processed in z-chunks (d_halo of a later chunk = the last
 * slice of the previous one) so that the exchange of one chunk overlaps the kernels of the next.
 * The Z phase takes the gathered buffer of sz records (all z, this rank's rows) and leaves the result
 * in the float part of every record: row (z, y) starts at d_records + z * record_floats + y * sx. */
int edt_hip_shard_records_supported(int dtype, int64_t sx, int64_t sy, int64_t sz);
size_t edt_hip_shard_record_floats(int64_t sx, int64_t y_rows);
size_t edt_hip_shard_records_workspace_bytes(int dtype, int64_t sx, int64_t sy, int64_t sz);
int edt_hip_shard_xy_records_device(const void *d_labels, const void *d_halo, int dtype, int64_t sx,
                                    int64_t sy, int64_t sz_local, float wx, float wy, int flags,
                                    int nparts, const int64_t *y_splits, void *const *d_blocks,
                                    void *d_workspace, size_t workspace_bytes, void *stream);
int edt_hip_shard_z_records_device(float *d_records, int64_t sx, int64_t sy_local, int64_t sz,
                                   float wz, int flags, void *d_workspace, size_t workspace_bytes,
                                   void *stream);
int edt_hip_shard_z_records_device_ex(float *d_records, int64_t sx, int64_t sy_local, int64_t sz,
                                      float wz, float field_floor, int flags, void *d_workspace,
                                      size_t workspace_bytes, void *stream);
/* The same for a caller that names all three voxel sizes (the ones of the XY phase that produced the records and this
 * phase's): field_floor = edt_hip_field_floor(wx, wy), and where the three sizes share a quantum ((1,1,1), (6,6,30) ...)
 * the pass runs on the 16-bit integer column kernel (csrc/edt_colq16.hip); the same bits either way. */
int edt_hip_shard_z_records_device_w(float *d_records, int64_t sx, int64_t sy_local, int64_t sz, float wx, float wy,
                                     float wz, int flags, void *d_workspace, size_t workspace_bytes, void *stream);

/* Slab records of 16-BIT values (2.25 bytes per voxel over the links instead of 4.25).  Where the three voxel sizes share a
 * quantum (csrc/edt_colq16.hip: w_i^2 = a_i q) and both column axes fit the integer kernel (97..1024 rows), the Y pass's
 * results are integers N < 2^16 in quanta: the record of destination h and slice z is then
 *     ylen_h * sx 16-bit values (row-major, as packed pairs)  |  the two bit planes as above
 * = edt_hip_shard_record16_words(sx, ylen_h) 4-byte words (an even number: blocks are 8-byte aligned, the gathered buffer and
 * d_out 16-byte).  edt_hip_shard_records16_supported: the extents of the WHOLE volume
 * and its voxel sizes allow it.  A tile the integer kernel cannot take (values beyond 16 bits, rows without any boundary)
 * has no 16-bit form: the XY phase adds the number of such tiles to *d_refused -- a device counter the caller zeroes and
 * reads -- and leaves their rows unspecified; a caller that finds it non-zero repeats the step with the fp32 records
 * (edt/distributed.py: the ranks agree on it with one all-reduce of the counter, off the critical path).  The Z phase reads
 * the gathered records (sz x record16 words) and writes the dense (sz, sy_local, sx) fp32 result to d_out; the same bits as
 * every other route.  No counterpart in the reference (src/edt.hpp:448-475 is what both phases replace). */
int edt_hip_shard_records16_supported(int dtype, int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz);
size_t edt_hip_shard_record16_words(int64_t sx, int64_t y_rows);
int edt_hip_shard_xy_records16_device(const void *d_labels, const void *d_halo, int dtype, int64_t sx, int64_t sy,
                                      int64_t sz_local, float wx, float wy, float wz, int flags, int nparts,
                                      const int64_t *y_splits, void *const *d_blocks, uint32_t *d_refused, void *d_workspace,
                                      size_t workspace_bytes, void *stream);
int edt_hip_shard_z_records16_device(const void *d_records, float *d_out, int64_t sx, int64_t sy_local, int64_t sz, float wx,
                                     float wy, float wz, int flags, void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- fused helpers on device-resident data ------------------------------------------ */
/* out[i] = a[i] - b[i]  (src/edt.pyx:156-158, sdf = edt(x) - edt(x == 0)) */
int edt_hip_subtract_device(const float *d_a, const float *d_b, float *d_out, int64_t count,
                            void *stream);
/* pyedt::extract_runs (src/edt_voxel_graph.hpp:238-268) on device-resident labels: the START offsets of the
 * maximal constant runs of the flattened array, ascending (run k = [starts[k], starts[k+1]) resp. up to count for
 * the last one; its label is labels[starts[k]]).  *d_count receives the number of runs; at most `capacity` starts are
 * written (capacity = 0, d_starts = NULL: count only).  Enqueue-only; scratch: edt_hip_runs_workspace_bytes. */
size_t edt_hip_runs_workspace_bytes(int64_t count);
int edt_hip_extract_runs_device(const void *d_labels, int dtype, int64_t count, int64_t *d_starts, int64_t capacity,
                                int64_t *d_count, void *d_workspace, size_t workspace_bytes, void *stream);

/* mask[i] = (labels[i] == 0) as one byte per voxel (the `data == 0` of src/edt.pyx:157) */
int edt_hip_is_background_device(const void *d_labels, int dtype, uint8_t *d_mask, int64_t count,
                                 void *stream);

/* pyedt::_edt2dsq_voxel_graph / _edt3dsq_voxel_graph (src/edt_voxel_graph.hpp:54-117, :120-214) on
 * device-resident data: labels, graph (one byte per voxel, bits as in the host entry points above) and
 * output live in HBM; enqueue-only on `stream`.  Scratch: edt_hip_voxel_graph_workspace_bytes (native form: the
 * even-x cells of the doubled grid as fp32, 4 x voxels floats, and the bit planes of its two column passes; the
 * up-sampled fallback for doubled axes beyond 2048 rows: the 2x uint8 volume, its fp32 transform and the
 * ordinary workspace of that volume). */
size_t edt_hip_voxel_graph_workspace_bytes(int ndim, int64_t sx, int64_t sy, int64_t sz);
int edt_hip_edtsq_voxel_graph_device(const void *d_labels, int dtype, const uint8_t *d_graph, int ndim,
                                     int64_t sx, int64_t sy, int64_t sz, float wx, float wy, float wz,
                                     int flags, float *d_output, void *d_workspace, size_t workspace_bytes,
                                     void *stream);

/* out[i] = (labels[i] == *key) ? dt[i] : 0 -- the image edt.each() yields for one label
 * (src/edt.pyx:950-994: zeros + transfer of the label's runs, src/edt_voxel_graph.hpp:290-310), as one
 * streaming kernel on device-resident data.  `key` is a HOST pointer to one value of the label dtype. */
int edt_hip_select_label_device(const void *d_labels, int dtype, const float *d_dt, const void *key,
                                float *d_out, int64_t count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* EDT_HIP_H */
