/* sda_hip_debug.h - TEST / MEASUREMENT ONLY entry points.  Not part of the drop-in boundary (include/sda_hip.h): nothing a Rust
 * shim binds.
 *
 * TWO LIBRARIES are built from the same objects (__graft_entry__.build(); only sda_capi.cpp is compiled twice):
 *   sda_amd/lib/libsda_hip.so       the RELEASE library.  Of this header it exports ONLY the two read-only queries
 *                                   sda_debug_last_kernel() and sda_debug_hooks_compiled_in() (= 0).  It has no knob table: no
 *                                   in-process caller can change which kernel a later handle gets, and it reads no environment variable.
 *   sda_amd/lib/libsda_hip_test.so  the same code with -DSDA_TEST_HOOKS: everything below (sda_debug_hooks_compiled_in() = 1).  What the
 *                                   parity tests of the NON-DEFAULT kernels and the A/B measurement scripts load
 *                                   (sda_amd.capi.use_test_hooks()); smoke(), bench.py's default line and the C / C++ examples never do.
 *
 * The release library reads no environment variable; the kernels behind one C-ABI call are all
 * bit-exact with each other, and which one serves a shape is the library's decision.  The parity tests still have to reach the
 * non-default kernels (the any-shape fallback, the 64-bit Montgomery form, the transform kernel on small tss-valid shapes, the
 * limb GEMM on shapes it is not the default for, both varint decode forms), and the A/B measurements of DESIGN.md have to
 * switch between them: that is what these knobs are for.  They are process-global and read when a handle is created (path
 * selection) or at the call (stream / grid choices).
 *
 * knob names (value 0 = default behaviour):
 *   SDA_FORCE_GENERIC 1        packed share generation through packed_gen_generic_kernel
 *   SDA_FORCE_MONT64 1         ... through the superseded 64-bit Montgomery kernel
 *   SDA_FORCE_FFT 1            the transform kernel for every tss-valid shape (default: k + t > 32)
 *   SDA_FORCE_MFMA 1 / SDA_NO_MFMA 1   the limb GEMM for every shape it covers / never
 *   SDA_NO_SIDE_STREAM 1       transform shapes: clerk sum on the caller's stream instead of the low-priority side stream
 *   SDA_SIDE_STREAM_WGS n, SDA_SIDE_STREAM_PRIORITY 1 (= high)   side-stream grid and priority
 *   SDA_FFT_G n, SDA_FFT_THREADS n   batches per workgroup / threads of the transform kernel
 *   SDA_VARINT_PATH 1 (stream) / 2 (scan)   pin one varint decode form
 *   SDA_NO_NARROW 1            primes below 2^31 through the 62-bit kernels too (default: the one-limb narrow kernels)
 *   SDA_NO_LAZY 1              narrow transform kernel with the conditional subtractions of the 64-bit form (default: lazy where it fits)
 *   SDA_WIRE_WG_PER_CU n, SDA_SBOX_WG_PER_CU n   residency caps (unused dynamic LDS) of the varint stream kernels / the XSalsa20 kernel
 *   SDA_NO_XCD_MAP 1           transform kernel with fewer than 8 batches per workgroup: plain group order (default: the
 *                              workgroups that share a 128-byte line of a clerk row are placed on one XCD)
 *   SDA_NO_NGEMM 1             large shapes over a prime below 2^23 through the transform kernel (default: the limb GEMM on the
 *                              matrix cores, ngemm_kernels.hip)
 *   SDA_NO_WIDE_GROUP 1        three-digit limb-31 shapes of 9 .. 12 terms (BASELINE config 4's (8,2)): the 7 + rest grouping even where
 *                              the constants admit the dot product as ONE group (default: one group, one reduction, no normalisation)
 *   SDA_NO_KARATSUBA 1         ... the one-group form with four multiply-adds per term even where the constants admit the Karatsuba form
 *                              (three per term, round 6)
 *   SDA_NGEMM_CLERK_WG 1       limb GEMM, dual-role launch: the clerk sum in clerk WORKGROUPS at fixed grid positions (rounds 4 - 5) instead
 *                              of the clerk WAVES inside every share-generation workgroup (round 6, the default)
 *   SDA_FORCE_COLLECTIVES 1    a one-rank communicator still goes through RCCL send/recv to itself
 * Built with -DSDA_AB_KNOBS (tools/build_ab_variant.sh; never by __graft_entry__.build()) an unset knob falls back to the
 * environment variable of the same name. */
#ifndef SDA_HIP_DEBUG_H
#define SDA_HIP_DEBUG_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* ---- in BOTH libraries (read-only) ---- */
int  sda_debug_hooks_compiled_in(void);                  /* 1 in libsda_hip_test.so, 0 in the release library */
/* ---- libsda_hip_test.so only (-DSDA_TEST_HOOKS), except sda_debug_last_kernel ---- */
int  sda_debug_set_knob(const char* name, long value);   /* SDA_ERR_INVALID_ARGUMENT for an unknown name */
void sda_debug_reset_knobs(void);
int  sda_debug_env_knobs_compiled_in(void);              /* 1 only in an SDA_AB_KNOBS build */
/* (BOTH libraries.)  What the library RAN: the kernel instance(s) launched by the last sda_share_generator_generate / _generate_batch_dev /
 * _generate_combine_dev call on this thread, named as rocprofv3 prints them ("fused_packed_l31_kernel<3, 1, 20>";
 * two launches: "packed_gen_fft_kernel<...> + combine_update_walk_kernel (side stream)").  bench.py prints this as roofline.kernel. */
const char* sda_debug_last_kernel(void);
/* The kernel-selection table without a device or a handle (sda_amd/csrc/path_select.hpp: the ONE place the decision is made):
 * `knobs` = comma-separated selection knob names from the list above (NULL or "" = defaults; the process-wide knob state is not
 * read).  Writes "wide=... narrow=... r_bits=... call20=... call12=... injected=... fused20=... fused12=... transform_shape=.
 * eight_term_ok=." (family names: additive n31 ngemm l31 mont64 l31_global fft mfma generic). */
/* streams and memory figures for the tests, so that they need no second HIP binding in their process (a Python process that
 * loads this library and then PyTorch ends up with two HIP runtimes): non-blocking streams, hipMemGetInfo */
int  sda_debug_stream_create(void** stream);
int  sda_debug_stream_destroy(void* stream);
int  sda_debug_stream_synchronize(void* stream);
int  sda_debug_mem_info(size_t* free_bytes, size_t* total_bytes);
struct sda_sharing_scheme;
int  sda_debug_select_path(const struct sda_sharing_scheme* scheme, const char* knobs, char* out, size_t cap);
#ifdef __cplusplus
}
#endif
#endif
