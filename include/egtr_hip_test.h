/*
 * egtr_hip_test.h -- TEST-ONLY entry points of libegtr_hip.so: explicit kernel-variant selection for A/B parity tests and
 * benchmarks (tests/test_gpu_kernels.py, tools/msda_bench.py).  Not part of the drop-in boundary (include/egtr_hip.h): a
 * product caller never chooses a kernel variant.
 */
#ifndef EGTR_HIP_TEST_H
#define EGTR_HIP_TEST_H

#include "egtr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* egtr_msda_forward_f32 with an explicit kernel choice (A/B parity tests): 0 = automatic, 1 = the wave-per-query
 * kernel (M = 8, D = 32, L*P = 16), 3 = the generic one-thread-per-element kernel (any shape).  Both compute the same
 * function; EGTR_E_UNSUPPORTED if the shape rules out the requested kernel. */
int egtr_msda_forward_f32_variant(egtr_stream_t stream, const float* value, const int64_t* spatial_shapes,
                                  const int64_t* level_start_index, const float* sampling_loc,
                                  const float* attn_weight, int batch, int spatial_size, int num_heads, int channels,
                                  int num_levels, int num_query, int num_point, float* out, int variant);

/* Same with an explicit kernel choice (A/B parity tests): 0 = automatic, 1 = wave-per-query with one global atomic per
 * (sample, corner, channel) like the reference, 2 = grad_attn / grad_loc by the wave-per-query kernel + grad_value as a
 * dense product per (query tile, head) on the matrix cores (encoder-shaped calls), 3 = generic. */
int egtr_msda_backward_f32_variant(egtr_stream_t stream, const float* grad_out, const float* value,
                                   const int64_t* spatial_shapes, const int64_t* level_start_index,
                                   const float* sampling_loc, const float* attn_weight, int batch, int spatial_size,
                                   int num_heads, int channels, int num_levels, int num_query, int num_point,
                                   float* grad_value, float* grad_sampling_loc, float* grad_attn_weight, int variant);

/* Fault injection for egtr_decoder_layer_f32 (tests/test_gpu_decoder_cluster.py): while `on` != 0, one wave of cluster 0
 * skips its second barrier arrival in every launch, so that barrier times out (status bit 0; the waves that gave up
 * NaN-poison the rows they hand out).  The barrier counters of the workspace are left short by one per launch: zero them
 * (the host binding does when it reports the status) before the next healthy launch. */
int egtr_test_decoder_drop_arrival(int on);

/* Measurement probe (csrc/probe_l1.hip): `blocks` workgroups of 256 threads, each gathering `iters` x 16 x 16 bytes per lane
 * from its own L1-resident 12 KiB region, every 8-lane group a different 128-byte line -- the MSDA gather's access shape
 * without misses.  `buf`: egtr_test_l1_gather_buffer_bytes(blocks) bytes of zeros; `out`: >= 256 floats (never written on a
 * zero buffer); *bytes_moved: bytes returned to registers by the launch.  bench.py times it next to the MSDA kernel
 * (roofline.l1_gather_ceiling_gbs). */
long long egtr_test_l1_gather_buffer_bytes(int blocks);
int egtr_test_l1_gather_bandwidth(egtr_stream_t stream, const void* buf, float* out, int iters, int blocks,
                                  long long* bytes_moved);

#ifdef __cplusplus
}
#endif
#endif /* EGTR_HIP_TEST_H */
