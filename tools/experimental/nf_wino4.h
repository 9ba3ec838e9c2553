/* EXPERIMENT, not part of the product ABI: Winograd F(4x4,3x3) for the stride-1 3x3 convolutions (tools/experimental/nf_wino4.hip).
 * Measured 1.7x slower than the shipped F(2x2,3x3) kernel, kept for the record; built and driven by
 * tools/bench_wino4.py only. */
#pragma once
#include "../../include/nerfool_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* The same stride-1 3x3 convolution as nf_conv3x3_wino in Winograd F(4x4, 3x3) form (csrc/nf_wino4.hip: 36 products per 4x4 output
 * tile instead of 64; fp32 error 2-4e-6 of full scale).  records = nf_wino4_pack(weight [c_out][c_in][3][3], backward) (HOST
 * pointers, nf_wino4_pack_floats(outputs, inputs of the packed convolution) floats).  pad 0: forward on pre-padded input; pad 2:
 * backward-data on the gradient with records packed with backward != 0. */
int64_t nf_wino4_pack_floats(int c_out, int c_in);
int nf_wino4_pack(const float* weight_host, int c_out, int c_in, int backward, float* records_host);
int nf_conv3x3_wino4(const float* records, const float* x, int64_t xs_n, int64_t xs_c, int64_t xs_h, int Hi, int Wi, int pad, float* y,
                     int64_t ys_n, int64_t ys_c, int64_t ys_h, int Ho, int Wo, int n_img, int c_in, int c_out, nf_stream_t stream);
#ifdef __cplusplus
}
#endif
