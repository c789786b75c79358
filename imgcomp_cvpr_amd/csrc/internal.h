// Internal (non-ABI) declarations shared between the translation units of libimgcomp_hip.so.
#pragma once
#include "common.h"

struct ConvArgs {
    const float* x; const float* w; const float* scale; const float* shift;
    const float* res1; const float* res2; float* y;
    const float* in_mean; const float* in_std; const float* out_mean; const float* out_std;
    int N, Cin, H, W, Cout, OH, OW, KH, KW, stride, pt, pl, relu;
    int w_sci, w_sco;   // filter strides (in floats) of the ci and co axes inside one tap
    int tune;           // h13 kernel: tiles per work-group (0 = automatic); from the caller's per-call flags
    int builtin_norm;   // bit 0: normalise the input with the reference's fixed image statistics,
                        // bit 1: de-normalise + clip the output with them (autoencoder.py:136-169), bit 2: clip only
    int out_phases;     // h1 kernel: 1 = write the output as four phase planes [N][4][Cout][OH/2][OW/2] (plane 2 py + px holds the
                        // pixels (2 i + py, 2 j + px)): what h2 reads in its F(4x4)-over-phases form (conv3x3_wino4.hip); OH, OW even
};

// fills OH/OW/pads/filter strides for a TF-SAME conv (transposed = stride-2 conv2d_transpose) and launches
int icx_conv2d(ConvArgs a, bool transposed, hipStream_t st);

// 5x5 / stride-2 transposed conv with <= 4 output channels (h13); `a` already completed by icx_conv2d's
// transposed branch.  IC_ERR_UNSUPPORTED when the shape does not fit.
int icx_deconv5_small_cout(const ConvArgs& a, hipStream_t st);

// the same layer for Cin = 64 on the matrix cores
int icx_deconv5_cout3_mfma(const ConvArgs& a, hipStream_t st);

// 3x3 / stride-2 transposed conv Cin (32 | 64) -> 128 on the matrix cores (from_bn); same contract
int icx_deconv3_mfma(const ConvArgs& a, hipStream_t st);

// 5x5 / stride-2 conv 3 -> 64 with optional input normalisation on the matrix cores (h1); same contract
int icx_conv5s2_cin3_mfma(const ConvArgs& a, hipStream_t st);

// probclass.hip: data gradient of a masked conv3d layer on the matrix cores (train_pc.hip); workspace 0 = shape not covered
size_t icx_pc_bwd_data_mfma_workspace(int N, int CinF, int CoutF, int OD, int OH, int OW);
int icx_pc_bwd_data_mfma(const float* g, const float* w, float* dx_raw, int N, int CinF, int CoutF, int OD, int OH, int OW,
                         const float* zero_bias, void* workspace, size_t workspace_bytes, hipStream_t st);
