// cczero_experiments.hip -- the shelved forms of the tower convolution (cczero_conv2.h, cczero_conv3.h, cczero_conv4.h) as their
// OWN translation unit and library: build/diag/libcczero_experiments.so (make -C chinesechesszero_amd/csrc experiments). Nothing
// here is part of the product: libcczero.so is compiled from csrc/cczero.hip alone, which no longer knows these kernels
// ("drop the hooks from the product translation unit"). The one exported entry point has the signature of
// ccz_conv3x3_c256_f16, so that profiles/conv_ab.py can time a shelved form next to the shipped one:
//     python profiles/conv_ab.py libcczero.so:1 libcczero_experiments.so:5      (bit 2: v2, two workgroups per CU, round 2)
//     python profiles/conv_ab.py libcczero.so:1 libcczero_experiments.so:9      (bit 3: v3, no barrier per half-step, round 3)
//     python profiles/conv_ab.py libcczero.so:65 libcczero_experiments.so:1025  (bit 10: v4, group-of-16 rows, scalar guards)
// Measured results: profiles/r02_conv_v2.json, r03_conv_v3.json, r03_conv_g16.json. The experiments keep the summation order
// they were written with (64-channel chunks): their outputs agree with the shipped kernels to float32 round-off, not bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cczero_conv2.h"
#include "cczero_conv3.h"
#include "cczero_conv4.h"

using namespace ccz;

extern "C" int ccz_abi_version(void) { return -1; } // not a product library

extern "C" int ccz_conv3x3_c256_f16(void *stream, const void *x_dev, const void *w_dev, const void *bias_f32_dev, const void *residual_dev,
                                    void *y_dev, int64_t n_pixels, int32_t relu)
{
    if (!x_dev || !w_dev || !bias_f32_dev || !y_dev || n_pixels <= 0) return -1;
    const int cin = 256;
    const unsigned tiles = (unsigned)((n_pixels + kCvBM - 1) / kCvBM);
    const _Float16 *X = (const _Float16 *)x_dev, *W = (const _Float16 *)w_dev, *R = (const _Float16 *)residual_dev;
    const float *B = (const float *)bias_f32_dev;
    _Float16 *Y = (_Float16 *)y_dev;
    hipStream_t s = (hipStream_t)stream;
    if (relu & 4) { // v2: two 4-wave workgroups per CU
        const unsigned grid = ((tiles + 7) / 8) * 16;
        if (R) hipLaunchKernelGGL(k_conv3x3_v2<true>, dim3(grid), dim3(256), 0, s, X, W, B, R, Y, (int)n_pixels, (int)(relu & 3), cin, (int)tiles);
        else hipLaunchKernelGGL(k_conv3x3_v2<false>, dim3(grid), dim3(256), 0, s, X, W, B, (const _Float16 *)nullptr, Y, (int)n_pixels, (int)(relu & 3), cin, (int)tiles);
    } else if (relu & 8) { // v3: weights per wave from global memory, 8 barriers per tile
        if (R) hipLaunchKernelGGL(k_conv3x3_v3<true>, dim3(tiles), dim3(512), 0, s, X, W, B, R, Y, (int)n_pixels, (int)(relu & 3), cin);
        else hipLaunchKernelGGL(k_conv3x3_v3<false>, dim3(tiles), dim3(512), 0, s, X, W, B, (const _Float16 *)nullptr, Y, (int)n_pixels, (int)(relu & 3), cin);
    } else if (relu & 1024) { // v4: group-of-16 rows, per-cell scalar guards
        if (n_pixels % 1440) return -1;
        if (R) hipLaunchKernelGGL(k_conv3x3_v4<true>, dim3(tiles), dim3(512), 0, s, X, W, B, R, Y, (int)n_pixels, (int)(relu & 3), cin);
        else hipLaunchKernelGGL(k_conv3x3_v4<false>, dim3(tiles), dim3(512), 0, s, X, W, B, (const _Float16 *)nullptr, Y, (int)n_pixels, (int)(relu & 3), cin);
    } else
        return -1; // the shipped forms live in libcczero.so
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
