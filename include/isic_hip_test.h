/* Test / benchmark-only entry points of libisic_hip.so -- NOT part of the drop-in ABI (include/isic_hip.h).
 *
 * They exist so that tests and tools/kernel_bench.py can pin the kernel an ABI entry dispatches to without any
 * global state in the library: the choice travels with the call.  The product path never uses them. */
#ifndef ISIC_HIP_TEST_H
#define ISIC_HIP_TEST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* isic_conv2d_igemm_bf16 (same arguments, same reference citation: the encoder convolutions of
 * BASELINE.json configs[1], save_latent.py:42-60) with the kernel choice pinned by `variant`, decimal digits:
 *   units    staging scheme of the generic implicit GEMM + 1 (0 = shipped default; 1..6 = MODE 0..5 of conv_igemm.hip)
 *   tens     64 -> 64 3x3 layers: 0 shipped default, 1 generic kernel, 2 tile-per-block halo kernel, 3 persistent halo kernel
 *   hundreds >= 128-channel 3x3 stride-1 layers (conv_halo.hip, pixels staged once for all nine taps):
 *            0 shipped default, 1 never, 2 wherever the kernel supports the shape
 *   thousands persistent short-K kernel (conv_pgemm.hip: strided 3x3, its data gradient, 1x1 downsample):
 *            0 shipped default, 1 never, 2 wherever the kernel supports the shape (Cout % 128 == 0)
 *   ten-thousands  conv_halo.hip: 0 = shipped K loop (one barrier per two K-tiles where Cin % 128 == 0), 1 = the round-3 loop
 *            (one barrier per K-tile, three weight stages); same products in the same order: bit-equal (tools/halo_ab.py)
 *            conv_c64.hip (persistent kernel): 1 = the non-temporal output stores of rounds 2-3 instead of ordinary ones (bit-equal);
 *            2..7 on either kernel = parts compiled out (timing ablations: the results are garbage)
 * variant = 0 is exactly isic_conv2d_igemm_bf16. */
int isic_test_conv2d_igemm_variant_bf16(const uint16_t* in, const uint16_t* w, uint16_t* out, int N, int Hin, int Win,
                                        int Cin, int Hout, int Wout, int Cout, int Kh, int Kw, int up, int down, int pad,
                                        const uint16_t* addend, double* stat_sum, double* stat_sumsq, int stat_slots,
                                        int variant, void* stream);

/* isic_conv2d_dgrad_pair_bf16 (same arguments) with the kernel choice pinned by `variant` as above (thousands digit 2: the
 * persistent short-K kernel also for 64-channel outputs). */
int isic_test_conv2d_dgrad_pair_variant_bf16(const uint16_t* dy, const uint16_t* w, const uint16_t* dy2, const uint16_t* w2,
                                             uint16_t* dx, int N, int Ho, int Wo, int Co, int H, int W, int C, int variant,
                                             void* stream);

/* isic_conv2d_wgrad_bf16 (same arguments) with `variant` bits: 0 = shipped; 16 = the 32-output-channel all-taps kernel
 * (conv_wgrad_c128.hip) where the 64-channel one (conv_wgrad_c128b.hip) ships; on that kernel: 1 = the block order it does
 * NOT ship with (XCD-grouped co-slice blocks vs pair-major), 2 / 4 / 8 = MFMAs / fragment reads / LDS-DMA compiled out
 * (timing ablations: the results are garbage).  Stride-2 3x3 layers: 16 = the per-tap kernel (conv_wgrad.hip) where the
 * all-taps strided kernel (conv_wgrad_s2.hip) ships, 32 = the all-taps strided kernel also where the per-tap one ships
 * (more than four (64 ci, 128 co) pairs). */
int isic_test_conv2d_wgrad_variant_bf16(const uint16_t* x, const uint16_t* dy, float* dw, int N, int Hin, int Win, int Cin,
                                        int Hout, int Wout, int Cout, int Kh, int Kw, int stride, int pad, void* workspace,
                                        size_t workspace_bytes, int variant, void* stream);

/* isic_gemm_f32_ws (same arguments) with the kernel pinned: variant 0 = shipped choice, 1 = always the 64 x 64 x 16
 * register-staged kernel (split-K through fp32 atomics), 2 = the persistent 256 x 128 x 32 kernel or
 * ISIC_ERR_UNSUPPORTED when the shape is not for it, 3 = the register-fed A^T B split-K kernel (small output, long
 * reduction) or ISIC_ERR_UNSUPPORTED, 4 = the row-panel kernel (long M, K <= 128) or ISIC_ERR_UNSUPPORTED.  A/B timing (tools/gemm_bench.py) and tests only. */
int isic_test_gemm_f32_variant(int variant, int transA, int transB, int M, int N, int K, const float* A, int lda,
                               const float* B, int ldb, float* C, int ldc, const float* bias, int act, float beta,
                               void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
