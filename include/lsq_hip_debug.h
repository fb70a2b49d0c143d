/*
 * lsq_hip_debug.h -- test hooks of liblsq_hip.so.  NOT part of the product ABI (lsq_hip.h): nothing on the
 * inference path reads or needs them, the defaults (all 0) are what ships, and a maintainer binding the library into
 * the reference would not declare them.
 *
 * They exist so that the parity tests can run the SAME call through an alternative implementation of one entry point
 * and compare the results bit for bit (tests/test_gpu_parity.py: integer-MFMA vs popcount convolution, single-launch
 * vs three-kernel quantizer, the quantizer's rare block / overflow paths on ordinary data).
 *
 * The switches are process-wide relaxed atomics: setting one while another thread has a call in flight makes THAT
 * call take either implementation (both give identical results); there is no other interaction.  Each setter returns
 * the previous value so that callers can restore it (quant._hip.debug_switches does, in a finally block).
 */
#ifndef LSQ_HIP_DEBUG_H_
#define LSQ_HIP_DEBUG_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* lsq_xnor_conv2d: 0 (default) the matrix-core kernel where it applies -- fp4 operands on v_mfma_scale_f32_32x32x64_f8f6f4 --,
 * 1 the popcount kernel for every geometry, 2 the int8 matrix-core kernel of rounds 2-5; all three give the same bits */
int lsq_debug_xnor_impl(int impl);
/* 1: lsq_act_quant / lsq_solve_rows take the streaming three-kernel path for every shape (default 0) */
int lsq_debug_force_streaming(int on);
/* single-launch quantizer, bit mask (default 0): 1 = every flagged bin of the round-2 solve through its block path, 2 = its
 * key list holds 2048 entries, 4 = no windowed level-1 histogram (the round-2 solve alone, as for rows without a symmetric
 * clamp), 8 = windowed histogram built, then every row handed to the fall-back (the call into the round-2 body) */
int lsq_debug_fused_mode(int mode);

/* device buffer of one int32 per row (sized by the caller for its largest batch; NULL = off, the default): every
 * LS-2 / LS-T solve -- lsq_act_quant and lsq_solve_rows, single-launch and streaming path -- stores the SORTED POSITION of
 * the candidate it chose (quant/binary/optimal.py:151: the argmin's element of the ascending sub-sample; the first
 * position of its run when several keys are equal), n + 1 for the ternary extra candidate (optimal.py:86-118), -1 when
 * the row has no candidate.  Lets the parity tests pin the candidate itself, not only its value.  Returns 0. */
int lsq_debug_solver_trace(int32_t* device_rows);

#ifdef __cplusplus
}
#endif
#endif /* LSQ_HIP_DEBUG_H_ */
