/*
 * include/qrw_hip_test.h -- entry points of libqrw_hip.so that exist for the TEST SUITE only (fault injection, register / LDS
 * poisoning).  Not part of the drop-in boundary: nothing a caller of the reference's interfaces needs is declared here, and a
 * deployment has no reason to include this file.  The product surface is include/qrw_hip.h.
 */
#ifndef QRW_HIP_TEST_H_
#define QRW_HIP_TEST_H_

#include "qrw_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tests only: leave every instance of a time-sliced handle (N > 16, batch above the resident slots) as a launch whose queue
 * gave up would -- parked at iteration `parked_at`, the warm-start slots holding values that are not OSQP's iterates -- so that
 * the cold restart of the next qrw_mpc_solve can be checked without provoking a give-up.  -1 on other handles. */
int qrw_test_poke_aborted(qrw_handle h, int32_t parked_at);

/* Tests only: (poison != 0) fill the LDS of every compute unit with `lds_pattern`, then run the known-answer solve of
 * mpc_solve_kernel for horizon N in launch form mode (0 one launch per call, 1 time-sliced: N > 16, 2 sequence) without the
 * per-process cache of qrw_create; 0 = the answer is right.  Catches reads of LDS the kernel has not written. */
int qrw_test_known_answer(int32_t N, int32_t mode, uint64_t lds_pattern, int32_t poison, int32_t *iters, int32_t *status,
                          double *rho, double *err);

/* Tests only: after the LDS / register-file fills of qrw_test_known_answer, what a new wavefront finds in an LDS word, v255 and a255
 * it has not written: h_out3[0..2], all-ones when the fills are effective on this device. */
int qrw_test_poison_probe(uint32_t *h_out3);

#ifdef __cplusplus
}
#endif
#endif /* QRW_HIP_TEST_H_ */
