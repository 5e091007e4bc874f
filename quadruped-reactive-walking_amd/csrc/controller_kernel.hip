// Controller glue between the planners, the MPC and the WBC — gfx950 (MI355X).  SURVEY.md §8(f) rank 3.
//
// Replaces, per instance, the element-wise parts of Controller.compute (/root/reference/scripts/Controller.py):
//   updateState                :381-426  perfect x/y/yaw integration, horizontal-frame velocity, oRh / oTh
//   WBC target assembly        :258-296  x_f_wbc, q_wbc, b_v, foot commands in the base frame
//   result + security_check    :306-310, :341-365
//   (+ the force-row shift of MPC_Wrapper.solve, scripts/MPC_Wrapper.py:89-102, used by the asynchronous MPC mode)
// so that a whole control iteration (planners -> MPC -> WBC -> PD targets) stays in HBM.
// One thread per instance; persistent state item-major cs[item][instance].
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "controller_glue.h"
#include "qrw_kernels.h"

namespace qrw {

__global__ __launch_bounds__(64) void controller_kernel(ControllerArgs a) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= a.B) return;
  if (a.mode == kCtrlInit) {
    const glue::CS s = glue::state_of(a, b);
    for (int i = 0; i < kCtrlStItems; i++) s(i) = 0.0;
    for (int i = 0; i < 12; i++) s(glue::cQDES + i) = a.in0 ? a.in0[(size_t)b * 12 + i] : 0.0;  // myController.qdes[7:] = q_init
  } else if (a.mode == kCtrlUpdateState) {
    glue::update_state(a, b);
  } else if (a.mode == kCtrlWbcInputs) {
    glue::wbc_inputs(a, b);
  } else if (a.mode == kCtrlResult) {
    glue::result(a, b);
  } else if (a.mode == kCtrlMpcShift) {
    glue::mpc_result_shift(a, b);
  }
}

int controller_launch(const ControllerArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(controller_kernel, dim3((a.B + 63) / 64), dim3(64), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace qrw
