/*
 * examples/cabi_demo.c — the drop-in boundary from plain C: no Python, no torch, no HIP headers.
 *
 * What a C/C++ caller on the reference side (e.g. a replacement for /root/reference/src/main.cpp, which drives
 * MPC::run directly, or the Boost.Python module of python/gepadd.cpp) does with libqrw_hip.so: create a handle,
 * hand over host buffers in the reference's own shapes, read the result and the solver statistics.
 *
 * Scenario: the reference's own four-stance known answer (scripts/test_mpc.py:54-85) for B robots standing on four
 * feet: the first call (num_iter = 0, MPC::run's set-up call, src/MPC.cpp:636-637) must converge in 350 ADMM
 * iterations to equal vertical forces whose sum carries the robot (2.5 kg * 9.81 m/s^2 = 24.53 N).
 *
 *   gcc -Iinclude examples/cabi_demo.c -o build/cabi_demo -Lquadruped-reactive-walking_amd -lqrw_hip -lm \
 *       -Wl,-rpath,$PWD/quadruped-reactive-walking_amd
 *   build/cabi_demo [B]        (exit code 0 = the known answer came out; needs an MI355X)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "qrw_hip.h"

#define N 16
#define NG 20

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 3;
  if (B < 1) return 2;
  qrw_config cfg = {B, N, NG, 0, 0.02, 0.32, 0.002};
  qrw_handle h = NULL;
  if (qrw_create(&cfg, &h) != 0) {
    fprintf(stderr, "qrw_create: %s\n", qrw_last_error());
    return 3;
  }
  double *xref = calloc((size_t)B * 12 * (N + 1), sizeof(double));
  double *fsteps = calloc((size_t)B * NG * 12, sizeof(double));
  double *out = calloc((size_t)B * 24 * N, sizeof(double));
  int32_t *iters = calloc(B, sizeof(int32_t)), *status = calloc(B, sizeof(int32_t));
  double *rho = calloc(B, sizeof(double));
  const double feet[12] = {0.195, 0.147, 0., 0.195, -0.147, 0., -0.195, 0.147, 0., -0.195, -0.147, 0.};
  for (int b = 0; b < B; b++) {
    for (int c = 0; c <= N; c++) xref[((size_t)b * 12 + 2) * (N + 1) + c] = 0.24474949993103629; /* height, every column */
    for (int k = 0; k < N; k++)
      for (int i = 0; i < 12; i++) fsteps[((size_t)b * NG + k) * 12 + i] = feet[i];               /* four feet down */
  }
  int rc = qrw_mpc_solve_host(h, xref, fsteps, NULL, 0, out);
  if (rc == 0) rc = qrw_mpc_get_stats(h, iters, status, rho, NULL, NULL);
  if (rc != 0) {
    fprintf(stderr, "solve: %s\n", qrw_last_error());
    return 4;
  }
  int bad = 0;
  for (int b = 0; b < B; b++) {
    const double *f = out + ((size_t)b * 24 + 12) * N; /* force rows, column 0 = the forces to apply now */
    double fz = 0.0, fxy = 0.0;
    for (int j = 0; j < 4; j++) {
      fz += f[(3 * j + 2) * N];
      fxy += fabs(f[(3 * j) * N]) + fabs(f[(3 * j + 1) * N]);
    }
    const int ok = iters[b] == 350 && status[b] == QRW_STATUS_SOLVED && fabs(fz - 24.534781284726584) < 1e-8 && fxy < 1e-8 &&
                   fabs(rho[b] / 1.0390579258297492e-3 - 1.0) < 1e-8;
    printf("robot %d: %d ADMM iterations, status %d, rho %.6e, sum f_z %.9f N, sum |f_xy| %.1e  %s\n", b, iters[b], status[b],
           rho[b], fz, fxy, ok ? "ok" : "MISMATCH");
    bad += !ok;
  }
  qrw_destroy(h);
  free(xref); free(fsteps); free(out); free(iters); free(status); free(rho);
  return bad ? 1 : 0;
}
