// Host cost of enqueueing a control iteration's two kernels: two hipLaunchKernelGGL calls against one hipGraphLaunch of the
// captured pair (64 workgroups of 64 threads each, 30 pointer-sized arguments like the real kernels).
// hipcc --offload-arch=gfx950 -O2 -o build/graph_launch scripts/ubench/graph_launch.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Args { double* p[30]; int k; };
__global__ void ka(Args a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[0][0] += (double)a.k; }
__global__ void kb(Args a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[1][0] += 1.0; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  double* d; hipMalloc(&d, 4096);  hipMemset(d, 0, 4096);
  Args a; for (int i = 0; i < 30; i++) a.p[i] = d + i; a.k = 1;
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int R = 2000;
  for (int rep = 0; rep < 3; rep++) {
    hipStreamSynchronize(s);
    double t0 = now();
    for (int i = 0; i < R; i++) { hipLaunchKernelGGL(ka, dim3(64), dim3(64), 0, s, a); hipLaunchKernelGGL(kb, dim3(64), dim3(64), 0, s, a); }
    double t1 = now();
    hipStreamSynchronize(s);
    double t2 = now();
    printf("direct: %.2f us host per pair (enqueue), %.2f us per pair incl. drain\n", (t1 - t0) / R * 1e6, (t2 - t0) / R * 1e6);
  }
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  hipLaunchKernelGGL(ka, dim3(64), dim3(64), 0, s, a); hipLaunchKernelGGL(kb, dim3(64), dim3(64), 0, s, a);
  hipStreamEndCapture(s, &g);
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
  for (int rep = 0; rep < 3; rep++) {
    hipStreamSynchronize(s);
    double t0 = now();
    for (int i = 0; i < R; i++) hipGraphLaunch(ge, s);
    double t1 = now();
    hipStreamSynchronize(s);
    double t2 = now();
    printf("graph : %.2f us host per pair (enqueue), %.2f us per pair incl. drain\n", (t1 - t0) / R * 1e6, (t2 - t0) / R * 1e6);
  }
  // paced: one pair, wait, next (latency of a lone iteration)
  for (int mode = 0; mode < 2; mode++) {
    double tot = 0;
    for (int i = 0; i < 500; i++) {
      double t0 = now();
      if (mode == 0) { hipLaunchKernelGGL(ka, dim3(64), dim3(64), 0, s, a); hipLaunchKernelGGL(kb, dim3(64), dim3(64), 0, s, a); }
      else hipGraphLaunch(ge, s);
      hipStreamSynchronize(s);
      tot += now() - t0;
    }
    printf("%s paced: %.2f us per pair launch + completion\n", mode ? "graph " : "direct", tot / 500 * 1e6);
  }
  double h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("check %.0f %.0f\n", h[0], h[1]);
  return 0;
}
