// Which XCDs / compute units does a hipExtStreamCreateWithCUMask stream run on?  Every workgroup records the XCC_ID and
// HW_ID of the wavefront that ran it; the host prints, per mask, how many distinct (XCD, SE, CU) triples were seen per XCD.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/cu_mask_probe.hip -o build/cu_mask_probe && build/cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <set>
#include <vector>

__global__ void probe(uint32_t* out, int spin) {
  uint32_t xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  // keep the workgroup resident for a while so that the whole masked partition fills up
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}

static void run(const char* name, const std::vector<int>& bits, int total_cus) {
  const int words = (total_cus + 31) / 32;
  std::vector<uint32_t> mask(words, 0u);
  for (int b : bits) mask[b >> 5] |= 1u << (b & 31);
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, words, mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
  const int blocks = 4096;
  uint32_t* d;
  hipMalloc(&d, blocks * 2 * sizeof(uint32_t));
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, s, d, 200000);
  hipStreamSynchronize(s);
  std::vector<uint32_t> h(blocks * 2);
  hipMemcpy(h.data(), d, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
  std::set<uint32_t> cus[16];
  for (int i = 0; i < blocks; i++) {
    const uint32_t xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;  // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    cus[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  printf("%-34s (%3zu mask bits): distinct compute units seen per XCD:", name, bits.size());
  int tot = 0;
  for (int x = 0; x < 8; x++) { printf(" %2zu", cus[x].size()); tot += (int)cus[x].size(); }
  printf("  total %d\n", tot);
  hipFree(d);
  hipStreamDestroy(s);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int T = p.multiProcessorCount;
  printf("%s: %d compute units\n", p.name, T);
  std::vector<int> v;
  for (int i = 0; i < T; i++) v.push_back(i);
  run("all bits", v, T);
  v.clear(); for (int i = 0; i < 32; i++) v.push_back(i);
  run("bits 0..31", v, T);
  v.clear(); for (int i = 32; i < T; i++) v.push_back(i);
  run("bits 32..255", v, T);
  v.clear(); for (int i = 0; i < 16; i++) v.push_back(i);
  run("bits 0..15", v, T);
  v.clear(); for (int i = 0; i < T; i += 8) v.push_back(i);
  run("bits 0,8,16,... (every 8th)", v, T);
  v.clear(); for (int i = 0; i < T; i++) if (i % 8 != 0) v.push_back(i);
  run("all but every 8th", v, T);
  v.clear(); for (int i = 0; i < 8; i++) v.push_back(i);
  run("bits 0..7", v, T);
  return 0;
}
