// Which compute units does a bit of hipExtStreamCreateWithCUMask's mask select on MI355X?
// Launches a long-enough kernel on streams with different masks and histograms the XCC_ID (and the
// number of distinct (XCC, SE, CU) triples) its workgroups report.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>

__global__ void where_kernel(unsigned* xcc_hist, unsigned* ids, int spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) {
    atomicAdd(&xcc_hist[xcc & 7u], 1u);
    ids[blockIdx.x] = ((xcc & 7u) << 16) | (hwid & 0xffffu);
  }
  // stay resident for a while so that every allowed compute unit gets workgroups
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  printf("device: %s, %d compute units\n", prop.name, ncu);
  unsigned *hist, *ids;
  const int nblk = 4096;
  hipMalloc(&hist, 8 * sizeof(unsigned));
  hipMalloc(&ids, nblk * sizeof(unsigned));
  struct Case { const char* name; int mode; };
  Case cases[] = {{"no mask", 0}, {"bits with i % 8 == 7", 1}, {"bits 224..255", 2}, {"bits 0..31", 3}, {"all but i % 8 == 7", 4}};
  for (const Case& c : cases) {
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int i = 0; i < ncu; ++i) {
      bool on = true;
      if (c.mode == 1) on = (i & 7) == 7;
      if (c.mode == 2) on = i >= 224;
      if (c.mode == 3) on = i < 32;
      if (c.mode == 4) on = (i & 7) != 7;
      if (on) mask[i / 32] |= 1u << (i % 32);
    }
    hipStream_t s;
    hipError_t st = c.mode == 0 ? hipStreamCreateWithFlags(&s, hipStreamNonBlocking)
                                : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (st != hipSuccess) { printf("%-24s stream creation failed: %s\n", c.name, hipGetErrorString(st)); continue; }
    hipMemsetAsync(hist, 0, 8 * sizeof(unsigned), s);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s);
    hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(256), 0, s, hist, ids, 2000);  // 20 us per workgroup
    hipEventRecord(b, s);
    hipStreamSynchronize(s);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    unsigned h[8];
    std::vector<unsigned> hid(nblk);
    hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(hid.data(), ids, nblk * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::set<unsigned> cus;
    for (unsigned v : hid) cus.insert(((v >> 16) << 16) | (v & 0x0000ff00u) | ((v >> 13 & 7u) << 4));  // xcc | cu_id(11:8) | se_id(15:13)
    printf("%-24s %7.2f ms  per-XCC workgroups:", c.name, ms);
    for (int x = 0; x < 8; ++x) printf(" %4u", h[x]);
    printf("   distinct (xcc, se, cu): %zu\n", cus.size());
    hipStreamDestroy(s);
  }
  return 0;
}
