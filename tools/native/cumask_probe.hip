// Which CUs does a CU-masked stream use on this machine, and what does it cost?  (development probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>
typedef double d4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters) {
  d4_t acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (d4_t){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 - a;
  for (int it = 0; it < iters; ++it)
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  if (threadIdx.x == 0) {
    unsigned hwid = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x] = ((unsigned long long)(xcc & 0xf) << 32) | hwid;
  }
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678) out[0] = 1;
}

static void run(const char* name, hipStream_t s, unsigned long long* d, std::vector<unsigned long long>& h, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, s, d, 64);
  hipEventRecord(e0, s);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, s, d, 2048);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  std::set<unsigned> cus;
  int per_xcc[16] = {0};
  std::set<unsigned> per_xcc_cu[16];
  for (int i = 0; i < blocks; ++i) {
    unsigned hw = (unsigned)h[i], xcc = (unsigned)(h[i] >> 32);
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;   // gfx9 HW_ID: CU_ID[11:8] SH_ID[12] SE_ID[15:13]
    unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    cus.insert(key);
    per_xcc[xcc]++;
    per_xcc_cu[xcc].insert(key);
  }
  double tf = (double)blocks * 4 * 2048.0 * 4 * 2048.0 / (ms * 1e-3) / 1e12;
  printf("%-28s %7.3f ms %6.1f TF/s  distinct CUs %3zu  per-XCC CUs:", name, ms, tf, cus.size());
  for (int x = 0; x < 8; ++x) printf(" %zu", per_xcc_cu[x].size());
  printf("\n");
  if (getenv("PROBE_VERBOSE")) {
    for (unsigned k : cus) printf("  xcc %u se %u sh %u cu %u\n", k >> 12, (k >> 8) & 0xf, (k >> 4) & 0xf, k & 0xf);
  }
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("device %s, CUs %d\n", prop.name, prop.multiProcessorCount);
  const int blocks = 4096;
  unsigned long long* d;
  hipMalloc(&d, blocks * 8);
  std::vector<unsigned long long> h(blocks);
  hipStream_t s0;
  hipStreamCreate(&s0);
  run("plain stream", s0, d, h, blocks);
  struct Cfg { const char* name; std::vector<uint32_t> mask; };
  std::vector<Cfg> cfgs;
  cfgs.push_back({"all 256 bits", std::vector<uint32_t>(8, 0xffffffffu)});
  { std::vector<uint32_t> m(8, 0xffffffffu); m[7] = 0x00ffffffu; cfgs.push_back({"bits 0..247", m}); }
  { std::vector<uint32_t> m(8, 0xffffffffu); m[0] = 0xffffff00u; cfgs.push_back({"bits 8..255", m}); }
  { std::vector<uint32_t> m(8, 0u); m[0] = 0xffu; cfgs.push_back({"bits 0..7 only", m}); }
  { std::vector<uint32_t> m(8, 0u); m[0] = 0xffffffffu; cfgs.push_back({"bits 0..31 only", m}); }
  { std::vector<uint32_t> m(8, 0u); for (int i = 0; i < 256; i += 8) m[i / 32] |= 1u << (i % 32); cfgs.push_back({"every 8th bit", m}); }
  { std::vector<uint32_t> m(8, 0xffffffffu); for (int i = 0; i < 8; ++i) m[0] &= ~(1u << i); m[0] |= 0; cfgs.push_back({"all but bits 0..7", m}); }
  { std::vector<uint32_t> m(8, 0xffffffffu); for (int i = 0; i < 256; i += 32) m[i / 32] &= ~1u; cfgs.push_back({"all but every 32nd bit", m}); }
  for (auto& c : cfgs) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)c.mask.size(), c.mask.data());
    if (e != hipSuccess) { printf("%-28s create failed: %s\n", c.name, hipGetErrorString(e)); continue; }
    run(c.name, s, d, h, blocks);
    hipStreamDestroy(s);
  }
  return 0;
}
