// C++ consumer of the sharded joint likelihood (no Python, no torch): F independent GP blocks are dealt to the R GPUs of
// one node by longest-processing-time-first, every GPU evaluates its blocks with ffgp_nlml_fused_async (one handle per
// GPU), and ONE ffgp_allreduce_sum of the F-vector per step (RCCL, one communicator per GPU from ncclCommInitAll) gives
// every GPU the joint value -- the reference's `loss += cigp_list[f].compute_loss(...)`, MFGP_ver2023May/ResGP.py:235,245.
// One process drives all GPUs (the collective is wrapped in ncclGroupStart/End), so no launcher is needed:
//
//   hipcc -O2 --offload-arch=gfx950 -I include -I /opt/rocm/include examples/joint_nll_rccl.cpp -L fidelityfusion_amd -lffgp \
//         -L /opt/rocm/lib -lrccl -Wl,-rpath,$PWD/fidelityfusion_amd -o /tmp/joint_nll_rccl
//   /tmp/joint_nll_rccl [ranks=all GPUs] [F=8] [N=2048] [D=8] [d=4] [steps=3]
//
// Exit code 0 iff every GPU holds the same F-vector after the all-reduce and it equals the values computed block by block.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ffgp.h"

#define CHK(x)                                                                                 \
  do {                                                                                         \
    if ((x) != 0) {                                                                            \
      fprintf(stderr, "%s failed at %s:%d\n", #x, __FILE__, __LINE__);                         \
      return 2;                                                                                \
    }                                                                                          \
  } while (0)

static double lcg(unsigned long long& s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(s >> 11) / 9007199254740992.0;
}

struct Block {
  int owner = 0;
  double *X = nullptr, *Y = nullptr, *w = nullptr, *amp = nullptr, *dadd = nullptr;
};

int main(int argc, char** argv) {
  int ndev = 0;
  CHK(hipGetDeviceCount(&ndev));
  const int R = std::max(1, std::min(argc > 1 ? atoi(argv[1]) : ndev, ndev));
  const int F = argc > 2 ? atoi(argv[2]) : 8, n = argc > 3 ? atoi(argv[3]) : 2048, D = argc > 4 ? atoi(argv[4]) : 8;
  const int d = argc > 5 ? atoi(argv[5]) : 4, steps = argc > 6 ? atoi(argv[6]) : 3;

  // one handle + one communicator + one F-vector per GPU
  std::vector<ffgp_handle*> h(R);
  std::vector<ncclComm_t> comm(R);
  std::vector<int> devs(R);
  std::vector<double*> vec(R);
  for (int r = 0; r < R; ++r) devs[r] = r;
  CHK(ncclCommInitAll(comm.data(), R, devs.data()));
  for (int r = 0; r < R; ++r) {
    CHK(hipSetDevice(r));
    CHK(ffgp_create(r, &h[r]));
    CHK(hipMalloc(&vec[r], F * sizeof(double)));
  }

  // blocks: equal cost here, so longest-processing-time-first is round robin; each lives on its owner's GPU
  std::vector<Block> blk(F);
  std::vector<double> load(R, 0.0);
  for (int f = 0; f < F; ++f) {
    const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
    load[r] += (double)n * n * n / 3.0 + (double)n * n * d;
    blk[f].owner = r;
    unsigned long long seed = 1000 + f;
    std::vector<double> X((size_t)n * D), Y((size_t)n * d), w(D);
    for (auto& v : X) v = lcg(seed);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < d; ++j) Y[(size_t)i * d + j] = sin(6.283185307179586 * X[(size_t)i * D + (j % D)] * (1 + j)) + 0.1 * (lcg(seed) - 0.5);
    for (int k = 0; k < D; ++k) w[k] = 0.8 + 0.04 * k + 0.01 * f;
    const double amp = 1.0 + 0.1 * f, dadd = exp(-1.0) + 1e-6;
    CHK(hipSetDevice(r));
    CHK(hipMalloc(&blk[f].X, X.size() * 8));
    CHK(hipMalloc(&blk[f].Y, Y.size() * 8));
    CHK(hipMalloc(&blk[f].w, D * 8));
    CHK(hipMalloc(&blk[f].amp, 8));
    CHK(hipMalloc(&blk[f].dadd, 8));
    CHK(hipMemcpy(blk[f].X, X.data(), X.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(blk[f].Y, Y.data(), Y.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(blk[f].w, w.data(), D * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(blk[f].amp, &amp, 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(blk[f].dadd, &dadd, 8, hipMemcpyHostToDevice));
  }

  std::vector<double> joint(F, 0.0);
  double ms = 0.0;
  for (int s = 0; s < steps; ++s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < R; ++r) {
      CHK(hipSetDevice(r));
      CHK(hipMemsetAsync(vec[r], 0, F * sizeof(double), nullptr));
      CHK(hipStreamSynchronize(nullptr));
    }
    // every GPU: its blocks, value written straight into its slot of the F-vector (enqueue only; blocks of different GPUs overlap)
    for (int f = 0; f < F; ++f) {
      const int r = blk[f].owner;
      CHK(hipSetDevice(r));
      ffgp_problem p = {};
      p.n = n; p.D = D; p.d = d;
      p.X_dev = blk[f].X; p.Y_dev = blk[f].Y; p.w_dev = blk[f].w; p.amp_dev = blk[f].amp;
      p.clamp_min = 1e-30;
      p.diag_add_dev = blk[f].dadd;
      p.ll_variant = FFGP_LL_V1;
      p.pi_const = 3.1415;
      CHK(ffgp_nlml_fused_async(h[r], &p, vec[r] + f, nullptr));
    }
    for (int r = 0; r < R; ++r) {     // a failing pivot would come back here
      CHK(hipSetDevice(r));
      CHK(ffgp_wait(h[r]));
    }
    // the one collective: sum of the F-vectors over the GPUs (each slot is non-zero on exactly one of them)
    CHK(ncclGroupStart());
    for (int r = 0; r < R; ++r) CHK(ffgp_allreduce_sum(h[r], comm[r], vec[r], F));
    CHK(ncclGroupEnd());
    for (int r = 0; r < R; ++r) {
      CHK(hipSetDevice(r));
      CHK(ffgp_wait(h[r]));
    }
    ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }

  // check: identical on every GPU, equal to the block-by-block values
  int bad = 0;
  for (int r = 0; r < R; ++r) {
    std::vector<double> got(F);
    CHK(hipSetDevice(r));
    CHK(hipMemcpy(got.data(), vec[r], F * sizeof(double), hipMemcpyDeviceToHost));
    if (r == 0) joint = got;
    for (int f = 0; f < F; ++f)
      if (got[f] != joint[f] || !std::isfinite(got[f])) ++bad;
  }
  double total = 0.0;
  for (int f = 0; f < F; ++f) {
    const int r = blk[f].owner;
    CHK(hipSetDevice(r));
    double* one = nullptr;
    CHK(hipMalloc(&one, 8));
    ffgp_problem p = {};
    p.n = n; p.D = D; p.d = d;
    p.X_dev = blk[f].X; p.Y_dev = blk[f].Y; p.w_dev = blk[f].w; p.amp_dev = blk[f].amp;
    p.clamp_min = 1e-30;
    p.diag_add_dev = blk[f].dadd;
    p.ll_variant = FFGP_LL_V1;
    p.pi_const = 3.1415;
    CHK(ffgp_nlml_fused(h[r], &p, one, nullptr));
    double v = 0.0;
    CHK(hipMemcpy(&v, one, 8, hipMemcpyDeviceToHost));
    if (std::fabs(v - joint[f]) > 1e-12 * std::fabs(v)) ++bad;
    total += joint[f];
    hipFree(one);
  }
  printf("joint NLML of %d blocks (N=%d D=%d d=%d) on %d GPU(s): %.10f   last step %.3f ms   mismatches %d\n", F, n, D, d, R, total, ms, bad);
  for (int r = 0; r < R; ++r) {
    hipSetDevice(r);
    ffgp_destroy(h[r]);
    ncclCommDestroy(comm[r]);
  }
  return bad ? 1 : 0;
}
