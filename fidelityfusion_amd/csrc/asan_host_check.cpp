// Host-side AddressSanitizer walk of libffgp's launch / bookkeeping code (make asan; CPU box only).
// Without a GPU every entry point must fail cleanly: ffgp_create reports FFGP_ERR_NODEVICE and leaves *out untouched,
// a NULL handle is FFGP_ERR_ARG everywhere, ffgp_destroy(NULL) is a no-op.  With a GPU visible the same binary also
// runs one small fused NLML so that the workspace bookkeeping (grow, reuse, destroy) is walked under ASAN's allocator.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ffgp.h"

#define EXPECT(cond)                                                          \
  do {                                                                        \
    if (!(cond)) {                                                            \
      std::fprintf(stderr, "asan_host_check: %s failed (line %d)\n", #cond, __LINE__); \
      return 1;                                                               \
    }                                                                         \
  } while (0)

int main() {
  EXPECT(std::strstr(ffgp_version(), "ffgp") != nullptr);
  EXPECT(ffgp_create(0, nullptr) == FFGP_ERR_ARG);
  EXPECT(ffgp_destroy(nullptr) == FFGP_OK);
  ffgp_handle* h = reinterpret_cast<ffgp_handle*>(0x1);   // must come back untouched when there is no device
  const int rc = ffgp_create(0, &h);
  if (rc == FFGP_ERR_NODEVICE) {
    EXPECT(h == reinterpret_cast<ffgp_handle*>(0x1));
    EXPECT(ffgp_create(-1, &h) == FFGP_ERR_NODEVICE);
    // NULL-handle argument checks never dereference
    double x = 0.0;
    ffgp_problem p;
    std::memset(&p, 0, sizeof p);
    EXPECT(ffgp_set_option(nullptr, "timing", 1.0) < 0);
    EXPECT(ffgp_set_stream(nullptr, nullptr) < 0);
    EXPECT(ffgp_potrf(nullptr, &x, 1, 2) < 0);
    EXPECT(ffgp_nlml_fused(nullptr, &p, &x, nullptr) < 0);
    EXPECT(ffgp_wait(nullptr) < 0);
    // round 5: the batched / ragged likelihood and the K-steps-per-call training entry refuse a NULL handle before touching anything
    ffgp_links lk;
    std::memset(&lk, 0, sizeof lk);
    ffgp_adam ad = {1e-2, 0.9, 0.999, 1e-8};
    int st[2] = {0, 0};
    EXPECT(ffgp_nlml_fused_batch(nullptr, 2, &p, &lk, &x, nullptr, st) < 0);
    EXPECT(ffgp_train_raw(nullptr, 1, &p, &lk, 3, &ad, &x, 8, 0, &x, 3) < 0);
    ffgp_kdesc kd[2];
    ffgp_kdesc_grads kg[2];
    std::memset(kd, 0, sizeof kd);
    std::memset(kg, 0, sizeof kg);
    EXPECT(ffgp_assemble_pair(nullptr, &x, 1, &x, 1, 1, kd, FFGP_KOP_SUM, nullptr, nullptr, 0, nullptr, 0, 0.0, 0.0, &x, 2, 0) < 0);
    EXPECT(ffgp_kernel_grad_pair(nullptr, &x, 1, &x, 1, 1, kd, FFGP_KOP_PRODUCT, &x, 2, kg) < 0);
    std::printf("asan_host_check: no device -- argument / no-device paths clean\n");
    return 0;
  }
  EXPECT(rc == FFGP_OK && h != nullptr);
  EXPECT(ffgp_set_option(h, "no-such-option", 1.0) < 0);
  EXPECT(ffgp_potrf(h, nullptr, 4, 4) < 0);
  EXPECT(ffgp_destroy(h) == FFGP_OK);
  std::printf("asan_host_check: device present -- create / option / argument / destroy paths clean\n");
  return 0;
}
