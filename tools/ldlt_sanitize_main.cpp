// Driver of tools/ldlt_sanitize.sh: the host-only factorisation under AddressSanitizer / UBSan (stand-ins for the two
// library functions the translation unit needs from context.hip).
#include <cstdio>
#include <cstdarg>
#include <cstdint>
#include <cstdlib>
#include <complex>
#include <map>
#include <random>
#include <thread>
#include <vector>
#include "rlhip.h"
namespace rlh { void set_error(const char *fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fprintf(stderr, "\n"); }
int host_threads() { return 4; } }
template <typename T> T mk(std::mt19937 &g, bool diag);
template <> double mk<double>(std::mt19937 &g, bool) { return std::normal_distribution<double>()(g); }
template <> std::complex<double> mk<std::complex<double>>(std::mt19937 &g, bool diag) {
  std::normal_distribution<double> d; return diag ? std::complex<double>(d(g), 0) : std::complex<double>(d(g), d(g)); }
template <typename T> int run(int dtype, int n, double dens, int zero_diag, unsigned seed) {
  std::mt19937 g(seed);
  std::vector<std::map<int, T>> rows(n);
  std::uniform_real_distribution<double> u(0, 1);
  for (int i = 0; i < n; ++i) {
    if (!zero_diag || (i % zero_diag)) rows[i][i] = mk<T>(g, true) + T(0.3);
    int k = (int)(dens * n) + 1;
    for (int q = 0; q < k; ++q) { int j = (int)(u(g) * n); if (j > i) rows[i][j] = mk<T>(g, false); }
    if (i + 1 < n) rows[i][i + 1] = T(1.0);          // structurally nonsingular enough
  }
  std::vector<int64_t> ip(n + 1, 0); std::vector<int32_t> ix; std::vector<T> v;
  for (int i = 0; i < n; ++i) { for (auto &e : rows[i]) { ix.push_back(e.first); v.push_back(e.second); } ip[i + 1] = (int64_t)ix.size(); }
  rlh_ldlt_t f = nullptr;
  int rc = rlh_ldlt_factor(&f, dtype, n, ip.data(), ix.data(), v.data(), nullptr, 0.01, 1e-13);
  if (rc) { printf("n=%d rc=%d\n", n, rc); return rc; }
  int64_t info[RLH_LDLT_INFO]; rlh_ldlt_info(f, info);
  std::vector<int64_t> lp(n + 1), ord(n); std::vector<int32_t> li(info[0] + 1); std::vector<T> lv(info[0] + 1), d(n), e(n); std::vector<int8_t> b(n);
  rlh_ldlt_get(f, lp.data(), li.data(), lv.data(), d.data(), e.data(), b.data(), ord.data());
  printf("dtype %d n=%d nnzL=%lld neg=%lld pos=%lld 2x2=%lld delayed=%lld maxfront=%lld perturbed=%lld\n", dtype, n, (long long)info[0], (long long)info[1],
         (long long)info[2], (long long)info[4], (long long)info[5], (long long)info[6], (long long)info[3]);
  rlh_ldlt_destroy(f);
  return 0;
}
int main() {
  int bad = 0;
  for (unsigned s = 0; s < 3; ++s) {
    bad |= run<double>(RLH_D, 1, 0.1, 0, s);
    bad |= run<double>(RLH_D, 2, 0.5, 1, s);
    bad |= run<double>(RLH_D, 57, 0.08, 3, s);
    bad |= run<double>(RLH_D, 400, 0.02, 2, s);
    bad |= run<double>(RLH_D, 1500, 0.004, 0, s);
    bad |= run<std::complex<double>>(RLH_Z, 300, 0.03, 2, s);
    bad |= run<std::complex<double>>(RLH_Z, 900, 0.006, 0, s);
  }
  bad |= run<double>(RLH_D, 3000, 0.003, 2, 7);
  printf(bad ? "FAILED\n" : "all ok\n");
  return bad;
}
