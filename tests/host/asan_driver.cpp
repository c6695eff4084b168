// Host-side exercise of the C-ABI library under AddressSanitizer + UBSan ("a -fsanitize=address
// host build of the C-ABI shim").  Built by tests/test_host_asan.py together with csrc/ica_api.hip:
//   hipcc --offload-arch=gfx950 -Xarch_host -fsanitize=address,undefined ... asan_driver.cpp ica_api.hip
// Only HOST code is sanitised and only host code runs here: argument validation of every entry point, the status
// strings, the no-device path, and the host-only algebra the library is built from (pyramid operator construction,
// LU inverse, compositional update, parameter zoom).  No kernel is launched; it runs on a machine without a GPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ica_hip.h"
#include "../../inverse_compositional_algorithm_amd/csrc/ica_math.hpp"
#include "../../inverse_compositional_algorithm_amd/csrc/ica_pyramid.hpp"

// The fused iteration kernel lives in translation units of its own (csrc/ica_iterate_inst.hip, one per transform type);
// nothing here launches a kernel, so the driver links against stand-ins instead of compiling them five more times.
#include "../../inverse_compositional_algorithm_amd/csrc/ica_iterate_launch.hpp"
namespace ica {
#define ICA_STUB(n) \
  bool launch_iterate_tt##n(int, int, int, unsigned, hipStream_t, const LevelArgs&) { std::abort(); }
ICA_STUB(1) ICA_STUB(2) ICA_STUB(3) ICA_STUB(4) ICA_STUB(5)
#undef ICA_STUB
}  // namespace ica

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static double rnd(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) & 0xFFFFFF) / double(0x1000000) - 0.5;
}

int main() {
  using namespace ica;
  // ---- C ABI: host-only entry points and argument validation ----
  CHECK(ica_version() == ICA_ABI_VERSION);
  const int want_n[] = {-1, 2, 3, 4, 6, 8, -1};
  for (int t = 0; t <= 6; ++t) CHECK(ica_nparams(t) == want_n[t]);
  int w = 0, h = 0;
  CHECK(ica_zoom_size(135, 1080, 0.5, &w, &h) == ICA_OK && w == 68 && h == 540);
  CHECK(ica_zoom_size(5, 3, 0.5, &w, &h) == ICA_OK && w == 2 && h == 2);
  CHECK(ica_zoom_size(5, 3, 0.5, nullptr, &h) == ICA_EINVAL_ARG);
  for (int s = 1; s >= -12; --s) CHECK(std::strlen(ica_strerror(s)) > 0);
  CHECK(std::strcmp(ica_last_error(nullptr), "") == 0);
  CHECK(ica_create(0, nullptr) == ICA_EINVAL_ARG);
  ica_ctx* ctx = nullptr;
  const int have = ica_device_count();
  const int rc = ica_create(have > 0 ? 0 : 0, &ctx);
  if (have <= 0) {
    CHECK(rc == ICA_ENODEVICE && ctx == nullptr);
    CHECK(ica_create(-1, &ctx) == ICA_ENODEVICE);
  } else {
    CHECK(rc == ICA_OK && ctx != nullptr);
  }
  ica_problem pr;
  std::memset(&pr, 0, sizeof(pr));
  float img = 0.f;
  double p8[8] = {0}, err = 0;
  int32_t nit[4] = {0};
  CHECK(ica_destroy(nullptr) == ICA_EINVAL_ARG);
  CHECK(ica_set_stream(nullptr, nullptr) == ICA_EINVAL_ARG);
  CHECK(ica_synchronize(nullptr) == ICA_EINVAL_ARG);
  CHECK(ica_set_profiling(nullptr, 1) == ICA_EINVAL_ARG);
  CHECK(ica_prepare(nullptr, &pr) == ICA_EINVAL_ARG);
  CHECK(ica_solve_batch_dev(nullptr, &pr, &img, &img, p8, &err, nit, nullptr, nullptr, nullptr, nullptr) == ICA_EINVAL_ARG);
  CHECK(ica_solve_batch(nullptr, &pr, &img, &img, p8, &err, nit, nullptr, nullptr, nullptr, nullptr) == ICA_EINVAL_ARG);
  CHECK(ica_op_minmax_dev(nullptr, 1, 1, 1, 1, &img, &img) == ICA_EINVAL_ARG);
  { double ghz = 0; CHECK(ica_op_shader_clock_dev(nullptr, 100, &ghz) == ICA_EINVAL_ARG); }
  CHECK(ica_op_warp_dev(nullptr, 1, 1, 8, 8, 1, &img, p8, nullptr, nullptr, &img, nullptr) == ICA_EINVAL_ARG);
  CHECK(ica_op_normal_equations_dev(nullptr, &pr, &img, &img, p8, nullptr, p8, p8) == ICA_EINVAL_ARG);
  CHECK(ica_op_solve_update_dev(nullptr, 1, 1, 0, p8, p8, p8, nullptr, nullptr) == ICA_EINVAL_ARG);
  CHECK(ica_op_pyramid_down_dev(nullptr, 1, 1, 8, 8, 0.5, &img, &img) == ICA_EINVAL_ARG);
  CHECK(ica_op_zoom_in_parameters_dev(nullptr, 1, 1, 4, 4, 8, 8, p8) == ICA_EINVAL_ARG);
  CHECK(ica_op_transform_image_dev(nullptr, 1, 1, 8, 8, &img, p8, &img) == ICA_EINVAL_ARG);
  CHECK(ica_op_ref_jacobian_dev(nullptr, 1, 2, 2, p8) == ICA_EINVAL_ARG);
  CHECK(ica_op_ref_steepest_descent_dev(nullptr, 2, 2, 1, 2, p8, p8, p8, p8) == ICA_EINVAL_ARG);
  CHECK(ica_op_ref_normal_dev(nullptr, 2, 2, 1, 2, p8, p8, nullptr, p8, p8) == ICA_EINVAL_ARG);
  CHECK(ica_op_ref_rhop_dev(nullptr, 4, p8, 1.0, 3, p8) == ICA_EINVAL_ARG);
  CHECK(ica_op_ref_robust_error_dev(nullptr, 2, 2, 1, p8, 1.0, 3, p8) == ICA_EINVAL_ARG);
  if (ctx) {
    // with a device: the validation order of the problem descriptor (ref :48-60) -- still no kernel launch
    pr.batch = 1; pr.height = 32; pr.width = 32; pr.channels = 7; pr.tol = 1e-3; pr.transform_type = 4; pr.nscales = 1;
    pr.max_iter = 30;
    CHECK(ica_prepare(ctx, &pr) == ICA_EINVAL_SHAPE);
    pr.channels = 3; pr.tol = 0.01;
    CHECK(ica_prepare(ctx, &pr) == ICA_EINVAL_TOL);
    pr.tol = 1e-3; pr.transform_type = 9;
    CHECK(ica_prepare(ctx, &pr) == ICA_EINVAL_TRANSFORM);
    pr.transform_type = 4; pr.robust_type = 7;
    CHECK(ica_prepare(ctx, &pr) == ICA_EINVAL_ROBUST);
    CHECK(ica_destroy(ctx) == ICA_OK);
  }

  // ---- pyramid operator construction (ica_pyramid.hpp) ----
  const int sizes[][2] = {{512, 256}, {135, 68}, {101, 61}, {300, 30}, {64, 48}, {7, 3}, {2, 1}, {1, 1}, {33, 33}};
  for (auto& sz : sizes) {
    HostBandOp op = build_rescale_op(sz[0], sz[1]);
    CHECK(op.n_in == sz[0] && op.n_out == sz[1] && op.K >= 1);
    CHECK((int)op.start.size() == sz[1] && op.w.size() == (size_t)sz[1] * op.K);
    for (int o = 0; o < sz[1]; ++o) {
      CHECK(op.start[o] >= 0 && op.start[o] < sz[0]);
      if (o) CHECK(op.start[o] >= op.start[o - 1]);
      double sum = 0;
      for (int k = 0; k < op.K; ++k) {
        if (op.start[o] + k >= sz[0]) CHECK(op.w[(size_t)o * op.K + k] == 0.0);  // nothing reaches past the input
        sum += op.w[(size_t)o * op.K + k];
      }
      // away from the zero-padded borders the operator preserves constants
      if (sz[0] >= 64 && o > sz[1] / 4 && o < 3 * sz[1] / 4) CHECK(std::fabs(sum - 1.0) < 1e-6);
    }
  }
  CHECK(round_half_even(2.5) == 2 && round_half_even(3.5) == 4 && round_half_even(67.5) == 68);

  // ---- decimation by 2: bit-identical interior weights, and the adjacent-output pass's per-lane window weights ----
  const int sizes2[][2] = {{1024, 512}, {512, 256}, {270, 135}, {128, 64}, {96, 48}, {135, 68}, {128, 96}, {300, 100}};
  for (auto& sz : sizes2) {
    HostBandOp op = build_rescale_op(sz[0], sz[1]);
    const HostBandOp before = op;
    int f = 0, e = 0;
    std::vector<float> sw;
    canonicalise_interior(op, 26, f, e, sw);
    if (sz[0] == 2 * sz[1]) {
      CHECK(op.K == 26 && f <= 6 && e >= sz[1] - 6 && e > f);
      for (int o = f; o < e; ++o) {
        CHECK(op.start[o] == 2 * o - 12);
        for (int k = 0; k < op.K; ++k) {
          CHECK(op.w[(size_t)o * op.K + k] == op.w[(size_t)f * op.K + k] && (float)op.w[(size_t)o * op.K + k] == sw[k]);
          CHECK(std::fabs(op.w[(size_t)o * op.K + k] - before.w[(size_t)o * op.K + k]) <= 1.3e-7);   // only the last bits moved
        }
      }
    } else {
      CHECK(f == e);
      CHECK(op.w == before.w);
    }
    // column blocks as get_op() cuts them: the largest multiple of 8 outputs whose bands fit 256 columns from a 4-aligned start
    const int n_out = sz[1];
    auto fits = [&](int nb) {
      for (int ob = 0; ob < n_out; ob += nb)
        if (op.start[std::min(ob + nb, n_out) - 1] + op.K - (op.start[ob] & ~3) > 256) return false;
      return true;
    };
    int nb = 128;
    while (nb >= 16 && !fits(nb)) nb -= 8;
    if (nb < 16) continue;
    std::vector<int> c0;
    for (int ob = 0; ob < n_out; ob += nb) c0.push_back(op.start[ob] & ~3);
    std::vector<float> wp;
    std::vector<int> off;
    const bool ok = build_pair_pass(op, nb, c0, 28, wp, off);
    if (sz[0] == 2 * sz[1]) CHECK(ok);
    if (!ok) continue;
    CHECK(wp.size() == c0.size() * 28 * 64 * 2 && off.size() == c0.size());
    for (int o = 0; o < n_out; ++o) {   // the window weights of an output are its band, nothing else
      const int bx = o / nb, l = (o - bx * nb) / 2, q = (o - bx * nb) & 1, w0 = c0[bx] + 4 * (off[bx] + l);
      CHECK(off[bx] >= -3 && off[bx] <= 1);
      double sum = 0;
      for (int j = 0; j < 28; ++j) {
        const float t = wp[(((size_t)bx * 28 + j) * 64 + l) * 2 + q];
        const int k = w0 + j - op.start[o];
        const float want = (k >= 0 && k < op.K) ? (float)op.w[(size_t)o * op.K + k] : 0.0f;
        CHECK(t == want);
        sum += t;
      }
      double full = 0;
      for (int k = 0; k < op.K; ++k) full += (float)op.w[(size_t)o * op.K + k];
      CHECK(std::fabs(sum - full) < 1e-12);   // no band weight fell outside the window
    }
    // lanes past a block's last output carry no weights
    for (size_t bx = 0; bx < c0.size(); ++bx) {
      const int outs = std::min(nb, n_out - (int)bx * nb);
      for (int l = (outs + 1) / 2; l < 64; ++l)
        for (int j = 0; j < 28; ++j) CHECK(wp[((bx * 28 + j) * 64 + l) * 2] == 0.0f && wp[((bx * 28 + j) * 64 + l) * 2 + 1] == 0.0f);
    }
  }

  // ---- LU inverse (np.linalg.inv semantics) ----
  unsigned seed = 12345u;
  for (int n : {2, 3, 4, 6, 8}) {
    double A[64], Hm[64], Hi[64], ws[kLuWork];
    for (int i = 0; i < n * n; ++i) A[i] = rnd(seed);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double v = (i == j) ? 0.5 : 0.0;
        for (int k = 0; k < n; ++k) v += A[i * n + k] * A[j * n + k];
        Hm[i * 8 + j] = v;
      }
    CHECK(lu_inverse(n, Hm, Hi, 8, ws));
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        double v = 0;
        for (int k = 0; k < n; ++k) v += Hm[i * 8 + k] * Hi[k * 8 + j];
        CHECK(std::fabs(v - (i == j ? 1.0 : 0.0)) < 1e-9);
      }
    for (int i = 0; i < 64; ++i) Hm[i] = 0.0;  // singular -> zeros, like the reference's LinAlgError branch
    CHECK(!lu_inverse(n, Hm, Hi, 8, ws));
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) CHECK(Hi[i * 8 + j] == 0.0);
  }

  // ---- compositional update: p o dp^-1 with dp = 0 is the identity; composing with dp then with -ish dp returns ----
  for (int tt = ICA_TRANSLATION; tt <= ICA_HOMOGRAPHY; ++tt) {
    const int n = nparams_of(tt);
    double p[8] = {0}, q[8] = {0}, dp[8] = {0}, a[8];
    for (int i = 0; i < n; ++i) { p[i] = 0.01 * rnd(seed); q[i] = p[i]; }
    update_transform(q, dp, tt, false);
    for (int i = 0; i < n; ++i) CHECK(std::fabs(q[i] - p[i]) < 1e-12);
    for (int i = 0; i < n; ++i) dp[i] = 0.001 * rnd(seed);
    update_transform(q, dp, tt, true);
    params_to_displacement(q, tt, a);
    for (int i = 0; i < 8; ++i) CHECK(std::isfinite(a[i]));
    double z[8];
    for (int i = 0; i < 8; ++i) z[i] = q[i];
    zoom_in_parameters(z, tt, 68, 120, 135, 240);
    CHECK(std::isfinite(z[0]) && (tt == ICA_HOMOGRAPHY ? z[6] * 2.0 == q[6] || q[6] == 0.0 || std::fabs(z[6] * 2.0 - q[6]) < 1e-15
                                                      : z[0] == q[0] * 2.0));
  }
  for (int k = 0; k < 8; ++k)
    for (int m = k; m < 8; ++m) CHECK(tri_index(k, m, 8) >= 0 && tri_index(k, m, 8) < kTri);
  std::printf("ASAN_DRIVER_OK devices=%d\n", have);
  return 0;
}
