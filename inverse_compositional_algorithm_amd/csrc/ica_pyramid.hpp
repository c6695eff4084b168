// Host-side construction of the 1-D pyramid operator.
//
// The reference builds its pyramid with skimage.transform.rescale(I, nu, mode='constant', cval=0,
// order=3, anti_aliasing=True, preserve_range=True) (inverse_compositional_algorithm.py:333-336), which
// in scikit-image >= 0.19 is, per spatial axis of length n_in -> n_out (f = n_in / n_out):
//   1. scipy.ndimage.gaussian_filter, sigma = max(0, (f-1)/2), truncate 4, zero padding;
//   2. scipy.ndimage.zoom(order=3, mode='grid-constant', grid_mode=True): pad 12 zeros on both sides,
//      cubic B-spline prefilter (pole sqrt(3)-2, mirror boundary on the padded line), then evaluate the
//      spline at (o + 0.5) * f - 0.5 + 12 for each output index o;
//   3. clip to the input's [min, max] (done in the kernel).
// Steps 1-2 are linear and act on one axis at a time, so the whole thing is one (n_out x n_in) matrix per
// axis.  It is built here in float64 by pushing unit impulses through the same three stages, then stored
// as a band (the prefilter's impulse response decays like 0.268^k).  The device applies the two banded
// operators separably (ica_resample_{v,h}_kernel).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace ica {

struct HostBandOp {
  int n_in = 0, n_out = 0, K = 0;
  std::vector<int> start;   // [n_out]
  std::vector<double> w;    // [n_out][K]
};

inline void bspline3_prefilter_mirror(double* c, int n) {
  // scipy/ndimage/src/ni_splines.c: gain, causal + anticausal recursion with mirror initialisation
  if (n < 2) return;
  const double z = std::sqrt(3.0) - 2.0;
  const double gain = (1.0 - z) * (1.0 - 1.0 / z);
  for (int i = 0; i < n; ++i) c[i] *= gain;
  {
    double z_i = z;
    const double z_n_1 = std::pow(z, (double)(n - 1));
    double c0 = c[0] + z_n_1 * c[n - 1];
    for (int i = 1; i < n - 1; ++i) {
      c0 += z_i * (c[i] + z_n_1 * c[n - 1 - i]);
      z_i *= z;
    }
    c[0] = c0 / (1.0 - z_n_1 * z_n_1);
  }
  for (int i = 1; i < n; ++i) c[i] += z * c[i - 1];
  c[n - 1] = (z * c[n - 2] + c[n - 1]) * z / (z * z - 1.0);
  for (int i = n - 2; i >= 0; --i) c[i] = z * (c[i + 1] - c[i]);
}

inline HostBandOp build_rescale_op(int n_in, int n_out) {
  HostBandOp op;
  op.n_in = n_in;
  op.n_out = n_out;
  const double f = (double)n_in / (double)n_out;
  const double sigma = std::max(0.0, (f - 1.0) / 2.0);
  std::vector<double> g;
  int radius = 0;
  if (sigma > 1e-15) {  // scipy _gaussian_kernel1d, truncate = 4.0
    radius = (int)(4.0 * sigma + 0.5);
    g.resize(2 * radius + 1);
    double sum = 0.0;
    for (int k = -radius; k <= radius; ++k) {
      g[k + radius] = std::exp(-0.5 / (sigma * sigma) * (double)(k * k));
      sum += g[k + radius];
    }
    for (auto& v : g) v /= sum;
  }
  const int npad = 12;
  const int L = n_in + 2 * npad;
  std::vector<double> F((size_t)n_out * n_in, 0.0);
  std::vector<double> c(L);
  // spline evaluation taps / weights per output index (scipy get_spline_interpolation_weights, order 3)
  std::vector<int> st(n_out);
  std::vector<double> wt((size_t)n_out * 4);
  for (int o = 0; o < n_out; ++o) {
    double cc = (double)o;
    cc += 0.5;
    cc *= f;
    cc -= 0.5;
    cc += (double)npad;
    const double fl = std::floor(cc);
    const double y = cc - fl, zc = 1.0 - y;
    st[o] = (int)fl - 1;
    double* w = &wt[(size_t)o * 4];
    w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
    w[2] = (zc * zc * (zc - 2.0) * 3.0 + 4.0) / 6.0;
    w[0] = zc * zc * zc / 6.0;
    w[3] = 1.0 - w[0] - w[1] - w[2];
  }
  for (int j = 0; j < n_in; ++j) {
    std::fill(c.begin(), c.end(), 0.0);
    if (radius > 0) {
      for (int i = std::max(0, j - radius); i <= std::min(n_in - 1, j + radius); ++i) c[npad + i] = g[(j - i) + radius];
    } else {
      c[npad + j] = 1.0;
    }
    bspline3_prefilter_mirror(c.data(), L);
    for (int o = 0; o < n_out; ++o) {
      double v = 0.0;
      for (int m = 0; m < 4; ++m) {
        const int idx = st[o] + m;
        if (idx >= 0 && idx < L) v += wt[(size_t)o * 4 + m] * c[idx];
      }
      F[(size_t)o * n_in + j] = v;
    }
  }
  // band extraction
  // The prefilter's impulse response decays like 0.268^k.  Weights below 2e-8 are dropped: together they move an
  // output by < 3e-5 grey levels on 8-bit-range data, the size of one float32 rounding of the stored pixel.
  const double eps = 2e-8;
  op.start.resize(n_out);
  std::vector<int> last(n_out);
  int K = 1;
  for (int o = 0; o < n_out; ++o) {
    int lo = n_in, hi = -1;
    for (int j = 0; j < n_in; ++j)
      if (std::fabs(F[(size_t)o * n_in + j]) > eps) { lo = std::min(lo, j); hi = std::max(hi, j); }
    if (hi < lo) { lo = 0; hi = 0; }
    op.start[o] = lo;
    last[o] = hi;
  }
  // The device passes take the first output of a group / tile as the one with the smallest start and the last as the
  // one with the largest end, i.e. they need non-decreasing starts.  Thresholding each row on its own does not give
  // that for non-integer factors (a start can step back by one): start every band at the smallest start of the
  // outputs that follow it, and size the common band length for that.
  for (int o = n_out - 2; o >= 0; --o) op.start[o] = std::min(op.start[o], op.start[o + 1]);
  for (int o = 0; o < n_out; ++o) K = std::max(K, last[o] - op.start[o] + 1);
  op.K = K;
  op.w.assign((size_t)n_out * K, 0.0);
  for (int o = 0; o < n_out; ++o)
    for (int j = op.start[o]; j <= last[o]; ++j) op.w[(size_t)o * K + (j - op.start[o])] = F[(size_t)o * n_in + j];
  return op;
}

// zoom.py:8-22: int(np.round(n * factor)), numpy rounds half to even.
inline int round_half_even(double v) {
  const double r = std::nearbyint(v);  // default FE_TONEAREST = half to even
  return (int)r;
}

}  // namespace ica
