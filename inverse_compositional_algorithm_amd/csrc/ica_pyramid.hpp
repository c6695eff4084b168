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

// Away from the border a decimation by exactly 2 is shift-invariant: output o is sum_k w[k] in[2 o - 12 + k] with the same
// `fast_k` weights; the float64 recursions that build the operator leave those rows equal to the last bit or two.  Make
// them bit-identical (every pyramid kernel then applies the same weights to the same samples in the same order, and a
// kernel may keep ONE copy of them) and report the range [inv_first, inv_end) and the weights `sw`.  Nothing happens
// (empty range) unless n_in == 2 n_out and the band has `fast_k` taps.
inline void canonicalise_interior(HostBandOp& h, int fast_k, int& inv_first, int& inv_end, std::vector<float>& sw) {
  inv_first = inv_end = 0;
  sw.assign(fast_k, 0.0f);
  if (h.n_in != 2 * h.n_out || h.K != fast_k || h.n_out < 2) return;
  const int mid = h.n_out / 2;
  auto same = [&](int o) {
    if (h.start[o] - h.start[mid] != 2 * (o - mid)) return false;
    for (int k = 0; k < h.K; ++k)
      if (std::fabs((float)h.w[(size_t)o * h.K + k] - (float)h.w[(size_t)mid * h.K + k]) > 1.2e-7f) return false;
    return true;
  };
  int a = mid, b = mid;
  while (a > 0 && same(a - 1)) --a;
  while (b + 1 < h.n_out && same(b + 1)) ++b;
  inv_first = a;
  inv_end = b + 1;
  for (int o = inv_first; o < inv_end; ++o)
    for (int k = 0; k < h.K; ++k) h.w[(size_t)o * h.K + k] = h.w[(size_t)mid * h.K + k];
  for (int k = 0; k < h.K; ++k) sw[k] = (float)h.w[(size_t)mid * h.K + k];
}

// Adjacent-output horizontal pass (ica_kernels.hpp, FusedCols).  The outputs are cut into column blocks of `nb` (even);
// lane l of block b produces outputs b nb + 2 l and b nb + 2 l + 1 from the `pair_win` input columns that start at
// c0[b] + 4 (l + hoff[b]).  Possible iff, for every block, a common 4-aligned origin exists (-3 .. +1 groups of 4
// columns from c0[b]) such that both bands of every pair -- their NONZERO weights: at an image border the stored band is
// longer than that -- lie inside the pair's window.  True for a decimation by exactly 2 (bands of 26 at 2 o - 12),
// borders included.  wpair[((b pair_win + j) 64 + l) 2 + q] = weight of window column j for output q of the pair.
inline bool build_pair_pass(const HostBandOp& h, int nb, const std::vector<int>& c0, int pair_win, std::vector<float>& wpair,
                            std::vector<int>& hoff) {
  const int n_out = h.n_out, nbk = (n_out + nb - 1) / nb;
  if (nb <= 0 || nb % 2 != 0 || nb > 128 || (int)c0.size() != nbk) return false;
  std::vector<int> f0(n_out), f1(n_out);   // first / last column with a nonzero weight
  for (int o = 0; o < n_out; ++o) {
    int a = h.K, b = -1;
    for (int k = 0; k < h.K; ++k)
      if (h.w[(size_t)o * h.K + k] != 0.0) { a = std::min(a, k); b = std::max(b, k); }
    if (b < 0) { a = 0; b = 0; }
    f0[o] = h.start[o] + a;
    f1[o] = h.start[o] + b;
  }
  wpair.assign((size_t)nbk * pair_win * 64 * 2, 0.0f);
  hoff.assign(nbk, 0);
  for (int bx = 0; bx < nbk; ++bx) {
    const int ob = bx * nb, oe = std::min(ob + nb, n_out);
    int org = f0[ob];
    for (int o = ob; o < oe; ++o) org = std::min(org, f0[o] - 4 * ((o - ob) / 2));
    org = (int)std::floor(org / 4.0) * 4;
    const int off = (org - c0[bx]) / 4;
    if ((org - c0[bx]) % 4 != 0 || off < -3 || off > 1) return false;
    hoff[bx] = off;
    for (int o = ob; o < oe; ++o) {
      const int l = (o - ob) / 2, q = (o - ob) & 1, w0 = org + 4 * l;
      if (f0[o] < w0 || f1[o] > w0 + pair_win - 1) return false;
      for (int k = 0; k < h.K; ++k) {
        const int j = h.start[o] + k - w0;
        if (j >= 0 && j < pair_win) wpair[(((size_t)bx * pair_win + j) * 64 + l) * 2 + q] = (float)h.w[(size_t)o * h.K + k];
      }
    }
  }
  return true;
}

// zoom.py:8-22: int(np.round(n * factor)), numpy rounds half to even.
inline int round_half_even(double v) {
  const double r = std::nearbyint(v);  // default FE_TONEAREST = half to even
  return (int)r;
}

}  // namespace ica
