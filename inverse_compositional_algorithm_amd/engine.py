"""Batched device-side engine: thin Python host code over the C ABI (include/ica_hip.h).

PyTorch is used for device memory and streams only; every computation happens in libica_hip.so.
Images are planar ``[B, C, H, W]`` CUDA(HIP) tensors, float32 or uint8 (8-bit images stay one byte per pixel in HBM;
the two storages give bit-identical results on the same pixel values); parameters are float64 ``[B, 8]``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import Problem, Stats, check
from .constants import LAMBDA_0, LAMBDA_N, LAMBDA_RATIO, MAX_ITER


@dataclass
class SolveResult:
    p: torch.Tensor                 # [B, 8] float64 (device)
    error: torch.Tensor             # [B] float64: |dp| of the last scale-0 iteration
    niter: torch.Tensor             # [B, nscales] int32, index 0 = finest scale
    DI: Optional[torch.Tensor] = None     # [B, C, H, W] float32 with NaN
    Iw: Optional[torch.Tensor] = None
    trace: Optional[torch.Tensor] = None  # [B, nscales, max_iter, 10] float64
    stats: dict = field(default_factory=dict)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class IcaEngine:
    """One engine (= one ``ica_ctx``) per GPU."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("IcaEngine needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
        self.device = torch.device("cuda", device)
        self.index = device
        ctx = C.c_void_p()
        check(self.lib.ica_create(device, C.byref(ctx)))
        self.ctx = ctx
        self._profiling = False

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.ica_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------------------
    def _bind_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.ica_set_stream(self.ctx, C.c_void_p(s)), self.ctx)

    def set_profiling(self, on: bool):
        self._profiling = bool(on)
        check(self.lib.ica_set_profiling(self.ctx, int(on)), self.ctx)

    def synchronize(self):
        check(self.lib.ica_synchronize(self.ctx), self.ctx)

    def shader_clock_ghz(self, duration_us: int = 20000) -> float:
        """The shader clock one wave sees over `duration_us` (s_memtime against the constant 100 MHz counter), on this
        engine's current stream: call it on a second engine / stream while another one computes to read the clock the
        chip sustains under that load."""
        self._bind_stream()
        ghz = C.c_double(0.0)
        check(self.lib.ica_op_shader_clock_dev(self.ctx, int(duration_us), C.byref(ghz)), self.ctx)
        return float(ghz.value)

    @staticmethod
    def problem(batch, height, width, channels, transform_type, robust_type=0, nscales=1, nu=0.5, tol=1e-3,
                lambda_=0.0, nanifoutside=True, delta=10, max_iter=MAX_ITER, flags=0, storage=0) -> Problem:
        """`storage`: _lib.ICA_STORAGE_F32 / ICA_STORAGE_U8; solve() and normal_equations() set it from the tensors' dtype."""
        return Problem(batch=batch, height=height, width=width, channels=channels, transform_type=int(transform_type),
                       robust_type=int(robust_type), nscales=nscales, max_iter=max_iter,
                       nanifoutside=int(bool(nanifoutside)), delta=int(delta), flags=flags, storage=int(storage), tol=tol,
                       lambda_=lambda_, nu=nu, lambda0=LAMBDA_0, lambdaN=LAMBDA_N, lambda_ratio=LAMBDA_RATIO)

    def _images(self, *ts, allow_u8=False):
        """Returns the storage code of the images (all of one dtype)."""
        dts = set()
        for t in ts:
            if t is None:
                continue
            ok = t.dtype == torch.float32 or (allow_u8 and t.dtype == torch.uint8)
            if not ok or t.dim() != 4 or not t.is_contiguous() or t.device != self.device:
                raise ValueError("images must be contiguous float32%s [B, C, H, W] tensors on the engine's device"
                                 % (" or uint8" if allow_u8 else ""))
            dts.add(t.dtype)
        if len(dts) > 1:
            raise ValueError("I1 and I2 must have the same dtype")
        return _lib.ICA_STORAGE_U8 if torch.uint8 in dts else _lib.ICA_STORAGE_F32

    def prepare(self, prob: Problem):
        check(self.lib.ica_prepare(self.ctx, C.byref(prob)), self.ctx)

    # -- the hot path ----------------------------------------------------------------------------
    def solve(self, I1: torch.Tensor, I2: torch.Tensor, prob: Problem, p0: Optional[torch.Tensor] = None,
              want_images: bool = False, want_trace: bool = False, want_stats: bool = False) -> SolveResult:
        prob.storage = self._images(I1, I2, allow_u8=True)
        if I1.shape != I2.shape:
            check(-2)
        B, Cc, H, W = I1.shape
        if (B, Cc, H, W) != (prob.batch, prob.channels, prob.height, prob.width):
            raise ValueError("problem descriptor does not match the image tensors")
        dev = self.device
        p = torch.zeros(B, 8, dtype=torch.float64, device=dev)
        if p0 is not None:
            p0 = torch.as_tensor(p0, dtype=torch.float64, device=dev).reshape(B, -1)
            p[:, :p0.shape[1]] = p0
        err = torch.empty(B, dtype=torch.float64, device=dev)
        nit = torch.empty(B, prob.nscales, dtype=torch.int32, device=dev)
        DI = torch.empty(I1.shape, dtype=torch.float32, device=dev) if want_images else None
        Iw = torch.empty(I1.shape, dtype=torch.float32, device=dev) if want_images else None
        tr = (torch.empty(B, prob.nscales, prob.max_iter, _lib.ICA_TRACE_STRIDE, dtype=torch.float64, device=dev)
              if want_trace else None)
        st = Stats() if want_stats else None
        self._bind_stream()
        check(self.lib.ica_solve_batch_dev(self.ctx, C.byref(prob), _ptr(I1), _ptr(I2), _ptr(p), _ptr(err), _ptr(nit),
                                           _ptr(DI), _ptr(Iw), _ptr(tr), C.byref(st) if st is not None else None),
              self.ctx)
        res = SolveResult(p=p, error=err, niter=nit, DI=DI, Iw=Iw, trace=tr)
        if st is not None:
            ns = prob.nscales
            res.stats = dict(pair_iterations=list(st.pair_iterations[:ns]),
                             pair_iterations_total=int(st.pair_iterations_total),
                             iter_launches=list(st.iter_launches[:ns]),
                             iter_kernel_ms=list(st.iter_kernel_ms[:ns]),
                             iter_kernel_bytes=list(st.iter_kernel_bytes[:ns]),
                             first_launches=list(st.first_launches[:ns]),
                             first_kernel_ms=list(st.first_kernel_ms[:ns]),
                             first_kernel_bytes=list(st.first_kernel_bytes[:ns]),
                             shader_clock_ghz=float(st.shader_clock_ghz),
                             level_size=[(st.level_width[s], st.level_height[s]) for s in range(ns)])
        return res

    # -- per-operator hooks ----------------------------------------------------------------------
    def minmax(self, I: torch.Tensor) -> torch.Tensor:
        self._images(I)
        B, Cc, H, W = I.shape
        out = torch.empty(B, 2, dtype=torch.float32, device=self.device)
        self._bind_stream()
        check(self.lib.ica_op_minmax_dev(self.ctx, B, Cc, H, W, _ptr(I), _ptr(out)), self.ctx)
        return out

    def warp(self, I2: torch.Tensor, p: torch.Tensor, transform_type: int, I1: Optional[torch.Tensor] = None,
             minmax: Optional[torch.Tensor] = None):
        """Returns Iw (and DI = Iw - I1 when I1 is given)."""
        self._images(I2, I1)
        B, Cc, H, W = I2.shape
        p = self._params(p, B)
        Iw = torch.empty_like(I2)
        DI = torch.empty_like(I2) if I1 is not None else None
        self._bind_stream()
        check(self.lib.ica_op_warp_dev(self.ctx, B, Cc, H, W, int(transform_type), _ptr(I2), _ptr(p), _ptr(minmax),
                                       _ptr(I1), _ptr(Iw), _ptr(DI)), self.ctx)
        return (Iw, DI) if I1 is not None else Iw

    def normal_equations(self, I1, I2, p, prob: Problem, lambda_it=None):
        """H [B,8,8], b [B,8] of one iteration at parameters p."""
        prob.storage = self._images(I1, I2, allow_u8=True)
        B = I1.shape[0]
        p = self._params(p, B)
        lam = None
        if lambda_it is not None:
            lam = torch.as_tensor(lambda_it, dtype=torch.float64, device=self.device).reshape(-1).expand(B).contiguous()
        Hm = torch.empty(B, 8, 8, dtype=torch.float64, device=self.device)
        bv = torch.empty(B, 8, dtype=torch.float64, device=self.device)
        self._bind_stream()
        check(self.lib.ica_op_normal_equations_dev(self.ctx, C.byref(prob), _ptr(I1), _ptr(I2), _ptr(p), _ptr(lam),
                                                   _ptr(Hm), _ptr(bv)), self.ctx)
        return Hm, bv

    def solve_update(self, Hm, bv, p, transform_type: int, flags: int = 0):
        """dp = inv(H) b, error = |dp|, p <- p o dp^-1.  Returns (p_new, dp, error)."""
        B = Hm.shape[0]
        Hm = torch.as_tensor(Hm, dtype=torch.float64, device=self.device).contiguous()
        bv = torch.as_tensor(bv, dtype=torch.float64, device=self.device).contiguous()
        p = self._params(p, B).clone()
        dp = torch.empty(B, 8, dtype=torch.float64, device=self.device)
        err = torch.empty(B, dtype=torch.float64, device=self.device)
        self._bind_stream()
        check(self.lib.ica_op_solve_update_dev(self.ctx, B, int(transform_type), flags, _ptr(Hm), _ptr(bv), _ptr(p),
                                               _ptr(dp), _ptr(err)), self.ctx)
        return p, dp, err

    def pyramid_down(self, I: torch.Tensor, nu: float) -> torch.Tensor:
        self._images(I)
        B, Cc, H, W = I.shape
        w, h = C.c_int(), C.c_int()
        check(self.lib.ica_zoom_size(W, H, nu, C.byref(w), C.byref(h)))
        out = torch.empty(B, Cc, max(h.value, 1), max(w.value, 1), dtype=torch.float32, device=self.device)
        self._bind_stream()
        check(self.lib.ica_op_pyramid_down_dev(self.ctx, B, Cc, H, W, float(nu), _ptr(I), _ptr(out)), self.ctx)
        return out

    # ---- the reference's per-operator array functions (float64, channel-last; include/ica_hip.h "ica_op_ref_*") ----
    def _f64(self, a) -> torch.Tensor:
        import numpy as np
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).to(self.device)

    def ref_jacobian(self, transform_type: int, nx: int, ny: int):
        n = self.lib.ica_nparams(int(transform_type))
        if n < 0:
            raise ValueError("Unknown transform type")
        J = torch.zeros(int(ny), int(nx), 2 * n, dtype=torch.float64, device=self.device)
        self._bind_stream()
        check(self.lib.ica_op_ref_jacobian_dev(self.ctx, int(transform_type), int(nx), int(ny), _ptr(J)), self.ctx)
        return J.cpu().numpy()

    @staticmethod
    def _want_shape(name: str, a, shape) -> None:
        """The reference's NumPy code rejects mismatched operands with a broadcast / index error; raw device pointers
        would read out of bounds instead, so the shapes are checked here."""
        import numpy as np
        if tuple(np.shape(a)) != tuple(shape):
            raise ValueError(f"{name} has shape {tuple(np.shape(a))}, expected {tuple(shape)}")

    def ref_steepest_descent(self, Ix, Iy, J, nparams: int):
        import numpy as np
        if np.ndim(Ix) != 3:
            raise ValueError("Ix and Iy must have shape (ny, nx, nz)")
        ny, nx, nz = Ix.shape
        self._want_shape("Iy", Iy, (ny, nx, nz))
        self._want_shape("J", J, (ny, nx, 2 * int(nparams)))
        DIJ = torch.zeros(ny, nx, nz, int(nparams), dtype=torch.float64, device=self.device)
        a, b, c = self._f64(Ix), self._f64(Iy), self._f64(J)
        self._bind_stream()
        check(self.lib.ica_op_ref_steepest_descent_dev(self.ctx, ny, nx, nz, int(nparams), _ptr(a), _ptr(b), _ptr(c), _ptr(DIJ)),
              self.ctx)
        return DIJ.cpu().numpy()

    def ref_normal(self, DIJ, DI=None, rho=None, want_h=True, want_b=False):
        import numpy as np
        if np.ndim(DIJ) != 4:
            raise ValueError("DIJ must have shape (ny, nx, nz, nparams)")
        ny, nx, nz, n = DIJ.shape
        if n > 8:
            raise ValueError("at most 8 parameters")
        if DI is not None:
            self._want_shape("DI", DI, (ny, nx, nz))
        if rho is not None:
            self._want_shape("rho", rho, (ny, nx))
        d = self._f64(DIJ)
        di = self._f64(DI) if DI is not None else None
        r = self._f64(rho) if rho is not None else None
        H = torch.zeros(n, n, dtype=torch.float64, device=self.device) if want_h else None
        b = torch.zeros(n, dtype=torch.float64, device=self.device) if want_b else None
        if n > 0:
            self._bind_stream()
            check(self.lib.ica_op_ref_normal_dev(self.ctx, ny, nx, nz, n, _ptr(d), _ptr(di) if di is not None else None,
                                                 _ptr(r) if r is not None else None, _ptr(H) if want_h else None,
                                                 _ptr(b) if want_b else None), self.ctx)
        return (H.cpu().numpy() if want_h else None), (b.cpu().numpy() if want_b else None)

    def ref_rhop(self, t2, lambda_: float, robust_type: int):
        import numpy as np
        t = self._f64(np.atleast_1d(t2))
        out = torch.empty_like(t)
        self._bind_stream()
        check(self.lib.ica_op_ref_rhop_dev(self.ctx, t.numel(), _ptr(t), float(lambda_), int(robust_type), _ptr(out)), self.ctx)
        return out.cpu().numpy().reshape(np.shape(t2))

    def ref_robust_error(self, DI, lambda_: float, robust_type: int):
        import numpy as np
        if np.ndim(DI) != 3:
            raise ValueError("DI must have shape (ny, nx, nz)")
        ny, nx, nz = DI.shape
        d = self._f64(DI)
        rho = torch.zeros(ny, nx, dtype=torch.float64, device=self.device)
        self._bind_stream()
        check(self.lib.ica_op_ref_robust_error_dev(self.ctx, ny, nx, nz, _ptr(d), float(lambda_), int(robust_type), _ptr(rho)),
              self.ctx)
        return rho.cpu().numpy()

    def zoom_in_parameters(self, p, transform_type: int, w: int, h: int, ww: int, hh: int):
        B = p.shape[0]
        p = self._params(p, B).clone()
        self._bind_stream()
        check(self.lib.ica_op_zoom_in_parameters_dev(self.ctx, B, int(transform_type), w, h, ww, hh, _ptr(p)), self.ctx)
        return p

    def transform_image(self, I: torch.Tensor, M) -> torch.Tensor:
        """Bilinear resampling out(x) = I(M x) (cval 0, clipped), M [B,3,3] float64 (transformation.py:316)."""
        self._images(I)
        B, Cc, H, W = I.shape
        M = torch.as_tensor(M, dtype=torch.float64, device=self.device).reshape(-1, 9)
        M = (M / M[:, 8:9]).expand(B, 9).contiguous()
        out = torch.empty_like(I)
        self._bind_stream()
        check(self.lib.ica_op_transform_image_dev(self.ctx, B, Cc, H, W, _ptr(I), _ptr(M), _ptr(out)), self.ctx)
        return out

    def _params(self, p, B):
        p = torch.as_tensor(p, dtype=torch.float64, device=self.device)
        if p.dim() == 1:
            p = p.reshape(1, -1).expand(B, -1)
        out = torch.zeros(B, 8, dtype=torch.float64, device=self.device)
        out[:, :p.shape[1]] = p
        return out

    # -- synthetic workloads on the device (same distribution as synthetic.make_pair) --------------
    def make_batch(self, seed: int, batch: int, height: int, width: int, channels: int, transform_type: int,
                   robust_clutter: bool = False, chunk: int = 64, dtype=torch.float32):
        """Bulk synthetic warped pairs generated on the GPU: multi-octave noise texture I2, random small
        ground-truth transform, I1 = round_u8(bicubic I2(M(p_gt) x)) through this library's own warp
        operator.  Returns (I1, I2, p_gt[B, 8]); pixel values are integers 0..255 either way, `dtype`
        (torch.float32 / torch.uint8) is how they are stored."""
        from . import synthetic as syn
        dev = self.device
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        rng = np.random.default_rng(seed)
        I1 = torch.empty(batch, channels, height, width, dtype=dtype, device=dev)
        I2 = torch.empty_like(I1)
        n = syn.NPARAMS[int(transform_type)]
        pg = np.zeros((batch, 8))
        for b in range(batch):
            pg[b, :n] = syn.ground_truth(rng, int(transform_type), height, width)
        p_gt = torch.from_numpy(pg).to(dev)
        for b0 in range(0, batch, chunk):
            nb = min(chunk, batch - b0)
            acc = torch.zeros(nb * channels, 1, height, width, device=dev)
            for k, s in enumerate(syn.OCTAVE_SIGMAS):
                noise = torch.randn(nb * channels, 1, height, width, device=dev, generator=g)
                r = int(4 * s + 0.5)
                xs = torch.arange(-r, r + 1, device=dev, dtype=torch.float32)
                kern = torch.exp(-0.5 * (xs / s) ** 2)
                kern = kern / kern.sum()
                pad = torch.nn.functional.pad(noise, (r, r, r, r), mode="reflect") if r < min(height, width) else \
                    torch.nn.functional.pad(noise, (r, r, r, r), mode="circular")
                t = torch.nn.functional.conv2d(pad, kern.view(1, 1, 1, -1))
                t = torch.nn.functional.conv2d(t, kern.view(1, 1, -1, 1))
                t = t.view(nb, channels, height, width)
                acc += (t / t.std(dim=(1, 2, 3), keepdim=True) * (1.5 ** k)).view(nb * channels, 1, height, width)
            acc = acc.view(nb, channels, height, width)
            lo = acc.amin(dim=(1, 2, 3), keepdim=True)
            hi = acc.amax(dim=(1, 2, 3), keepdim=True)
            i2 = torch.round((acc - lo) / (hi - lo) * 255.0)
            I2[b0:b0 + nb] = i2.to(dtype)
            iw = self.warp(i2.contiguous(), p_gt[b0:b0 + nb], transform_type)
            iw = torch.nan_to_num(iw, nan=0.0)
            if robust_clutter:
                iw = iw + 5.0 * torch.randn(iw.shape, device=dev, generator=g)
                oh, ow = int(round(height * 0.316)), int(round(width * 0.316))
                for i in range(nb):
                    oy = int(rng.integers(0, height - oh))
                    ox = int(rng.integers(0, width - ow))
                    iw[i, :, oy:oy + oh, ox:ox + ow] = torch.randint(0, 256, (channels, oh, ow), device=dev,
                                                                     generator=g).float()
            I1[b0:b0 + nb] = torch.clamp(torch.round(iw), 0, 255).to(dtype)
        return I1, I2, p_gt
