"""Host-side mirror of the reference's call seams for the ConvONet-Opt path.

The reference exposes the path through module-level globals (``generator.model.encode_inputs`` /
``.decode`` / ``repulsion_loss`` / ``optimize_points``, ConvONet/opt_defense.py:212,221,300,182).
``Restorer`` offers the same calls with the same argument meaning, backed by libifd.so's HIP kernels;
PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import IfdConfig, IfdMeshParams, IfdOptParams, IfdPrepParams

PLANE_ORDER = ("xz", "xy", "yz")
# measurement hook: IFD_SPLIT=1|2|4 overrides the automatic choice of ifd_opt_params.split (results do not depend on it)
import os as _os
_ENV_SPLIT = int(_os.environ.get("IFD_SPLIT", "0"))
# ifd_opt_params.precision by name.  Where a caller passes None the module default applies: "f32" unless a test / measurement
# set another one (set_default_precision; tests/conftest.py runs the parity matrix in both modes that way).  The environment's
# IFD_PRECISION is honoured only together with IFD_ENABLE_TEST_HOOKS=1 (round-5 advisor: a stray variable must not switch a
# production process to another arithmetic); the CLIs and pipeline.py always pass an explicit precision.
PRECISIONS = {"f32": 0, "bf16x6": 1, "bf16x3": 2}
_DEFAULT_PRECISION = "f32"
if _os.environ.get("IFD_ENABLE_TEST_HOOKS", "") == "1" and _os.environ.get("IFD_PRECISION"):
    _DEFAULT_PRECISION = _os.environ["IFD_PRECISION"]


def default_precision() -> str:
    return _DEFAULT_PRECISION


def set_default_precision(name: str) -> str:
    """Test / measurement hook: the precision used where a caller passes None.  Returns the previous default."""
    global _DEFAULT_PRECISION
    if name not in PRECISIONS:
        raise ValueError("precision must be one of %s" % sorted(PRECISIONS))
    prev, _DEFAULT_PRECISION = _DEFAULT_PRECISION, name
    return prev


def precision_code(precision) -> int:
    if precision is None:
        precision = _DEFAULT_PRECISION
    if isinstance(precision, str):
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(PRECISIONS))
        return PRECISIONS[precision]
    return int(precision)


class IfdError(RuntimeError):
    pass


def planes_to_channel_last(planes: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Reference layout {'xz','xy','yz': [B,32,64,64]} -> device layout [B,3,64,64,32] (include/ifd.h)."""
    return torch.stack([planes[k] for k in PLANE_ORDER], dim=1).permute(0, 1, 3, 4, 2).contiguous()


def planes_from_channel_last(planes: torch.Tensor) -> Dict[str, torch.Tensor]:
    return {k: planes[:, i].permute(0, 3, 1, 2).contiguous() for i, k in enumerate(PLANE_ORDER)}


def _f32(t: torch.Tensor, device: torch.device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


class Restorer:
    """One context = one model on one GPU (ifd_create ... ifd_destroy)."""
    model_name = "convonet"

    def __init__(self, weights: np.ndarray, device=None, padding: float = 0.1, threshold: float = 0.2):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise IfdError("no GPU visible: the restoration path only runs on an MI355X (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise IfdError("Restorer needs a cuda (ROCm) device, got %s" % self.device)
        self.threshold = float(threshold)
        w = np.ascontiguousarray(weights, dtype=np.float32)
        if w.size != self.lib.ifd_weight_count():
            raise IfdError("expected %d weights, got %d" % (self.lib.ifd_weight_count(), w.size))
        cfg = IfdConfig(C.sizeof(IfdConfig), 64, 32, 32, 5, 4, 32, float(padding))
        with torch.cuda.device(self.device):
            self.ctx = self.lib.ifd_create(w.ctypes.data, w.size, C.byref(cfg), self.device.index or 0)
        if not self.ctx:
            raise IfdError((self.lib.ifd_last_error(None) or b"ifd_create failed").decode())
        self._fn_decode, self._fn_optimize = self.lib.ifd_decode_ex, self.lib.ifd_optimize
        self.n_sel = 600                  # encoder subset (convonet_3plane_mn40.yaml:7 pointcloud_n)

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.ifd_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    # ---------------------------------------------------------------- helpers
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _check(self, rc: int):
        if rc != _lib.IFD_OK:
            raise IfdError("libifd error %d: %s" % (rc, (self.lib.ifd_last_error(self.ctx) or b"").decode()))

    @staticmethod
    def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
        return None if t is None else t.data_ptr()

    def _cond(self, c) -> torch.Tensor:
        """The conditioning code in device layout (planes here; the latent code in OnetRestorer)."""
        return self._planes(c)

    def _planes(self, c) -> torch.Tensor:
        if isinstance(c, dict):
            c = planes_to_channel_last({k: _f32(v, self.device) for k, v in c.items()})
        c = _f32(c, self.device)
        if c.dim() != 5 or tuple(c.shape[1:]) != (3, 64, 64, 32):
            raise IfdError("planes must be [B,3,64,64,32] channel-last or the reference's dict of [B,32,64,64]")
        return c

    # ---------------------------------------------------------------- pre-processing
    def sor(self, pc: torch.Tensor, k: int = 2, alpha: float = 1.1, want_value: bool = False):
        """SORDefense(k, alpha) keep-mask (defense/SOR.py:22-49): pc [B,K,3] -> uint8 [B,K] (and float64 value)."""
        pc = _f32(pc, self.device)
        B, K = pc.shape[:2]
        keep = torch.empty(B, K, device=self.device, dtype=torch.uint8)
        val = torch.empty(B, K, device=self.device, dtype=torch.float64) if want_value else None
        with torch.cuda.device(self.device):
            self._check(self.lib.ifd_sor(self.ctx, pc.data_ptr(), B, K, int(k), float(alpha), keep.data_ptr(),
                                         self._ptr(val), self._stream()))
        return (keep, val) if want_value else keep

    def prepare(self, pc: torch.Tensor, keep: Optional[torch.Tensor] = None, n_sel: int = 600, n_opt: int = 1024,
                padding_scale: float = 0.9, init_sigma: float = 0.01, seed: int = 0, cloud_index_base: int = 0,
                sel_idx: Optional[torch.Tensor] = None, init_idx: Optional[torch.Tensor] = None,
                noise: Optional[torch.Tensor] = None, want_proc: bool = False):
        """preprocess_pc + init_points (opt_defense.py:114-179) for a batch.  Returns a dict with
        sel [B,n_sel,3], t_per_cloud [B], init [B,n_opt,3], n_kept [B] (and proc [B,K,3] if asked)."""
        pc = _f32(pc, self.device)
        B, K = pc.shape[:2]
        i32 = lambda t: None if t is None else t.to(device=self.device, dtype=torch.int32).contiguous()
        sel_idx, init_idx = i32(sel_idx), i32(init_idx)
        noise = None if noise is None else _f32(noise, self.device)
        keep = None if keep is None else keep.to(device=self.device, dtype=torch.uint8).contiguous()
        sel = torch.empty(B, n_sel, 3, device=self.device, dtype=torch.float32)
        tpc = torch.empty(B, device=self.device, dtype=torch.int32)
        init = torch.empty(B, n_opt, 3, device=self.device, dtype=torch.float32)
        nk = torch.empty(B, device=self.device, dtype=torch.int32)
        proc = torch.zeros(B, K, 3, device=self.device, dtype=torch.float32) if want_proc else None
        prm = IfdPrepParams(C.sizeof(IfdPrepParams), int(n_sel), int(n_opt), float(padding_scale), float(init_sigma),
                            int(seed) & 0xFFFFFFFFFFFFFFFF, int(cloud_index_base))
        with torch.cuda.device(self.device):
            self._check(self.lib.ifd_prepare(self.ctx, pc.data_ptr(), self._ptr(keep), B, K, C.byref(prm),
                                             self._ptr(sel_idx), self._ptr(init_idx), self._ptr(noise), sel.data_ptr(),
                                             tpc.data_ptr(), init.data_ptr(), nk.data_ptr(), self._ptr(proc),
                                             self._stream()))
        out = {"sel": sel, "t_per_cloud": tpc, "init": init, "n_kept": nk}
        if want_proc:
            out["proc"] = proc
        return out

    # ---------------------------------------------------------------- encoder
    def encode_points(self, sel: torch.Tensor, t_per_cloud: Optional[torch.Tensor] = None, want_c: bool = False):
        """Point-wise half of encode_inputs (pointnet.py:124-156 + scatter_mean): sel [B,T,3] -> pre-U-Net
        planes [B,3,64,64,32] channel-last (and the per-point features c [B,T,32])."""
        sel = _f32(sel, self.device)
        B, T = sel.shape[:2]
        tpc = None if t_per_cloud is None else t_per_cloud.to(device=self.device, dtype=torch.int32).contiguous()
        pre = torch.empty(B, 3, 64, 64, 32, device=self.device, dtype=torch.float32)
        c = torch.empty(B, T, 32, device=self.device, dtype=torch.float32) if want_c else None
        with torch.cuda.device(self.device):
            self._check(self.lib.ifd_encode_points(self.ctx, sel.data_ptr(), self._ptr(tpc), B, T, pre.data_ptr(),
                                                   self._ptr(c), self._stream()))
        return (pre, c) if want_c else pre

    def encode_inputs(self, sel: torch.Tensor, t_per_cloud: Optional[torch.Tensor] = None) -> torch.Tensor:
        """generator.model.encode_inputs(x) (opt_defense.py:300): [B,T,3] -> planes [B,3,64,64,32] channel-last
        (use planes_from_channel_last for the reference's dict of [B,32,64,64])."""
        sel = _f32(sel, self.device)
        B, T = sel.shape[:2]
        tpc = None if t_per_cloud is None else t_per_cloud.to(device=self.device, dtype=torch.int32).contiguous()
        planes = torch.empty(B, 3, 64, 64, 32, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._check(self.lib.ifd_encode_planes(self.ctx, sel.data_ptr(), self._ptr(tpc), B, T, planes.data_ptr(),
                                                   self._stream()))
        return planes

    def unet(self, planes_pre: torch.Tensor) -> torch.Tensor:
        """The shared U-Net on pre-U-Net planes [B,3,64,64,32] (src/encoder/unet.py:225-239)."""
        pre = _f32(planes_pre, self.device)
        out = torch.empty_like(pre)
        with torch.cuda.device(self.device):
            self._check(self.lib.ifd_unet(self.ctx, pre.data_ptr(), pre.shape[0], out.data_ptr(), self._stream()))
        return out

    # ---------------------------------------------------------------- call seams
    def decode(self, p: torch.Tensor, c, want_grad: bool = False, precision=None):
        """generator.model.decode(p, c).logits -> [B,K]; with want_grad also d(sum logits)/dp [B,K,3].  precision as in
        optimize_points (None: the module default): the seam computes in the arithmetic the optimiser would (ifd_decode_ex)."""
        planes = self._cond(c)
        p = _f32(p, self.device)
        B, K = p.shape[:2]
        logits = torch.empty(B, K, device=self.device, dtype=torch.float32)
        grad = torch.empty(B, K, 3, device=self.device, dtype=torch.float32) if want_grad else None
        with torch.cuda.device(self.device):
            self._check(self._fn_decode(self.ctx, planes.data_ptr(), p.data_ptr(), B, K, precision_code(precision),
                                            logits.data_ptr(), self._ptr(grad), self._stream()))
        return (logits, grad) if want_grad else logits

    def repulsion_loss(self, p: torch.Tensor, want_grad: bool = False, want_idx: bool = False):
        """repulsion_loss(p) -> [B] (defense/repulsion_loss.py:18-54)."""
        p = _f32(p, self.device)
        B, K = p.shape[:2]
        loss = torch.empty(B, device=self.device, dtype=torch.float32)
        grad = torch.empty(B, K, 3, device=self.device, dtype=torch.float32) if want_grad else None
        idx = torch.empty(B, K, 5, device=self.device, dtype=torch.int32) if want_idx else None
        with torch.cuda.device(self.device):
            self._check(self.lib.ifd_repulsion(self.ctx, p.data_ptr(), B, K, loss.data_ptr(), self._ptr(grad),
                                               self._ptr(idx), self._stream()))
        out = (loss,) + ((grad,) if want_grad else ()) + ((idx,) if want_idx else ())
        return out if len(out) > 1 else loss

    def optimize_points(self, opt_points: torch.Tensor, c, rep_weight: float = 1.0, iterations: int = 1000,
                        lr: float = 1e-3, loss_batch=None, normalize: bool = True,
                        state: Optional[Tuple[torch.Tensor, torch.Tensor, int]] = None,
                        return_state: bool = False, return_loss: bool = False, steps: Optional[int] = None,
                        knn_scan_every_step: bool = False, printing: bool = False, split: int = 0,
                        planes_shared: bool = False, rep_radius: float = 0.07, rep_h: float = 0.03, check: bool = True,
                        precision=None, knn_reference_form: bool = False):
        """optimize_points(opt_points, z, c, rep_weight, iterations) (opt_defense.py:182-239).

        Runs ``iterations + 1`` Adam steps (the reference's ``range(iterations + 1)``) unless ``steps``
        is given.  ``loss_batch`` is the reference batch size whose 1/B factor scales both losses
        (default: the number of clouds passed in); an int, or an int32 tensor [B] with one value per cloud.  ``state=(m, v, t0)`` resumes / teacher-forces.
        ``split``: CUs per cloud (ifd_opt_params.split: 0 automatic, 1 / 2 / 4 forced; same results).
        ``knn_reference_form``: validation only - the reference's neighbour choice bug for bug (ifd_opt_params.knn_reference_form:
        float32 expanded-form distances, top-6 minus column 0, pn_utils.py:72-83) instead of the exact 5-NN.
        ``precision``: arithmetic of the decoder's dense layers (ifd_opt_params.precision): "f32" / 0 (default), "bf16x6" / 1
        (f32-equivalent on the bf16 matrix core), "bf16x3" / 2 (reduced); None takes the module default ("f32"; set_default_precision).
        ``rep_radius`` / ``rep_h``: RepulsionLoss(radius, h) (defense/repulsion_loss.py:9-10; the reference never changes them).
        ``check`` (default on): synchronise and raise IfdError on a device-side failure (``check_status``: a split cloud's
        bounded wait that gave up, repulsion sums near their range) - a caller that gets points back can trust them.  The drivers
        of pipeline.py pass ``check=False`` and check once per file instead, where the result is consumed (no synchronisation
        inside the streamed passes).
        Returns the points as a torch tensor on the device ([B,K,3]); the reference's ``.cpu().numpy()``
        is left to the caller.
        """
        if printing and state is None and not return_state and not return_loss:
            return self._optimize_points_printing(opt_points, c, rep_weight, int(iterations) + 1 if steps is None else int(steps),
                                                  lr, loss_batch, normalize, knn_scan_every_step, precision, check)
        planes = self._cond(c)
        p = _f32(opt_points, self.device).clone()
        B, K = p.shape[:2]
        n_steps = int(iterations) + 1 if steps is None else int(steps)
        t0 = 0
        m = v = None
        if state is not None:
            m, v, t0 = _f32(state[0], self.device).clone(), _f32(state[1], self.device).clone(), int(state[2])
        elif return_state:
            m, v = torch.zeros_like(p), torch.zeros_like(p)
        loss = torch.empty(B, 2, device=self.device, dtype=torch.float32) if return_loss else None
        lb_arr = None
        if torch.is_tensor(loss_batch):
            lb_arr = loss_batch.to(device=self.device, dtype=torch.int32).contiguous()
            if lb_arr.numel() != B:
                raise IfdError("loss_batch tensor must have one entry per cloud")
            loss_batch = B
        prm = IfdOptParams(C.sizeof(IfdOptParams), n_steps, t0, int(loss_batch or B), int(bool(normalize)),
                           float(lr), float(rep_weight), self.threshold, float(rep_radius), float(rep_h), 1e-12,
                           int(bool(knn_scan_every_step)), int(split or _ENV_SPLIT), int(bool(planes_shared)),
                           int(bool(knn_reference_form)), precision_code(precision))
        with torch.cuda.device(self.device):
            self._check(self._fn_optimize(self.ctx, planes.data_ptr(), p.data_ptr(), B, K, C.byref(prm),
                                              self._ptr(lb_arr), self._ptr(m), self._ptr(v), self._ptr(loss),
                                              self._stream()))
        if check:
            self.check_status()
        out = (p,)
        if return_state:
            out += ((m, v, t0 + n_steps),)
        if return_loss:
            out += (loss,)
        return out if len(out) > 1 else p

    def _optimize_points_printing(self, opt_points, c, rep_weight, n_steps, lr, loss_batch, normalize, scan, precision=None, check=True):
        """printing=True of the reference's optimize_points (opt_defense.py:229-236): at iterations 0, 100, 200, ... it
        prints the loss, the two loss terms and the mean occupancy probability, all evaluated at that iteration's
        pre-update points.  The run is cut so that every such iteration is a launch of its own (the kernel reports the
        losses of a launch's last step; the probability comes from one ifd_decode of the points it starts from) and the
        Adam state is carried across: the result is bit-identical to the uncut run (the neighbour lists are exact at every
        step, the moments are passed on as they are).
        Scaling: the reference prints per BATCH of `batch_size` clouds; a device pass here may hold several reference
        batches (or a shard of one), so the printed terms are the means over this pass's clouds of the per-cloud terms -
        equal to the reference's scalars when the pass is exactly one reference batch, their average otherwise."""
        p, st, t = opt_points, None, 0
        B = opt_points.shape[0]
        while t < n_steps:
            nxt = t if t % 100 == 0 else min(n_steps, (t // 100 + 1) * 100)     # next printing iteration (or the end)
            if nxt > t:                                                          # plain steps up to it
                last = nxt == n_steps
                p, st = self.optimize_points(p, c, rep_weight=rep_weight, steps=nxt - t, lr=lr, loss_batch=loss_batch,
                                             normalize=normalize and last, state=st, return_state=True,
                                             knn_scan_every_step=scan, precision=precision, check=check)
                t = nxt
                continue
            prob = float(torch.sigmoid(self.decode(p, c, precision=precision)).mean())   # occ_value of iteration t (pre-update points)
            last = t + 1 == n_steps
            p, st, loss = self.optimize_points(p, c, rep_weight=rep_weight, steps=1, lr=lr, loss_batch=loss_batch,
                                               normalize=normalize and last, state=st, return_state=True, return_loss=True,
                                               knn_scan_every_step=scan, precision=precision, check=check)
            l = loss.double().cpu()
            if torch.is_tensor(loss_batch):
                lbv = loss_batch.double().cpu()
            else:
                lbv = torch.full((B,), float(loss_batch or B), dtype=torch.float64)
            # per-cloud terms carry their own 1/B; the reference's scalars are sums over the batch of B clouds
            occ = float((l[:, 0]).sum()) * float(B) / float(lbv.sum()) if lbv.numel() else 0.0
            rep = float(l[:, 1].mean()) * rep_weight
            print('iter {}, loss {:.4f}'.format(t, occ + rep))
            print('occ loss: {:.4f}, rep loss: {:.4f}\nocc value mean: {:.4f}'.format(occ, rep, prob))
            t += 1
        return p

    def check_status(self):
        """ifd_optimize_status: raise IfdError if an optimise launch since the last check failed on the device - a cross-CU
        wait of a split cloud that gave up, fixed-point repulsion sums near their range (include/ifd.h).  Synchronises the
        current stream."""
        with torch.cuda.device(self.device):
            self._check(self.lib.ifd_optimize_status(self.ctx, self._stream()))

    def counters(self) -> Dict[str, int]:
        """Diagnostic counters of the last optimize_points call (synchronises)."""
        buf = (C.c_uint64 * 16)()
        self._check(self.lib.ifd_get_counters(self.ctx, buf, 16))
        return {"knn_rebuilds": int(buf[0]), "knn_brute_scans": int(buf[1]), "knn_passes": int(buf[2]),
                "cloud0_shader_cycles": int(buf[3]), "knn_ring_evals": int(buf[4]), "knn_exact_evals": int(buf[5]), "knn_refresh_waves": int(buf[6]), "knn_lists_built": int(buf[7]),
                "mesh_points": int(buf[8]), "mesh_rounds": int(buf[9]),
                "prof_cycles": [int(buf[i]) for i in range(8, 16)]}      # only in -DIFD_PROF diagnostic builds

    def wave_trace(self):
        """Time stamps (shader cycles) of one optimiser step of cloud 0, [8 waves][32 slots] - only filled by a
        -DIFD_TRACE diagnostic build of libifd.so (scripts/trace_step.py)."""
        buf = (C.c_uint64 * (16 + 8 * 32))()
        self._check(self.lib.ifd_get_counters(self.ctx, buf, 16 + 8 * 32))
        return [[int(buf[16 + w * 32 + i]) for i in range(32)] for w in range(8)]

    def tile_trace(self):
        """Time stamps (shader cycles) inside ONE decoder tile per wave of the traced step, [8 waves][128 slots] - only filled
        by a -DIFD_TRACE -DIFD_TRACE2=<n> diagnostic build (scripts/tile_trace.py; slot map in optimize.hip decoder_tile3)."""
        base = 16 + 8 * 32 + 3          # behind the three status words (ifd_internal.h TRACE2_BASE)
        buf = (C.c_uint64 * (base + 8 * 128))()
        self._check(self.lib.ifd_get_counters(self.ctx, buf, base + 8 * 128))
        return [[int(buf[base + w * 128 + i]) for i in range(128)] for w in range(8)]

    def normalize_batch_pc(self, points: torch.Tensor) -> torch.Tensor:
        """normalize_batch_pc (opt_defense.py:76-83); returns a new tensor."""
        p = _f32(points, self.device).clone()
        with torch.cuda.device(self.device):
            self._check(self.lib.ifd_normalize_unit_sphere(self.ctx, p.data_ptr(), p.shape[0], p.shape[1],
                                                           self._stream()))
        return p


class OnetRestorer(Restorer):
    """ONet-Opt (ONet/opt_defense.py): same call seams, Occupancy-Network model (ifd_onet_create).

    ``encode_inputs`` returns the latent code c [B,512] (generator.model.encode_inputs, :300); ``decode`` and
    ``optimize_points`` take it where ConvONet takes the planes (z is empty: z_dim 0 in configs/onet_mn40.yaml)."""
    model_name = "onet"

    def __init__(self, weights: np.ndarray, device=None, threshold: float = 0.2):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise IfdError("no GPU visible: the restoration path only runs on an MI355X (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise IfdError("OnetRestorer needs a cuda (ROCm) device, got %s" % self.device)
        self.threshold = float(threshold)
        w = np.ascontiguousarray(weights, dtype=np.float32)
        if w.size != self.lib.ifd_onet_weight_count():
            raise IfdError("expected %d weights, got %d" % (self.lib.ifd_onet_weight_count(), w.size))
        with torch.cuda.device(self.device):
            self.ctx = self.lib.ifd_onet_create(w.ctypes.data, w.size, self.device.index or 0)
        if not self.ctx:
            raise IfdError((self.lib.ifd_last_error(None) or b"ifd_onet_create failed").decode())
        self._fn_decode, self._fn_optimize = self.lib.ifd_onet_decode_ex, self.lib.ifd_onet_optimize
        self.n_sel = 300                  # onet_mn40.yaml:6 pointcloud_n

    def _cond(self, c) -> torch.Tensor:
        c = _f32(c, self.device)
        if c.dim() != 2 or c.shape[1] != 512:
            raise IfdError("the ONet conditioning code must be [B,512]")
        return c

    def encode_inputs(self, sel: torch.Tensor, t_per_cloud: Optional[torch.Tensor] = None) -> torch.Tensor:
        sel = _f32(sel, self.device)
        B, T = sel.shape[:2]
        tpc = None if t_per_cloud is None else t_per_cloud.to(device=self.device, dtype=torch.int32).contiguous()
        c = torch.empty(B, 512, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._check(self.lib.ifd_onet_encode(self.ctx, sel.data_ptr(), self._ptr(tpc), B, T, c.data_ptr(),
                                                 self._stream()))
        return c

    def mesh_sample(self, c: torch.Tensor, n_sample: int = 1024, resolution0: int = 32, upsampling_steps: int = 2,
                    padding: float = 0.1, seed: int = 0, cloud_index_base: int = 0, max_triangles: int = 400000,
                    want_grid: bool = False, want_triangles: bool = False, threshold: Optional[float] = None, precision=None):
        """reconstruct_mesh + trimesh.sample.sample_surface (ONet/remesh_defense.py:128-157) for a batch of latent codes:
        c [B,512] -> dict(points [B,n_sample,3] (not normalised), n_triangles [B] int32, optionally grid [B,P,P,P] and
        triangles [B,max_triangles,9]).  ``precision``: arithmetic of the grid evaluation's decoder layers (ifd_mesh_params.precision,
        "f32" default / "bf16x6" / "bf16x3")."""
        c = self._cond(c)
        B = c.shape[0]
        P = (resolution0 << upsampling_steps) + 1
        pts = torch.zeros(B, n_sample, 3, device=self.device, dtype=torch.float32)
        ntri = torch.zeros(B, device=self.device, dtype=torch.int32)
        grid = torch.empty(B, P, P, P, device=self.device, dtype=torch.float32) if want_grid else None
        tris = torch.zeros(B, max_triangles, 9, device=self.device, dtype=torch.float32) if want_triangles else None
        prm = IfdMeshParams(C.sizeof(IfdMeshParams), int(resolution0), int(upsampling_steps), int(n_sample), int(max_triangles),
                            float(padding), float(self.threshold if threshold is None else threshold), int(seed),
                            int(cloud_index_base), precision_code(precision), 0)
        with torch.cuda.device(self.device):
            self._check(self.lib.ifd_onet_mesh_sample(self.ctx, c.data_ptr(), B, C.byref(prm), pts.data_ptr(), ntri.data_ptr(),
                                                      self._ptr(grid), self._ptr(tris), self._stream()))
        out = {"points": pts, "n_triangles": ntri}
        if want_grid:
            out["grid"] = grid
        if want_triangles:
            out["triangles"] = tris
        return out

    def encode_points(self, *a, **k):
        raise IfdError("encode_points / unet belong to the ConvONet model")

    unet = encode_points
