"""studiosr.models.common surface (reference: studiosr/models/common.py) on the HIP hot path.

`Model` keeps the reference's public contract -- attributes scale / n_colors / img_range, `inference`,
`inference_with_self_ensemble`, `get_model_config`, `get_training_config`, `from_pretrained`, `export`
(common.py:29-98) -- while `forward` of every subclass runs hand-written HIP kernels through the C ABI.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import threading

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops, packing
from ..runtime import Workspace, compute_dtype, require_device, sr_dtype

Tensor = torch.Tensor
RGB_MEAN = (0.4488, 0.4371, 0.4040)  # common.py:112,223
_PREC_TLS = threading.local()  # per-thread precision overrides of inference(): {id(model): precision}


class Model(nn.Module):
    """Base class (reference: studiosr/models/common.py:29-98)."""

    def __init__(self, scale: int = 4, n_colors: int = 3, img_range: float = 1.0) -> None:
        super().__init__()
        self.scale: int = scale
        self.n_colors: int = n_colors
        self.img_range: float = img_range
        # 'auto': bf16 under torch.autocast(bfloat16) (reference Trainer), fp32 otherwise (reference inference)
        self.precision: str = "auto"
        self._packed: Dict = {}
        self._ws: Optional[Workspace] = None

    # ------------------------------------------------------------------ HIP plumbing
    def set_precision(self, precision: str) -> "Model":
        compute_dtype(precision)  # validates
        self.precision = precision
        return self

    def _param_version(self):
        """(data_ptr, _version) of every parameter and buffer: in-place updates through autograd-visible ops (optimizer.step,
        copy_, load_state_dict) bump _version, .to()/.cuda() change data_ptr.  Edits through `.data` are invisible to both:
        call repack() after them."""

        def ver(t):
            try:
                return t._version
            except RuntimeError:  # inference tensors do not track versions
                return -1

        return tuple((t.data_ptr(), ver(t)) for t in list(self.parameters()) + list(self.buffers()))

    def repack(self) -> "Model":
        """Drop the fragment-packed weight cache (needed only after editing parameters through `.data`)."""
        self._packed = {}
        return self

    def load_state_dict(self, *args, **kwargs):
        self._packed = {}
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):  # .to() / .cuda() / .half(): parameters are replaced
        self._packed = {}
        self._ws = None
        return super()._apply(fn, *args, **kwargs)

    def invalidate_packed(self) -> None:
        """Drop every cache derived from the parameters (fragment-ordered weights of the inference path, the fused training plan's packed operands).
        `_get_packed` notices ordinary in-place updates through the parameters' version counters; an update that does not bump them
        (torch.optim.Adam(fused=True): torch._fused_adam_; raw kernels writing p.data) needs this call -- studiosr_amd.optim.Adam and train() / eval() make it."""
        self._packed = {}
        self.__dict__["_recorded_since_pack"] = False
        plan = self.__dict__.get("_fast_plan")
        if plan is not None:
            plan.packed_version = None

    def train(self, mode: bool = True):
        # The Trainer's train -> evaluate -> train pattern (trainer.py:125-131): never evaluate on weights packed before the last optimizer steps.  Only when a
        # differentiable forward ran since the cache was filled (a pure inference user toggling modes keeps its packed weights: a captured HIP graph holds pointers to them).
        if mode != self.training and self.__dict__.get("_recorded_since_pack"):
            self.invalidate_packed()
        return super().train(mode)

    def __getstate__(self):  # deepcopy / pickle: derived device state (workspace, packed weights) is rebuilt on demand
        state = dict(self.__dict__)
        state["_packed"], state["_ws"] = {}, None
        state.pop("_side", None)  # HIP stream(s) of the two-branch blocks / part batches: per process, re-created on demand
        state.pop("_part_streams", None)
        state.pop("_fast_plan", None)  # the fused training plan (static device buffers, events, offsets keyed by parameter identity): rebuilt by the copy's first step
        return state

    def __call__(self, *args, **kwargs):
        """Launches go to the CURRENT device's current stream: make the input's device current for the whole forward
        (model.to('cuda:1') with device 0 current is the standard multi-GPU idiom)."""
        from ..runtime import x3_mode

        x = args[0] if args else None
        with x3_mode(self.precision == "fp32x3"):
            if isinstance(x, torch.Tensor) and x.is_cuda and x.device.index != torch.cuda.current_device():
                with torch.cuda.device(x.device):
                    return super().__call__(*args, **kwargs)
            return super().__call__(*args, **kwargs)

    def _get_packed(self, dt) -> Dict:
        """Fragment-ordered weights for compute dtype dt, rebuilt when a parameter changes."""
        if self.precision == "fp32x3" and dt == torch.float32:
            from ..runtime import X3_KEY

            dt = X3_KEY  # split-operand packing (hi | lo); everything else of the fp32 path is unchanged
        ver = self._param_version()
        ent = self._packed.get(dt)
        if ent is None or ent["__version__"] != ver:
            with torch.no_grad():
                ent = self._pack(dt)
            ent["__version__"] = ver
            self._packed[dt] = ent
        return ent

    def _pack(self, dt: torch.dtype) -> Dict:  # pragma: no cover - abstract
        raise NotImplementedError

    def _workspace(self, device: torch.device) -> Workspace:
        if self._ws is None or self._ws.device != device:
            self._ws = Workspace(device)
        self._ws.begin_forward()
        return self._ws

    def _check_input(self, x: Tensor) -> Tensor:
        require_device(x)
        pdev = next(self.parameters()).device
        if pdev != x.device:
            raise RuntimeError(f"input is on {x.device} but the model parameters are on {pdev}")
        if x.dim() != 4 or x.shape[1] != self.n_colors:
            raise RuntimeError(f"expected input [B,{self.n_colors},H,W], got {tuple(x.shape)}")
        return x.detach().to(torch.float32).contiguous()

    def _train_forward(self, x: Tensor) -> Optional[Tensor]:
        """The differentiable path (studiosr_amd/models/train.py: a graph of HIP forward / backward ops) whenever autograd is recording
        -- in train() AND eval() mode, so gradients never silently vanish -- and whenever DropPath is active (train() with
        drop_path_rate > 0, swinir.py:137,171-172 / hat.py:148,192-193), also under no_grad.  None = take the inference path."""
        recording = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        stochastic = self.training and getattr(self, "drop_path_rate", 0.0) > 0.0
        if not (recording or stochastic):
            return None
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("the HIP training path provides parameter gradients only (the reference Trainer never differentiates w.r.t. the LR image)")
        from . import train

        if recording:
            self.__dict__["_recorded_since_pack"] = True  # (see train(): an optimizer step may follow, visibly or not)
        return train.FORWARDS[type(self).__name__](self, self._check_input(x))

    # ------------------------------------------------------------------ reference API
    def _io_scale(self) -> float:
        return 255.0 if self.img_range == 1.0 else 1.0  # common.py:39

    def _inference_precision(self):
        """Context for inference() / inference_batch() / inference_with_self_ensemble(): precision "auto" outside bf16 autocast runs the
        reference-precision FAST path "fp32x3" (fp32 tensors and op order, every contraction as split-operand bf16 with fp32
        accumulation: max |error| 3e-6 against the exact-fp32 path, i.e. the metric's 1e-3 dB PSNR bar with 4 orders of margin, at 4x
        (SwinIR) / 2.5x (EDSR) its speed).  `set_precision("fp32")` keeps the exact parity mode; forward() under "auto" is unchanged.
        The override is per THREAD (the `precision` property reads it): the model object itself is not modified, so a forward running
        concurrently on another thread keeps its own precision."""
        model = self

        class _Ctx:
            def __enter__(self_inner):
                ov = getattr(_PREC_TLS, "ov", None)
                if ov is None:
                    ov = _PREC_TLS.ov = {}
                self_inner.prev = ov.get(id(model))
                if model._precision == "auto" and self_inner.prev is None and compute_dtype("auto") == torch.float32:
                    ov[id(model)] = "fp32x3"

            def __exit__(self_inner, *exc):
                ov = _PREC_TLS.ov
                if self_inner.prev is None:
                    ov.pop(id(model), None)
                else:
                    ov[id(model)] = self_inner.prev

        return _Ctx()

    @property
    def precision(self) -> str:
        ov = getattr(_PREC_TLS, "ov", None)
        if ov:
            v = ov.get(id(self))
            if v is not None:
                return v
        return self._precision

    @precision.setter
    def precision(self, value: str) -> None:
        object.__setattr__(self, "_precision", value)

    @torch.inference_mode()
    def inference(self, image: np.ndarray) -> np.ndarray:
        """uint8 HWC -> uint8 HWC (studiosr/models/common.py:36-48): /scale, NCHW, forward, *scale, round-half-even, clip,
        uint8; scale = 255 iff img_range == 1.0.  The uint8 image crosses PCIe as uint8; both conversions are HIP kernels
        with the reference's fp32 arithmetic (sr_u8_to_nchw / sr_nchw_to_u8)."""
        return self.inference_batch([image])[0]

    @torch.inference_mode()
    def inference_batch(self, images: List[np.ndarray]) -> List[np.ndarray]:
        """`inference` of several images at once: images of equal size share one forward (batch items are independent, so
        every result equals the one-image call); results come back in input order."""
        self.eval()
        scale = self._io_scale()
        device = next(self.parameters()).device
        require_device(torch.empty(0, device=device))
        outs: List[Optional[np.ndarray]] = [None] * len(images)
        groups: Dict = {}
        for i, im in enumerate(images):
            if im.ndim != 3 or im.shape[2] != self.n_colors:
                raise RuntimeError(f"expected uint8 image [H,W,{self.n_colors}], got {tuple(im.shape)}")
            groups.setdefault(tuple(im.shape), []).append(i)
        for idxs in groups.values():
            u8 = torch.from_numpy(np.ascontiguousarray(np.stack([images[i] for i in idxs]).astype(np.uint8, copy=False))).to(device)
            with self._inference_precision():
                y = ops.nchw_to_u8(self(ops.u8_to_nchw(u8, scale)).contiguous(), scale).cpu().numpy()
            for j, i in enumerate(idxs):
                outs[i] = y[j]
        return outs  # type: ignore[return-value]

    @torch.inference_mode()
    def inference_with_self_ensemble(self, image: np.ndarray) -> np.ndarray:
        """8 rot/flip variants averaged (studiosr/models/common.py:10-26,50-67).  Variants of equal shape are
        batched into one forward (the reference runs 8 sequential forwards; results are identical because
        batch items are independent)."""
        self.eval()
        scale = self._io_scale()
        device = next(self.parameters()).device
        img = torch.from_numpy(np.ascontiguousarray(image.astype(np.uint8, copy=False))).to(device)  # rot / flip on uint8: exact
        variants = []
        for k in range(4):
            r = torch.rot90(img, k, dims=[0, 1])
            variants += [r, torch.fliplr(r)]
        outs: List[Optional[Tensor]] = [None] * 8
        groups: Dict = {}
        for i, v in enumerate(variants):
            groups.setdefault(tuple(v.shape), []).append(i)
        for idxs in groups.values():
            xb = ops.u8_to_nchw(torch.stack([variants[i] for i in idxs]).contiguous(), scale)
            with self._inference_precision():
                yb = self(xb)
            for j, i in enumerate(idxs):
                outs[i] = yb[j].permute(1, 2, 0)
        merged = []
        for i, o in enumerate(outs):
            o = torch.fliplr(o) if i & 1 else o
            merged.append(torch.rot90(o, i // 2, dims=[1, 0]))
        out = torch.stack(merged).mean(dim=0)  # converge_images (common.py:19-26)
        return ops.nchw_to_u8(out.permute(2, 0, 1).unsqueeze(0).contiguous(), scale)[0].cpu().numpy()

    def get_model_config(self) -> Dict:
        return dict(scale=self.scale, n_colors=self.n_colors, img_range=self.img_range)

    def get_training_config(self) -> Dict:
        return dict()

    @classmethod
    def from_pretrained(cls, scale: int = 4) -> "Model":
        return cls(scale=scale)

    def export(self, path: Optional[str] = None, input_shape: List[int] = [1, 3, 256, 256], format: str = "onnx") -> str:
        """Reference: common.py:86-98 (torch.onnx.export of the module's traced forward).  The forward here is a HIP launch sequence, which
        torch.onnx cannot trace, so the export is ROUTED: format "checkpoint" writes the model in the reference's own
        terms -- class name, get_model_config() and the state_dict with the reference's keys (torch.save) -- which the reference class loads with
        `cls(**config).load_state_dict(...)` and exports with its own export(); format "onnx" says so instead of writing a wrong graph."""
        format = format.lower()
        if format == "onnx":
            raise NotImplementedError(
                "the MI355X forward is a HIP launch sequence and cannot be traced to ONNX: use export(format='checkpoint') and export that "
                "checkpoint with the reference implementation's export() (same class name, config and state_dict keys)")
        if format != "checkpoint":
            raise ValueError(f"unknown export format {format!r} (expected 'onnx' or 'checkpoint')")
        if path is None:
            path = f"{self.__class__.__name__}x{self.scale}.pth"
        torch.save({"class": self.__class__.__name__, "config": self.get_model_config(), "input_shape": list(input_shape),
                    "state_dict": {k: v.detach().cpu() for k, v in self.state_dict().items()}}, path)
        return path


BaseModule = Model


def conv2d(in_channels: int, out_channels: int, kernel_size: int) -> nn.Conv2d:
    """Parameter container with the reference's shapes/init (common.py:104-105)."""
    return nn.Conv2d(in_channels, out_channels, kernel_size, padding=kernel_size // 2)


class Upsampler(nn.Sequential):
    """Parameter container for the conv->PixelShuffle chain (common.py:124-137); state_dict keys
    `0.weight`, `2.weight`, ... as in the reference.  Executed by `run_upsampler`."""

    def __init__(self, scale: int, n_feats: int, num_out_ch: Optional[int] = None) -> None:
        m: List[nn.Module] = []
        self.stages: List = []
        if num_out_ch is not None:
            m += [conv2d(n_feats, scale * scale * num_out_ch, 3), nn.PixelShuffle(scale)]
            stages = [(0, scale, num_out_ch)]
        elif (scale & (scale - 1)) == 0:
            stages = []
            for i in range(int(round(np.log2(scale)))):
                m += [conv2d(n_feats, 4 * n_feats, 3), nn.PixelShuffle(2)]
                stages.append((2 * i, 2, n_feats))
        else:
            m += [conv2d(n_feats, scale * scale * n_feats, 3), nn.PixelShuffle(scale)]
            stages = [(0, scale, n_feats)]
        super().__init__(*m)
        self.stages = stages  # (index of conv in the Sequential, r, channels after the shuffle)


def pack_upsampler(up: Upsampler, cin_p: int, dt: torch.dtype, last_cps_p: Optional[int] = None):
    """[(Wp, bias, r, cps_p)] for every conv+PixelShuffle stage; weight rows permuted so that the conv
    epilogue can store straight through the shuffle."""
    out = []
    for si, (idx, r, c_ps) in enumerate(up.stages):
        conv = up[idx]
        cps_p = packing.round_up(c_ps, 32)
        if last_cps_p is not None and si == len(up.stages) - 1:
            cps_p = last_cps_p
        rows = packing.pixel_shuffle_rows(c_ps, cps_p, r)
        wp, b = packing.pack_conv3x3(conv.weight, conv.bias, cin_p, rows, dt)
        out.append((wp, b, r, cps_p))
        cin_p = cps_p
    return out


def conv_call(x: Tensor, wp: Tensor, bias: Optional[Tensor], out: Tensor, cdt: torch.dtype, *, act: int = L.ACT_NONE,
              out_scale: float = 1.0, skip: Optional[Tensor] = None, out_mode: int = L.OUT_NHWC, ps_r: int = 0, cps_p: int = 0,
              pool: Optional[Tensor] = None, fin=None, cout_p: Optional[int] = None, act_slope: float = 0.0, tile_rows: int = 0) -> Tensor:
    """One sr_conv3x3 launch on NHWC tensors (shapes are taken from the tensors)."""
    B, H, W, cin_p = x.shape
    if cout_p is None:
        cout_p = out.shape[-1] if out_mode == L.OUT_NHWC else ps_r * ps_r * cps_p
    kw = dict(
        x=x.data_ptr(), Wp=wp.data_ptr(), bias=None if bias is None else bias.data_ptr(), out=out.data_ptr(),
        skip=None if skip is None else skip.data_ptr(), pool_partial=None if pool is None else pool.data_ptr(),
        B=B, H=H, W=W, Cin_p=cin_p, Cout_p=cout_p,
        x_dtype=sr_dtype(x.dtype), out_dtype=sr_dtype(out.dtype), skip_dtype=sr_dtype(skip.dtype) if skip is not None else 0,
        compute_dtype=sr_dtype(cdt), act=act, out_scale=out_scale, out_mode=out_mode, ps_r=ps_r, cps_p=cps_p, act_slope=act_slope, tile_rows=tile_rows,
    )
    if fin is not None:
        fscale, fbias, fc, fh, fw = fin
        kw.update(fin_scale=fscale.data_ptr(), fin_bias=fbias.data_ptr(), fin_c=fc, fin_h=fh, fin_w=fw)
    ops.conv3x3(**kw)
    return out


def run_upsampler(packed, x: Tensor, ws: Workspace, cdt: torch.dtype, name: str) -> Tensor:
    for si, (wp, b, r, cps_p) in enumerate(packed):
        B, H, W, _ = x.shape
        out = ws.get(f"{name}.up{si}", (B, H * r, W * r, cps_p), cdt)
        conv_call(x, wp, b, out, cdt, out_mode=L.OUT_PIXEL_SHUFFLE, ps_r=r, cps_p=cps_p)
        x = out
    return x
