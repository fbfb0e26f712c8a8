"""Host mirror of the reference's training iteration over ``include/nsr_train.h``.

``Trainer`` holds what ``NeRFDownXModel`` holds in train mode (``models/nerf_downX_model.py:133-205``): the two
networks' 24-tensor state dicts as fp32 device tensors, their gradients, and the Adam moments; its methods follow
the reference's call protocol:

    set_input(rays, rgbs)  ->  optimize_parameters()            (:235-248, :398-408)
        = forward (train mode) + comp_low_res_output + calculate_losses + backward + optimizer.step

All arithmetic runs in libnsr.so (``nsr_train_loss_and_grads``, ``nsr_adam_step``); torch supplies device memory,
the stream and the random draws (``torch.rand`` / ``torch.randn`` on the device, exactly the four tensors the
reference draws).  There is no CPU path.
"""
from __future__ import annotations

from ctypes import c_void_p
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .ops import _f32, _p, _ray_stride, _stream
from .weights import STATE_DICT_SPEC, check_state_dict, pad_no_dir, DIR_W

OUT_KEYS = ("coarse_comp_rgbs", "coarse_depth", "coarse_opacity", "coarse_weights",
            "fine_comp_rgbs", "fine_depth", "fine_opacity", "fine_weights")


def _ptr_array(tensors):
    return (c_void_p * len(tensors))(*[c_void_p(t.data_ptr()) for t in tensors])


def _flat_like(params: Dict[str, torch.Tensor]):
    """One flat fp32 buffer + per-tensor views with the state_dict's shapes (so a whole network's gradients or
    Adam moments are ONE contiguous range: a single all-reduce / memset covers them).  Views start 16-byte aligned."""
    sizes = [(k, v.numel(), tuple(v.shape)) for k, v in params.items()]
    offs, total = [], 0
    for _, n, _ in sizes:
        offs.append(total)
        total += (n + 3) // 4 * 4
    flat = torch.zeros(total, dtype=torch.float32, device=next(iter(params.values())).device)
    views = {k: flat[o:o + n].view(shape) for (k, n, shape), o in zip(sizes, offs)}
    return flat, views


def _to_dev(sd, device, no_dir: bool = False) -> Dict[str, torch.Tensor]:
    check_state_dict(sd, no_dir)
    out = {}
    for k in STATE_DICT_SPEC:
        v = sd[k]
        v = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v.detach()
        v = v.to(device=device, dtype=torch.float32)
        out[k] = (pad_no_dir(v) if (no_dir and k == DIR_W) else v).contiguous().clone()
    return out


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None, act: int = 0, transposed: bool = False):
    """One ``nn.Linear`` (+ activation) on the training GEMM (``nsr_linear``): y = act(x w^T + b).
    K must be a multiple of 32; returns y (P, N) and, if asked, also y^T (N, P)."""
    x, w = _f32(x, "x"), _f32(w, "w")
    P, K = x.shape
    N = w.shape[0]
    y = torch.empty(P, N, dtype=torch.float32, device=x.device)
    yt = torch.empty(N, P, dtype=torch.float32, device=x.device) if transposed else None
    _lib.check(_lib.load().nsr_linear(_p(x), K, _p(w), K, _p(None if b is None else _f32(b, "b")), act, _p(y), N,
                                      _p(yt), P, P, K, N, _stream()), "nsr_linear")
    return (y, yt) if transposed else y


class Trainer:
    """Coarse + fine networks, gradients, Adam state and one-call training iterations."""

    def __init__(self, sd_coarse, sd_fine, N_coarse: int = 64, N_importance: int = 64, white_bkgd: bool = False,
                 lindisp: bool = False, downscale: int = 2, randomized: bool = True, noise_std: float = 0.0,
                 lr: float = 5e-4, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                 lambda_coarse_mse: float = 1.0, lambda_fine_mse: float = 1.0, ray_chunk: int = 4096,
                 precision: str = "f16x3", device="cuda", gamma_correct: bool = False,
                 use_var_loss: bool = False, lambda_coarse_var: float = 0.01, lambda_fine_var: float = 0.01,
                 use_depth_var_loss: bool = False, lambda_coarse_depth_var: float = 0.01, lambda_fine_depth_var: float = 0.01,
                 no_dir: bool = False, sigma_activation: str = "relu", color_activation: str = "sigmoid", stop_grad: bool = False):
        # render_rays applies rgb ** (1 / 2.2) per sample under --gamma_correct (models/nerf_downX_model.py:271-276) in training
        # too: NSR_TRAIN_GAMMA_CORRECT of the step's option word (include/nsr_train.h)
        self.gamma_correct = bool(gamma_correct)
        # --sigma_activation softplus (models/rendering.py:69-73), --color_activation none (models/networks.py:173-180)
        if sigma_activation not in ("relu", "softplus") or color_activation not in ("sigmoid", "none"):
            raise ValueError("sigma_activation: 'relu' or 'softplus'; color_activation: 'sigmoid' or 'none'")
        if gamma_correct and color_activation == "none":
            raise ValueError("gamma_correct with color_activation='none': pow(rgb, 1 / 2.2) of an unbounded head is NaN for every negative value")
        self.sigma_activation, self.color_activation = sigma_activation, color_activation
        self.stop_grad = bool(stop_grad)       # --stop_grad (models/networks.py:127, 218-219): the colour branch's input is detached
        if precision not in _lib.TRAIN_PRECISIONS:
            raise ValueError("precision must be 'fp32' (every product on the fp32 MFMA, layer by layer), 'f16x3' (the chain "
                             "kernels: forward on the split-fp16 MFMA, weight gradients on one fp16 MFMA per product, input "
                             "gradients on the default number of MFMA terms), 'f16x3_bwd3' / 'f16x3_bwd2' / 'f16x3_bwd1' / 'f16x3_bwdm' (the "
                             "chain kernels with three / two / one / mixed two-one MFMAs per product of the input gradients, include/nsr_train.h) "
                             "or 'f16x3_gemm' (layer by layer, forward products split-fp16, gradients fp32)")
        self.precision, self._prec = precision, _lib.TRAIN_PRECISIONS[precision]
        self.device = torch.device(device)
        # --no_dir (models/networks.py:160-169): the networks are trained as the full layout with 27 zero columns in
        # dir_encoding's weight (weights.pad_no_dir) whose gradients are dropped every step, so Adam never moves them
        # (m = v = 0 -> update 0): the function, its gradients and the trajectory are the narrow network's
        self.no_dir = bool(no_dir)
        self.params = [_to_dev(sd_coarse, self.device, self.no_dir), _to_dev(sd_fine, self.device, self.no_dir)]
        flat = [_flat_like(p) for p in self.params]
        self.flat_grads, self.grads = [f for f, _ in flat], [v for _, v in flat]
        self.exp_avg = [_flat_like(p)[1] for p in self.params]
        self.exp_avg_sq = [_flat_like(p)[1] for p in self.params]
        # data parallelism: every rank scales its loss by 1 / world so that the all-reduce SUM is the gradient of the
        # global-batch MEAN loss, like DistributedDataParallel's averaging (models/networks.py:84).  None = derive it
        # from the default process group at each step; a number pins it (e.g. a custom group).
        self.grad_scale: Optional[float] = None
        self.group = None          # process group of the data-parallel replicas (None = the default group); the 1 / world
        #                            loss scale, the gradient all-reduce and the finite-loss vote all use THIS group
        self.check_finite = False  # debug: raise when the loss is not finite (see optimize_parameters)
        self.N_coarse, self.N_importance = int(N_coarse), int(N_importance)
        self.white_bkgd, self.lindisp = bool(white_bkgd), bool(lindisp)
        self.s2 = int(downscale) ** 2
        self.randomized, self.noise_std = bool(randomized), float(noise_std)
        self.lr, self.beta1, self.beta2, self.eps = float(lr), float(beta1), float(beta2), float(eps)
        self.lambda_coarse, self.lambda_fine = float(lambda_coarse_mse), float(lambda_fine_mse)
        # the optional variance losses of comp_low_res_output (models/nerf_downX_model.py:107-112, 332-336, 349-353, 374-378);
        # 0 = off, like the reference's flags.  [coarse rgb, fine rgb, coarse depth, fine depth]
        self.lambda_var = [float(lambda_coarse_var) if use_var_loss else 0.0, float(lambda_fine_var) if use_var_loss else 0.0,
                           float(lambda_coarse_depth_var) if use_depth_var_loss else 0.0,
                           float(lambda_fine_depth_var) if use_depth_var_loss else 0.0]
        self.var_losses = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.ray_chunk = int(ray_chunk) - int(ray_chunk) % self.s2
        self.step = 0
        self._ws = None
        self.losses = torch.zeros(2, dtype=torch.float32, device=self.device)
        self.out: Dict[str, torch.Tensor] = {}

    # -- reference protocol ------------------------------------------------------------------------------
    def set_input(self, rays: torch.Tensor, rgbs: torch.Tensor):
        """rays (N_lr, s2, 8 | 11) or (N_lr * s2, 8 | 11); rgbs (N_lr, 3) LR targets (set_input, :235-248)."""
        rays = _f32(rays.reshape(-1, rays.shape[-1]), "rays")
        self.data_rays, self.data_rgbs = rays, _f32(rgbs.reshape(-1, 3), "rgbs")
        if rays.shape[0] != self.data_rgbs.shape[0] * self.s2:
            raise ValueError("rays must hold s^2 sub-rays per LR target pixel")

    def draw(self, R: int) -> Dict[str, Optional[torch.Tensor]]:
        """The random tensors of one train-mode forward, in the reference's order (models/utils.py:40, 210, 73)."""
        d = {"u_coarse": None, "noise_coarse": None, "u_fine": None, "noise_fine": None}
        if self.randomized:
            nc, nf = self.N_coarse, self.N_coarse + self.N_importance
            d["u_coarse"] = torch.rand(R, nc, device=self.device)
            if self.noise_std > 0:
                d["noise_coarse"] = torch.randn(R, nc, device=self.device)
            d["u_fine"] = torch.rand(R, self.N_importance, device=self.device)
            if self.noise_std > 0:
                d["noise_fine"] = torch.randn(R, nf, device=self.device)
        return d

    def loss_and_grads(self, draws: Optional[Dict[str, Optional[torch.Tensor]]] = None):
        """forward + comp_low_res_output + calculate_losses + backward (:316-396): fills ``self.out``,
        ``self.losses`` (device float[2], UNscaled: this rank's lambda-weighted MSEs) and ``self.grads`` (already
        scaled by 1 / world for the data-parallel SUM, see ``grad_scale``)."""
        from .dist import _world
        gs = float(self.grad_scale) if self.grad_scale is not None else 1.0 / _world(self.group)[1]
        rays = self.data_rays
        R, stride = rays.shape[0], _ray_stride(rays)
        if draws is None:
            draws = self.draw(R)
        draws = {k: (None if v is None else _f32(torch.as_tensor(v, device=self.device), k)) for k, v in draws.items()
                 if k != "noise_std"}
        nc, nf = self.N_coarse, self.N_coarse + self.N_importance
        chunk = min(self.ray_chunk, R) if self.ray_chunk > 0 else R
        lib = _lib.load()
        need = lib.nsr_train_workspace_bytes_for(self._prec, chunk, nc, self.N_importance)   # the buffers of the path this precision takes
        if need == 0:
            raise _lib.NsrError("sample counts outside the built path")
        if self._ws is None or self._ws.numel() < need:
            carried = self.status() if self._ws is not None else 0     # a regrown workspace keeps the sticky flags
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            _lib.check(_lib.load().nsr_train_status_reset(_p(self._ws), _stream()), "nsr_train_status_reset")
            if carried:
                self._ws[:4].copy_(torch.tensor([carried], dtype=torch.int32).view(torch.uint8))
        dev = self.device
        o = {"coarse_comp_rgbs": torch.empty(R, 3, device=dev), "coarse_depth": torch.empty(R, device=dev),
             "coarse_opacity": torch.empty(R, device=dev), "coarse_weights": torch.empty(R, nc, device=dev),
             "fine_comp_rgbs": torch.empty(R, 3, device=dev), "fine_depth": torch.empty(R, device=dev),
             "fine_opacity": torch.empty(R, device=dev), "fine_weights": torch.empty(R, nf, device=dev)}
        lr_c = torch.empty(R // self.s2, 3, device=dev)
        lr_f = torch.empty(R // self.s2, 3, device=dev)
        outs = (c_void_p * 8)(*[c_void_p(o[k].data_ptr()) for k in OUT_KEYS])
        wc, wf = _ptr_array(list(self.params[0].values())), _ptr_array(list(self.params[1].values()))
        gc, gf = _ptr_array(list(self.grads[0].values())), _ptr_array(list(self.grads[1].values()))
        common = (wc, wf, gc, gf, _p(rays), stride, R, self.s2, _p(self.data_rgbs), nc, self.N_importance,
                  int(self.white_bkgd) | (_lib.NSR_TRAIN_GAMMA_CORRECT if self.gamma_correct else 0)
                  | (_lib.NSR_SIGMA_SOFTPLUS if self.sigma_activation == "softplus" else 0)
                  | (_lib.NSR_TRAIN_COLOR_NONE if self.color_activation == "none" else 0)
                  | (_lib.NSR_TRAIN_STOP_GRAD if self.stop_grad else 0), int(self.lindisp),
                  _p(draws.get("u_coarse")), _p(draws.get("u_fine")),
                  _p(draws.get("noise_coarse")), _p(draws.get("noise_fine")), self.noise_std,
                  self.lambda_coarse * gs, self.lambda_fine * gs, self._prec, chunk, outs, _p(lr_c), _p(lr_f), _p(self.losses),
                  _p(self._ws), self._ws.numel(), _stream())
        if any(self.lambda_var):
            import ctypes
            # self.far of the reference (nerf_downX_model.py:284): forward_rays overwrites it per ray_chunk, so what
            # calculate_losses divides by is the far bound of the FIRST ray of the LAST chunk.  Only the depth-variance
            # terms use it: read (one host sync) only when one of them is on; the colour-variance terms never need it.
            far = 1.0
            if any(self.lambda_var[2:]):
                far = float(rays[((R - 1) // chunk) * chunk, 7].item())
            var = (ctypes.c_float * 5)(*[l * gs for l in self.lambda_var], far)       # struct nsr_train_var_losses
            _lib.check(lib.nsr_train_loss_and_grads_var(*common, ctypes.cast(var, c_void_p), _p(self.var_losses)),
                       "nsr_train_loss_and_grads_var")
        else:
            _lib.check(lib.nsr_train_loss_and_grads(*common), "nsr_train_loss_and_grads")
        if self.no_dir:
            for g in self.grads:
                g[DIR_W][:, 256:].zero_()
        if gs != 1.0:
            self.losses.mul_(1.0 / gs)          # report this rank's own losses, not the 1 / world share
            self.var_losses.mul_(1.0 / gs)
        o["lr_coarse"], o["lr_fine"] = lr_c, lr_f
        self.out = o
        return self.losses, self.grads

    def all_reduce_grads(self, group=None):
        """Data-parallel training (replaces DistributedDataParallel's bucketed gradient all-reduce,
        models/networks.py:84): every rank renders its own ray batch with ``grad_scale = 1 / world`` and ONE
        all-reduce (SUM) per network of the flat 2.4 MB gradient buffer (RCCL over xGMI; gloo in the CPU tests of
        ``dist.py``) leaves the global-mean gradient on every rank; the Adam step then runs replicated."""
        from .dist import all_reduce_sum_
        all_reduce_sum_(self.flat_grads, self.group if group is None else group)

    def optimizer_step(self):
        """torch.optim.Adam.step over both networks (:201-204, :408)."""
        self.step += 1
        for n in range(2):
            _lib.check(_lib.load().nsr_adam_step(
                _ptr_array(list(self.params[n].values())), _ptr_array(list(self.grads[n].values())),
                _ptr_array(list(self.exp_avg[n].values())), _ptr_array(list(self.exp_avg_sq[n].values())),
                self.step, self.lr, self.beta1, self.beta2, self.eps, _stream()), "nsr_adam_step")

    def status(self, clear: bool = False) -> int:
        """Sticky numerics status word of the training step (``NSR_FLAG_*``, include/nsr_train.h): 0 = every iteration so
        far packed in-range weights and saw in-range inputs / activations and finite outputs.  Waits for the stream."""
        if self._ws is None:
            return 0
        import ctypes
        flags = ctypes.c_uint(0)
        _lib.check(_lib.load().nsr_train_status(_p(self._ws), int(bool(clear)), ctypes.byref(flags), _stream()), "nsr_train_status")
        return int(flags.value)

    def optimize_parameters(self, draws=None):
        """One training iteration (:398-408); returns the device tensor [coarse_mse, fine_mse] (lambda-weighted).

        ``precision='f16x3'`` carries the forward activations as fp16 (hi, lo) pairs: values beyond 65,504 saturate.
        None of the reference's configurations come near (|h| ~ 1e2), but a diverging run would turn into inf / NaN
        weights silently; ``self.check_finite = True`` adds a host check of the loss (one sync per step) that raises
        and points at ``precision='fp32'`` (the reference itself drops into pdb on NaN, nerf_downX_model.py:273)."""
        self.loss_and_grads(draws)
        if self.check_finite:
            # every rank votes and every rank raises: a rank that raised alone would leave the others waiting in the
            # gradient all-reduce
            from .dist import all_ranks_agree
            flags = self.status(clear=True)
            if not all_ranks_agree(flags == 0, self.device, self.group):
                raise _lib.NsrNumericsError(
                    f"training step {self.step + 1}: numerics status {flags:#x} = {' | '.join(_lib.flag_names(flags)) or 'raised on another rank'}"
                    " -- weights or activations left the split-fp16 operand range; retry with precision='fp32'", flags)
            if not all_ranks_agree(bool(torch.isfinite(self.losses).all()), self.device, self.group):
                raise FloatingPointError(
                    f"non-finite training loss at step {self.step + 1} (this rank: {self.losses.tolist()}): the run diverged"
                    + (" or left the fp16 range of the split-fp16 forward; retry with precision='fp32'"
                       if self.precision != "fp32" else ""))
        self.all_reduce_grads()
        self.optimizer_step()
        return self.losses

    def update_learning_rate(self, epoch: int, lr_policy: str = "exp", n_epochs: int = 20, n_epochs_decay: int = 10,
                             lr_final: float = 5e-6, lr_decay_epochs: int = 10, lr_decay_gamma: float = 0.1,
                             lr_initial: Optional[float] = None) -> float:
        """The reference's per-epoch schedules (models/networks.py:89-118 `get_scheduler`, called from
        base_model.update_learning_rate): 'linear' / 'exp' interpolate (in lr / in log lr) from ``lr`` to ``lr_final``
        over the last ``n_epochs_decay`` of ``n_epochs`` epochs, 'step' multiplies by gamma every ``lr_decay_epochs``.
        ``epoch`` is the number of finished epochs (the scheduler's ``last_epoch``); returns and installs the new lr."""
        import math
        lr0 = self._lr_initial = lr_initial if lr_initial is not None else getattr(self, "_lr_initial", self.lr)
        if lr_policy in ("linear", "exp"):
            t = max(0, epoch + 1 - n_epochs + n_epochs_decay) / float(n_epochs_decay + 1)
            self.lr = (lr0 * (1 - t) + lr_final * t) if lr_policy == "linear" else \
                math.exp(math.log(lr0) * (1 - t) + math.log(lr_final) * t)
        elif lr_policy == "step":
            self.lr = lr0 * lr_decay_gamma ** (epoch // lr_decay_epochs)
        else:
            raise NotImplementedError(f"learning rate policy [{lr_policy}] is not implemented")
        return self.lr

    def state_dicts(self):
        return [{k: (v[:, :256] if (self.no_dir and k == DIR_W) else v).clone() for k, v in p.items()} for p in self.params]
