"""The step tail -- gradient clipping by global norm, then Adam -- as two launches over ALL parameter tensors
(csrc/optim.hip) instead of torch's chain (per-tensor norms, stack, norm, clamp, a multiply per tensor, the multi-tensor
Adam): the reference's ``Trainer(gradient_clip_val=1.0)`` + ``torch.optim.Adam`` (train_gcl.py:63-68, train_kge.py:44-50,
gcl_module.py:78-87, kge_module.py:208-217).

``clip_adam_step(optimizer, max_norm)`` works ON a ``torch.optim.Adam`` object: it reads the hyper-parameters from its
single param group and keeps the state in the optimizer's own ``state`` dict under torch's keys (``step``, ``exp_avg``,
``exp_avg_sq``), so ``state_dict()`` / ``load_state_dict()`` / LR schedulers / checkpoints are torch's.  Differences to
``clip_grad_norm_`` + ``step()``: the gradients are not scaled in place (the clip factor is applied inside the update),
and the squared norm is summed in a fixed block order (bitwise repeatable).  Configurations it does not cover (several
param groups, amsgrad, weight decay, maximize, non-fp32 or non-contiguous tensors, more than 48 tensors, an optimizer
with step hooks registered -- the fused path never calls ``step()``, so hooks would not run) take torch's path.  A capturable optimizer (device ``step`` / ``lr``: hipGraph-captured steps, biomedkg_amd/graphs.py) is updated
through its device scalars."""
from __future__ import annotations

import ctypes

import importlib

import torch

from . import _lib as L

_torch_optimizer = importlib.import_module("torch.optim.optimizer")     # (torch.optim deletes the attribute after import)
_SCRATCH = {}


def _supported(opt, params):
    if not isinstance(opt, torch.optim.Adam) or len(opt.param_groups) != 1:
        return False
    g = opt.param_groups[0]
    if g.get("amsgrad") or g.get("weight_decay") or g.get("maximize") or g.get("differentiable"):
        return False
    if not params or len(params) > int(L.lib().bkg_clip_adam_max_tensors()):
        return False
    return all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad is not None
               and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and not p.grad.is_sparse for p in params)


def clip_adam_step(opt, max_norm=None):
    """``clip_grad_norm_(params, max_norm)`` (skipped when ``max_norm`` is falsy) + ``opt.step()``.  Returns the total
    gradient norm as a device scalar when the fused path ran, else whatever torch's clip returns (or None)."""
    group = opt.param_groups[0]
    params = [p for p in group["params"] if p.grad is not None]
    # step hooks (optimizer.register_step_pre_hook / _post_hook, the global ones, a profiler's): this path never calls
    # opt.step(), so an optimizer that carries any takes torch's path, where they run
    hooked = bool(getattr(opt, "_optimizer_step_pre_hooks", None) or getattr(opt, "_optimizer_step_post_hooks", None)
                  or getattr(_torch_optimizer, "_global_optimizer_pre_hooks", None)
                  or getattr(_torch_optimizer, "_global_optimizer_post_hooks", None))
    if hooked or not _supported(opt, params):
        norm = torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"]], max_norm) if max_norm else None
        opt.step()
        return norm
    # what LRScheduler's patch of ``step`` records: schedulers check it to warn about "scheduler before optimizer"
    opt._opt_called = True
    dev = params[0].device
    capturable = bool(group.get("capturable"))
    for p in params:
        st = opt.state[p]
        if len(st) == 0:                                   # torch.optim.Adam._init_group
            st["step"] = (torch.zeros((), dtype=torch.float32, device=dev) if (capturable or group.get("fused"))
                          else torch.tensor(0.0, dtype=torch.float32))      # (torch keeps a fused optimizer's counter on the device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
    steps = [opt.state[p]["step"] for p in params]
    lr = group["lr"]
    lr_dev = lr if isinstance(lr, torch.Tensor) and lr.is_cuda else None
    step_dev = None
    if steps[0].is_cuda:
        # torch keeps one counter per parameter; they move in lockstep: the first one is the kernel's, the others follow
        step_dev = steps[0]
        step_host = 0.0
    else:
        for s in steps:
            s += 1
        step_host = float(steps[0])
    count = len(params)
    arr = ctypes.c_void_p * count
    numel = (ctypes.c_int64 * count)(*[p.numel() for p in params])
    chunks = int(L.lib().bkg_clip_adam_chunks(numel, count))
    key = (str(dev), torch.cuda.current_stream().cuda_stream if torch.cuda.is_current_stream_capturing() else 0)
    scratch = _SCRATCH.get(key)
    if scratch is None or scratch.numel() < chunks + 1:
        scratch = _SCRATCH[key] = torch.empty(max(chunks + 1, 4096), dtype=torch.float32, device=dev)
    norm = torch.empty((), dtype=torch.float32, device=dev)
    b1, b2 = group["betas"]
    # weights with bf16 images (ops._weight_bf16: the operands of the bf16-mode products): the update rewrites the images
    # too, so the next step finds them current instead of launching one conversion per weight
    images = [_image_of(p) for p in params] if FUSE_WEIGHT_IMAGES else [None] * count
    if any(im is not None for im in images):
        i32 = ctypes.c_int32 * count
        L.call("bkg_clip_adam_images_f32", count, arr(*[p.data_ptr() for p in params]),
               arr(*[p.grad.data_ptr() for p in params]), arr(*[opt.state[p]["exp_avg"].data_ptr() for p in params]),
               arr(*[opt.state[p]["exp_avg_sq"].data_ptr() for p in params]), numel,
               arr(*[im[0].data_ptr() if im else None for im in images]), arr(*[im[1].data_ptr() if im else None for im in images]),
               i32(*[im[2] if im else 1 for im in images]), i32(*[im[3] if im else 1 for im in images]),
               float(max_norm or 0.0), 0.0 if lr_dev is not None else float(lr), L.ptr(lr_dev), float(b1), float(b2),
               float(group["eps"]), step_host, L.ptr(step_dev), L.ptr(scratch), scratch.numel(), L.ptr(norm), L.stream())
    else:
        L.call("bkg_clip_adam_f32", count, arr(*[p.data_ptr() for p in params]), arr(*[p.grad.data_ptr() for p in params]),
               arr(*[opt.state[p]["exp_avg"].data_ptr() for p in params]),
               arr(*[opt.state[p]["exp_avg_sq"].data_ptr() for p in params]), numel, float(max_norm or 0.0),
               0.0 if lr_dev is not None else float(lr), L.ptr(lr_dev), float(b1), float(b2), float(group["eps"]), step_host,
               L.ptr(step_dev), L.ptr(scratch), scratch.numel(), L.ptr(norm), L.stream())
    if step_dev is not None and len(steps) > 1:            # (one tiny multi-tensor copy; captured with the step)
        torch._foreach_copy_(steps[1:], [step_dev] * (len(steps) - 1))
    # the kernels wrote through raw pointers: tell autograd's version counters (everything cached per weight ``_version``
    # -- the bf16 weight images of ops._weight_bf16 -- must see the update, exactly as after an in-place torch op)
    for p in params:
        torch.autograd.graph.increment_version(p)
    if any(im is not None for im in images):
        from . import ops
        for p, im in zip(params, images):
            if im is not None:                 # the images are those of the NEW version
                p._bkg_w16 = (ops.weight_image_key(p), im[0], im[1])
    return norm


FUSE_WEIGHT_IMAGES = __import__("os").environ.get("BKG_FUSE_WEIGHT_IMAGES", "1") != "0"


def _image_of(p):
    """(plain, transposed, rows, cols) of a parameter's cached bf16 images, if it has them and they can be rewritten in
    place: created by ``ops._weight_bf16`` for this very storage and shape, in bf16 mode."""
    cache = getattr(p, "_bkg_w16", None)
    if cache is None or p.dim() not in (2, 3):
        return None
    from . import ops
    if not ops.bf16_storage():
        return None
    plain, trans = cache[1], cache[2]
    if (cache[0][0] != p.data_ptr() or plain.shape != p.shape or plain.device != p.device or not plain.is_contiguous()
            or not trans.is_contiguous()):
        return None
    return plain, trans, int(p.shape[-2]), int(p.shape[-1])
