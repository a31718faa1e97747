"""Reconstruction-stage loss of the reference's Trainer.calc_loss (trainers/base.py:251-304) as one fused HIP op:

    loss = MSE(rgb_map, target_rgb[pix]) + ce_lambda * CrossEntropy(classes, target_cls[pix])        (ce_lambda = 0.001, :281)

value and gradient in ONE pass over the rays (nsr_recon_loss), the target gather fused, optionally multiplied by the
device-side loss scale (optim.LossScaler) and a host factor (1 / world size).  The reference -- and rounds 1-2 of this build --
spend ~50 small torch kernels on it per step (slices, subtraction, square, mean, logsumexp, gather and their autograd
twins: 1.7 ms of a 37 ms full-frame step, a third of a 4 096-ray step).  Any other loss can still be written in torch on the
renderer's outputs; this op is the hot-path form of the reference's own loss."""
import torch

from . import _lib as L


class _recon_loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_map, classes, target_rgb, target_cls, pix, ce_lambda, factor, scale):
        ctx.set_materialize_grads(False)
        N = rgb_map.shape[0]
        dev = rgb_map.device
        rgb_map = rgb_map.detach().to(torch.float32).contiguous()
        with_ce = classes is not None and classes.shape[1] > 0 and ce_lambda != 0.0
        cls = classes.detach().to(torch.float32).contiguous() if with_ce else None
        nc = cls.shape[1] if with_ce else 0
        g_rgb = torch.empty_like(rgb_map)
        g_cls = torch.empty_like(cls) if with_ce else None
        out = torch.empty(3, dtype=torch.float32, device=dev)
        ws = torch.empty(int(L.lib().nsr_recon_loss_workspace_bytes(N)) // 4, dtype=torch.float32, device=dev)
        assert target_rgb.dtype == torch.float32 and target_rgb.is_contiguous()
        if with_ce:
            assert target_cls.dtype == torch.int64 and target_cls.is_contiguous()
        if pix is not None:
            assert pix.dtype == torch.int64 and pix.is_contiguous() and pix.numel() == N
        else:
            assert target_rgb.shape[0] == N
        L.check(L.lib().nsr_recon_loss(L.p(rgb_map), L.p(cls), N, nc, L.p(target_rgb), L.p(target_cls) if with_ce else None, L.p(pix),
                                       float(ce_lambda), float(factor), L.p(scale), L.p(g_rgb), L.p(g_cls), L.p(out), L.p(ws),
                                       L.stream()), 'recon_loss')
        ctx.save_for_backward(g_rgb, g_cls)
        ctx.terms = out
        return out[0]

    @staticmethod
    def backward(ctx, go):
        g_rgb, g_cls = ctx.saved_tensors
        if go is None:
            return (None,) * 8
        # the stored gradients already carry scale * factor; `go` is 1 for loss.backward()
        return g_rgb * go, (g_cls * go if g_cls is not None else None), None, None, None, None, None, None


def recon_loss(rgb_map, classes, target_rgb, target_cls=None, pix=None, ce_lambda=1e-3, factor=1.0, scale=None, backward=False):
    """-> 0-dim loss tensor = factor * scale * (mse + ce_lambda * ce).  rgb_map [N,3], classes [N,nc] (or None) from
    Renderer.render(training=True); target_rgb [P,3] f32 / target_cls [P] int64 resident on the device, pix [N] int64 their rows
    (None: P == N, row n).  scale: 0-dim float32 device tensor (LossScaler.scale_tensor) or None.
    backward=True: the gradient the kernel has just written is back-propagated from here -- autograd.backward(outputs of the
    renderer, d loss / d outputs) -- and the returned loss is detached: `loss.backward()` on the plain form costs a ones-fill and
    one multiply per stored gradient (three launches, 1.5 % of a 4 096-ray captured step) to apply a factor that is 1."""
    loss = _recon_loss.apply(rgb_map, classes, target_rgb, target_cls, pix, ce_lambda, factor, scale)
    if not backward or loss.grad_fn is None:
        return loss
    g_rgb, g_cls = loss.grad_fn.saved_tensors
    outs, grads = [rgb_map], [g_rgb]
    if g_cls is not None and classes is not None and classes.requires_grad:
        outs.append(classes)
        grads.append(g_cls)
    if not rgb_map.requires_grad:
        outs, grads = outs[1:], grads[1:]
    if outs:
        torch.autograd.backward(outs, grads)
    det = loss.detach()
    det._nsr_terms = loss.grad_fn.terms
    return det


def last_terms(loss):
    """(mse, ce_lambda * ce) of a loss returned by recon_loss, unscaled, as a device tensor [2] (no host sync)."""
    if hasattr(loss, '_nsr_terms'):
        return loss._nsr_terms[1:]
    fn = loss.grad_fn
    return fn.terms[1:] if fn is not None and hasattr(fn, 'terms') else None
