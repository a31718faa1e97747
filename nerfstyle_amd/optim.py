"""Fused optimiser for the parameter arena: Adam (trainers/base.py:216-221: lr, betas (0.9, 0.999),
eps 1e-15) + GradScaler unscale + EMA shadow (utils/__init__.py:116-142) + gradient zero-fill +
f16 gather-table refresh in ONE streaming pass per region (nsr_adam_step), instead of
torch.optim.Adam's elementwise launches, zero_grad, torch_ema and the per-forward
`embeddings.to(torch.half)` cast of the reference.

Keyword selection mirrors Trainer._reset_optim (base.py:185-214): `keywords=None` trains
everything, `['x_color_embedder']` (StyleTrainer.OPTIM_KEYS, style.py:25) trains the colour table
only -- expressed as an element mask on the interleaved tables."""
import torch

from . import _lib as L
from .style_nerf import MLP_LAYOUT, StyleTCNerf


def select_regions(model: StyleTCNerf, keywords=None):
    """-> (table_mask4, [(offset, length) of trained MLP blocks, relative to the MLP block])"""
    def on(name):
        return keywords is None or any(kw in name for kw in keywords)
    mask = (0x3 if on('x_density_embedder.embeddings') else 0) | (0xC if on('x_color_embedder.embeddings') else 0)
    nets = [(off, n) for (name, off, n) in MLP_LAYOUT if on(name + '.params')]
    if mask == 0 and not nets:
        raise ValueError('Keywords {} not found in parameter names'.format(keywords))
    return mask, nets


class FusedAdam:
    def __init__(self, model: StyleTCNerf, lr=1e-2, betas=(0.9, 0.999), eps=1e-15, keywords=None, ema_decay=None):
        self.model = model
        self.table_mask, self.nets = select_regions(model, keywords)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.base_lr = lr
        self.step_count = 0
        a = model.arena.detach()
        self.exp_avg = torch.zeros_like(a)
        self.exp_avg_sq = torch.zeros_like(a)
        self.ema_decay = ema_decay
        self.ema = a.clone() if ema_decay is not None else None
        self.ema_updates = 0
        model._ensure_grad()
        model.train_density_table = bool(self.table_mask & 0x3)
        model.train_color_table = bool(self.table_mask & 0xC)
        # torch.optim-like surface for LR schedulers
        self.param_groups = [{'lr': lr, 'initial_lr': lr, 'params': [model.arena]}]

    def zero_grad(self, set_to_none=False):
        self.model._ensure_grad().zero_()

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        """One update; gradients are zeroed on the way out.  grad_scale: the loss scale the
        gradients carry (GradScaler): they are multiplied by 1/grad_scale inside the kernel."""
        m = self.model
        g = m._ensure_grad()
        self.step_count += 1
        lr = self.param_groups[0]['lr']
        decay = 0.0
        if self.ema is not None:
            # torch_ema: decay = min(decay, (1 + n) / (10 + n)) with n counted before the update
            self.ema_updates += 1
            decay = min(self.ema_decay, (1 + self.ema_updates) / (10 + self.ema_updates))
        a = m.arena.detach()
        half = m.half_tables() if m.table_dtype == torch.float16 else None

        def run(off, n, mask, half_ptr):
            ptr = lambda t: t.data_ptr() + off * 4
            L.check(L.lib().nsr_adam_step(
                ptr(a), ptr(g), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.ema) if self.ema is not None else None,
                half_ptr, n, float(lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                float(1.0 / grad_scale), float(decay), self.step_count, mask, L.stream()), 'adam_step')

        if self.table_mask:
            run(0, m.table_elems, self.table_mask, half.data_ptr() if half is not None else None)
        else:
            g[:m.table_elems].zero_()
        trained = set()
        for (off, n) in self.nets:
            run(m.table_elems + off, n, 0xF, None)
            trained.add(off)
        for (name, off, n) in MLP_LAYOUT:
            if off not in trained:
                g[m.table_elems + off: m.table_elems + off + n].zero_()
        if half is not None and self.table_mask:
            m.mark_half_synced()

    def state_dict(self):
        return {'step': self.step_count, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq, 'ema': self.ema,
                'ema_updates': self.ema_updates, 'lr': self.param_groups[0]['lr']}

    def load_state_dict(self, sd):
        self.step_count = sd['step']
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        if self.ema is not None and sd.get('ema') is not None:
            self.ema.copy_(sd['ema'])
        self.ema_updates = sd.get('ema_updates', 0)
        self.param_groups[0]['lr'] = sd.get('lr', self.lr)


def exp_lr(initial_lr, it, decay_steps):
    """trainers/base.py:223-227: lr = initial * 0.1 ** (it / learning_rate_decay)"""
    return initial_lr * (0.1 ** (it / decay_steps)) if decay_steps > 0 else initial_lr


class LossScaler:
    """Dynamic loss scaling with torch.cuda.amp.GradScaler's policy (trainers/base.py:228,420-425:
    init 65536, x2 after 2000 clean steps, x0.5 and skip the step on inf/nan), for the f16 MFMA path:
    with f16 operands the gradients of a mean-over-rays loss underflow unscaled (the reference trains
    tcnn's fp16 networks under GradScaler for the same reason).  bf16 compute needs no scaling.

        loss = scaler.scale(loss); loss.backward(); stepped = scaler.step(opt)

    The inf/nan check is one reduction over the flat gradient arena and one host read per step."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.scale_value = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.enabled = enabled
        self._good_steps = 0

    def scale(self, loss):
        return loss * self.scale_value

    def get_scale(self):
        return self.scale_value

    def state_dict(self):
        """torch.cuda.amp.GradScaler.state_dict's keys (trainers/base.py:28, SD_SAVE_KEYS 'scaler')"""
        return {'scale': float(self.scale_value), 'growth_factor': self.growth_factor, 'backoff_factor': self.backoff_factor,
                'growth_interval': self.growth_interval, '_growth_tracker': int(self._good_steps)}

    def load_state_dict(self, sd):
        self.scale_value = float(sd['scale'])
        self.growth_factor, self.backoff_factor = sd['growth_factor'], sd['backoff_factor']
        self.growth_interval = sd['growth_interval']
        self._good_steps = int(sd.get('_growth_tracker', 0))

    def step(self, opt: FusedAdam) -> bool:
        if not self.enabled:
            opt.step()
            return True
        g = opt.model._ensure_grad()
        finite = bool(torch.isfinite(g.abs().max()).item())
        if finite:
            opt.step(grad_scale=self.scale_value)
            self._good_steps += 1
            if self._good_steps >= self.growth_interval:
                self.scale_value *= self.growth_factor
                self._good_steps = 0
            return True
        g.zero_()
        self.scale_value *= self.backoff_factor
        self._good_steps = 0
        return False
