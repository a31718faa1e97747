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
        model.train_mlps = bool(self.nets)
        # torch.optim-like surface for LR schedulers
        self.param_groups = [{'lr': lr, 'initial_lr': lr, 'params': [model.arena]}]

    def zero_grad(self, set_to_none=False):
        self.model._ensure_grad().zero_()

    def _regions(self):
        """[(offset, n, elem_mask4, half_n)] of the arena the optimiser trains, merged where contiguous: everything trained
        (the reconstruction stage) is ONE launch over the whole arena, colour table only (stylisation) is one over the tables."""
        m = self.model
        all_nets = len(self.nets) == len(MLP_LAYOUT)
        if self.table_mask == 0xF and all_nets:
            return [(0, m.arena.numel(), 0xF, m.table_elems)]
        out = []
        if self.table_mask:
            out.append((0, m.table_elems, self.table_mask, m.table_elems))
        if all_nets:
            out.append((m.table_elems, m.arena.numel() - m.table_elems, 0xF, 0))
        else:
            out += [(m.table_elems + off, n, 0xF, 0) for (off, n) in self.nets]
        return out

    def _zero_untrained(self, g):
        m = self.model
        if not self.table_mask:
            g[:m.table_elems].zero_()
        trained = {off for (off, n) in self.nets}
        for (name, off, n) in MLP_LAYOUT:
            if off not in trained:
                g[m.table_elems + off: m.table_elems + off + n].zero_()

    def _ema_decay_now(self):
        if self.ema is None:
            return 0.0
        # torch_ema: decay = min(decay, (1 + n) / (10 + n)) with n counted before the update
        self.ema_updates += 1
        return min(self.ema_decay, (1 + self.ema_updates) / (10 + self.ema_updates))

    @torch.no_grad()
    def step(self, grad_scale=1.0, scaler=None, lr_decay_steps=0.0):
        """One update; gradients are zeroed on the way out.
        scaler=None: host-side scalars -- grad_scale is the loss scale the gradients carry (multiplied by 1/grad_scale inside
        the kernel), lr = param_groups[0]['lr'], the step count lives here.
        scaler=LossScaler: GradScaler semantics ON THE DEVICE (nsr_grad_check -> nsr_scaler_update -> nsr_adam_step_scaled):
        inf/nan check over the trained regions, skip + back-off or step + growth, the step count, lr = param_groups[0]
        ['initial_lr'] * 0.1^(steps / lr_decay_steps) (trainers/base.py:223-226: LambdaLR, advanced only on steps that
        were not skipped) and the bias corrections all stay in `scaler.state`; nothing is read back."""
        m = self.model
        g = m._ensure_grad()
        a = m.arena.detach()
        half = m.half_tables() if m.table_dtype == torch.float16 else None
        ptr = lambda t, off: t.data_ptr() + off * 4
        regions = self._regions()
        if scaler is None:
            self.step_count += 1
            decay = self._ema_decay_now()
            lr = self.param_groups[0]['lr']
            for (off, n, mask, half_n) in regions:
                # the host-scalar entry point has no half_n: a whole-arena region is split at the table end
                parts = [(off, n)] if half_n in (0, n) or half is None else [(off, half_n), (off + half_n, n - half_n)]
                for (o, k) in parts:
                    L.check(L.lib().nsr_adam_step(
                        ptr(a, o), ptr(g, o), ptr(self.exp_avg, o), ptr(self.exp_avg_sq, o),
                        ptr(self.ema, o) if self.ema is not None else None,
                        half.data_ptr() if (half is not None and o == 0 and half_n) else None, k, float(lr), float(self.betas[0]),
                        float(self.betas[1]), float(self.eps), float(1.0 / grad_scale), float(decay), self.step_count, mask,
                        L.stream()), 'adam_step')
        else:
            if getattr(self, '_scaler', None) is not scaler:
                # first step with this scaler: the EMA update count moves to the device (torch_ema's decay schedule follows it)
                scaler.adopt_ema_updates(self.ema_updates, a.device)
                self._scaler = scaler
            st = scaler.state_on(a.device)
            for (off, n, mask, half_n) in regions:
                L.check(L.lib().nsr_grad_check(ptr(g, off), n, mask, L.p(st), L.stream()), 'grad_check')
            L.check(L.lib().nsr_scaler_update(L.p(st), float(self.param_groups[0]['initial_lr']), float(lr_decay_steps),
                                              float(self.betas[0]), float(self.betas[1]), float(scaler.growth_factor),
                                              float(scaler.backoff_factor), int(scaler.growth_interval), int(scaler.enabled),
                                              float(self.ema_decay) if self.ema is not None else -1.0, L.stream()), 'scaler_update')
            for (off, n, mask, half_n) in regions:
                L.check(L.lib().nsr_adam_step_scaled(
                    ptr(a, off), ptr(g, off), ptr(self.exp_avg, off), ptr(self.exp_avg_sq, off),
                    ptr(self.ema, off) if self.ema is not None else None,
                    half.data_ptr() if (half is not None and off == 0 and half_n) else None, n, half_n if half is not None else 0,
                    float(self.betas[0]), float(self.betas[1]), float(self.eps), mask, L.p(st), L.stream()),
                    'adam_step_scaled')
        self._zero_untrained(g)
        if half is not None and self.table_mask:
            m.mark_half_synced()

    @property
    def steps_taken(self):
        """Optimiser steps really taken (host read when a device-side scaler keeps the count)."""
        sc = getattr(self, '_scaler', None)
        return int(sc.state[3].item()) if sc is not None and sc.state is not None else self.step_count

    @property
    def ema_updates_made(self):
        sc = getattr(self, '_scaler', None)
        return int(sc.state[5].item()) if sc is not None and sc.state is not None else self.ema_updates

    def state_dict(self):
        return {'step': self.steps_taken, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq, 'ema': self.ema,
                'ema_updates': self.ema_updates_made, 'lr': self.param_groups[0]['lr']}

    def load_reference_state(self, optim_sd, ema_sd=None):
        """State of the REFERENCE's optimiser objects: torch.optim.Adam.state_dict() ({'state': {i: {'step', 'exp_avg',
        'exp_avg_sq'}}, 'param_groups': ...}, parameters in named_parameters() order filtered by the trainer's keywords,
        trainers/base.py:185-221) and torch_ema's state_dict ({'decay', 'num_updates', 'shadow_params': [...]}, over ALL
        model parameters, base.py:229) -> the flat arenas kept here."""
        m = self.model
        trained = set(off for off, _ in self.nets)
        names = []
        if self.table_mask & 0x3:
            names.append('x_density_embedder.embeddings')
        if self.table_mask & 0xC:
            names.append('x_color_embedder.embeddings')
        names += [name + '.params' for (name, off, n) in MLP_LAYOUT if off in trained]
        state = optim_sd['state']
        keys = sorted(state.keys(), key=int)
        if len(keys) != len(names):
            raise RuntimeError('optimiser state holds {} parameters, this optimiser trains {} ({})'.format(len(keys), len(names), names))

        def views(flat):
            t = flat[:m.table_elems].view(m.rows, 2, 2)
            out = {'x_density_embedder.embeddings': t[:, 0, :], 'x_color_embedder.embeddings': t[:, 1, :]}
            for (name, off, n) in MLP_LAYOUT:
                out[name + '.params'] = flat[m.table_elems + off: m.table_elems + off + n]
            return out
        va, vs = views(self.exp_avg), views(self.exp_avg_sq)
        steps = set()
        for k, name in zip(keys, names):
            va[name].copy_(state[k]['exp_avg'].reshape(va[name].shape))
            vs[name].copy_(state[k]['exp_avg_sq'].reshape(vs[name].shape))
            steps.add(int(state[k]['step']))
        self.step_count = max(steps) if steps else 0
        self.param_groups[0]['lr'] = optim_sd['param_groups'][0].get('lr', self.lr)
        self.param_groups[0]['initial_lr'] = optim_sd['param_groups'][0].get('initial_lr', self.param_groups[0]['initial_lr'])
        if ema_sd is not None and self.ema is not None and ema_sd.get('shadow_params') is not None:
            ve = views(self.ema)
            order = ['x_density_embedder.embeddings', 'x_color_embedder.embeddings'] + [name + '.params' for (name, _, _) in MLP_LAYOUT]
            for name, t in zip(order, ema_sd['shadow_params']):
                ve[name].copy_(t.reshape(ve[name].shape))
            self.ema_updates = int(ema_sd.get('num_updates') or 0)

    def load_state_dict(self, sd):
        if 'param_groups' in sd and 'state' in sd:
            return self.load_reference_state(sd)
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
    """torch.cuda.amp.GradScaler's policy (trainers/base.py:228,420-425: init 65536, x2 after 2000 clean steps, x0.5 and skip
    the step on inf/nan) with every piece of state ON THE DEVICE (`state`: int32[16], layout in include/nsr.h): the scale the
    loss is multiplied by is a device scalar, the inf/nan check is one streaming pass over the trained gradient regions, the
    skip decision is taken by the optimiser kernel itself.  No `.item()`, legal under hipGraph capture.  With f16 operands
    the gradients of a mean-over-rays loss underflow unscaled (the reference trains tcnn's fp16 networks under GradScaler for
    the same reason); bf16 compute needs no scaling (enabled=False: scale 1, never skips).

        loss = scaler.scale(loss); loss.backward(); scaler.step(opt)          # opt: FusedAdam
    """

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.init_scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.enabled = enabled
        self.state = None
        # host-side values, uploaded when the device is known (and the source of state_dict() until then)
        self._pending = {'scale': self.init_scale, '_growth_tracker': 0, 'steps': 0, 'skipped': 0, 'ema_updates': 0}

    def state_on(self, device):
        if self.state is None or self.state.device != device:
            host = torch.zeros(16, dtype=torch.int32)
            vals = self._pending if self._pending is not None else self._read_back()
            host[0:1].view(torch.float32)[0] = float(vals['scale'])
            host[1], host[3], host[4] = int(vals['_growth_tracker']), int(vals['steps']), int(vals['skipped'])
            host[5] = int(vals.get('ema_updates', 0))
            self.state = host.to(device)
            self._pending = None
        return self.state

    def _read_back(self):
        st = self.state.cpu()
        return {'scale': float(st[0:1].view(torch.float32)[0]), '_growth_tracker': int(st[1]), 'steps': int(st[3]), 'skipped': int(st[4]),
                'ema_updates': int(st[5])}

    def adopt_ema_updates(self, n, device):
        """The optimiser's host-side EMA update count becomes the device-side one (first scaled step / after a restore)."""
        if self.state is None:
            self._pending['ema_updates'] = max(int(n), int(self._pending.get('ema_updates', 0)))
            self.state_on(device)
        else:
            self.state[5:6].clamp_(min=int(n))                  # device op: no host read

    def scale_tensor(self, device):
        """0-dim float32 view of the device-side scale (what `scale(loss)` multiplies by)."""
        return self.state_on(device)[0:1].view(torch.float32)[0]

    def scale(self, loss):
        return loss * self.scale_tensor(loss.device)

    def get_scale(self):
        """Host read (diagnostics / checkpoints only)."""
        return float(self.state[0:1].view(torch.float32)[0].item()) if self.state is not None else float(self._pending['scale'])

    def steps_skipped(self):
        return int(self.state[4].item()) if self.state is not None else 0

    def state_dict(self):
        """torch.cuda.amp.GradScaler.state_dict's keys (trainers/base.py:28, SD_SAVE_KEYS 'scaler') + the step counters"""
        v = self._read_back() if self.state is not None else dict(self._pending)
        v.update({'growth_factor': self.growth_factor, 'backoff_factor': self.backoff_factor, 'growth_interval': self.growth_interval})
        return v

    def load_state_dict(self, sd):
        self.growth_factor, self.backoff_factor = sd['growth_factor'], sd['backoff_factor']
        self.growth_interval = sd['growth_interval']
        self._pending = {'scale': float(sd['scale']), '_growth_tracker': int(sd.get('_growth_tracker', 0)),
                         'steps': int(sd.get('steps', 0)), 'skipped': int(sd.get('skipped', 0)),
                         'ema_updates': int(sd.get('ema_updates', 0))}
        dev = self.state.device if self.state is not None else None
        self.state = None
        if dev is not None:
            self.state_on(dev)

    def step(self, opt: FusedAdam, lr_decay_steps=0.0) -> None:
        opt.step(scaler=self, lr_decay_steps=lr_decay_steps)
