"""Checkpoints with the reference's top-level layout (trainers/base.py:26-28,231-249; render.py:36-87):

    {'version', 'log_dir', 'iter_ctr', 'cfg', 'dataset_cfg', 'train_cfg', 'net_cfg', 'render_cfg',     # SAVE_KEYS
     'renderer', 'optim', 'scheduler', 'scaler', 'ema'}                                                 # SD_SAVE_KEYS

The reference pickles its config dataclasses, a pathlib.Path and (inside `renderer`) an Intrinsics object
by class reference.  Files written here hold the same keys but only tensors and plain containers
(dataclasses become `{'__dataclass__': name, ...fields}` dicts, paths become strings), so they load with
`torch.load(weights_only=True)` -- nothing from the file is executed.  A file written by the reference
itself names its config dataclasses, `common.Intrinsics` and `pathlib.PosixPath`: `load_checkpoint` reads it through the
inert stand-ins of reference_schema.py on the weights-only allow-list (still nothing is executed); anything else is refused.  `renderer['model']` uses the reference's parameter
names; the MLP `*.params` vectors are in the layout of nerfstyle_amd/network.py (tinycudann's internal
layout is not part of the reference tree: "parity unpinned")."""
import dataclasses
import os
import pickle
from pathlib import Path
from typing import Any, Dict, Optional

import torch

from .common import Intrinsics

SAVE_KEYS = ['version', 'log_dir', 'iter_ctr', 'cfg', 'dataset_cfg', 'train_cfg', 'net_cfg', 'render_cfg']
SD_SAVE_KEYS = ['renderer', 'optim', 'scheduler', 'scaler', 'ema']
FORMAT_VERSION = 'nerfstyle_amd-1'


def to_plain(v: Any) -> Any:
    """tensors stay (detached, on the CPU); dataclasses, paths and containers become plain data"""
    if torch.is_tensor(v):
        return v.detach().cpu()
    if dataclasses.is_dataclass(v) and not isinstance(v, type):
        d = {'__dataclass__': type(v).__name__}
        for f in dataclasses.fields(v):
            d[f.name] = to_plain(getattr(v, f.name))
        return d
    if isinstance(v, Path):
        return str(v)
    if isinstance(v, dict):
        return {str(k): to_plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [to_plain(x) for x in v]
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    if hasattr(v, 'item') and getattr(v, 'shape', None) == ():      # numpy scalars
        return v.item()
    raise TypeError('cannot store a {} in a checkpoint'.format(type(v)))


def intrinsics_from_plain(d) -> Intrinsics:
    if isinstance(d, Intrinsics):
        return d
    return Intrinsics(d['h'], d['w'], d['fx'], d['fy'], d['cx'], d['cy'])


def save_checkpoint(path, renderer, optim=None, scaler=None, iter_ctr: int = 0, log_dir: str = '', cfg=None,
                    dataset_cfg=None, train_cfg=None, net_cfg=None, render_cfg=None) -> None:
    """trainers/base.py:231-249.  `optim` is a FusedAdam (it carries the EMA the reference keeps in `ema`)."""
    sd: Dict[str, Any] = {
        'version': FORMAT_VERSION, 'log_dir': str(log_dir), 'iter_ctr': int(iter_ctr),
        'cfg': to_plain(cfg), 'dataset_cfg': to_plain(dataset_cfg), 'train_cfg': to_plain(train_cfg),
        'net_cfg': to_plain(net_cfg if net_cfg is not None else getattr(renderer.model, 'cfg', None)),
        'render_cfg': to_plain(render_cfg if render_cfg is not None else renderer.cfg),
        'renderer': to_plain(renderer.state_dict()),
        'optim': to_plain(optim.state_dict()) if optim is not None else None,
        'scheduler': None,                       # the learning rate is a closed form of iter_ctr (optim.exp_lr)
        'scaler': to_plain(scaler.state_dict()) if scaler is not None and hasattr(scaler, 'state_dict') else None,
        'ema': {'shadow': optim.ema.detach().cpu(), 'num_updates': optim.ema_updates}
               if optim is not None and getattr(optim, 'ema', None) is not None else None,
    }
    tmp = str(path) + '.tmp'
    torch.save(sd, tmp)
    os.replace(tmp, str(path))


def load_checkpoint(path, map_location='cpu') -> Dict[str, Any]:
    """Safe load (weights_only).  A file of this build loads as it is; a file written by the reference names its config
    dataclasses, common.Intrinsics and pathlib.PosixPath and is read through the inert stand-ins of reference_schema.py on
    the weights-only allow-list.  Anything else raises RuntimeError; nothing from a checkpoint file is ever executed."""
    try:
        sd = torch.load(str(path), map_location=map_location, weights_only=True)
    except pickle.UnpicklingError:
        from .reference_schema import load_reference_checkpoint
        try:
            sd = load_reference_checkpoint(path, map_location)
        except (pickle.UnpicklingError, RuntimeError) as e:
            raise RuntimeError(
                "{} holds pickled Python objects outside the reference's checkpoint schema (config dataclasses, "
                "common.Intrinsics, pathlib.PosixPath: those are read through inert stand-ins): it is not loaded, nothing from a "
                "checkpoint file is executed.  Loader said: {}".format(path, e)) from None
    missing = [k for k in SAVE_KEYS + SD_SAVE_KEYS if k not in sd]
    if missing:
        raise RuntimeError('{} is not a trainer checkpoint: missing keys {}'.format(path, missing))
    return sd


def restore(sd: Dict[str, Any], renderer, optim=None, scaler=None) -> int:
    """render.py:84-85 / trainers/base.py:160-168: state into an already constructed renderer (+ optimiser);
    returns iter_ctr."""
    rs = dict(sd['renderer'])
    rs['intr'] = intrinsics_from_plain(rs['intr'])
    renderer.load_state_dict(rs)
    if optim is not None and sd.get('optim') is not None:
        if 'param_groups' in sd['optim'] and 'state' in sd['optim']:
            # a reference-written file: torch.optim.Adam's and torch_ema's own state dicts
            optim.load_reference_state(sd['optim'], sd.get('ema'))
        else:
            optim.load_state_dict(sd['optim'])
    if scaler is not None and sd.get('scaler') is not None and hasattr(scaler, 'load_state_dict'):
        sc = dict(sd['scaler'])
        if 'steps' not in sc and optim is not None:
            # torch's GradScaler does not count optimiser steps; the device-side scaler does (bias corrections, LambdaLR)
            sc['steps'] = int(optim.step_count)
        scaler.load_state_dict(sc)
    return int(sd['iter_ctr'])
