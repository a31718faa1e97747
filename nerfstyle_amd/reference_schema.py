"""Reading a checkpoint written BY THE REFERENCE (trainers/base.py:231-249) without executing anything from the file.

The reference pickles class instances by reference: its config dataclasses (`config.BaseConfig`, `DatasetConfig`,
`NetworkConfig` + nested `HashGridConfig`, `RendererConfig`, `TrainConfig` + nested `TrainIntervalConfig`,
`DatasetConfig.ReplicaConfig`; config.py:222-446), `common.Intrinsics` inside `renderer` (renderer.py:78-91) and a
`pathlib.PosixPath` (`log_dir`, the Path-typed config fields).  torch.load(weights_only=True) refuses such a file unless the
globals it names are allow-listed.  Here every one of those names is mapped to an inert STAND-IN defined in this module:

  * the dataclasses and Intrinsics -> empty classes that only receive the pickled field dict (the unpickler's NEWOBJ + BUILD:
    `cls.__new__(cls)` then `obj.__dict__.update(state)`; no __init__, no __setstate__, no methods);
  * `pathlib.PosixPath` -> a function that joins the pickled path parts into a str;
  * `builtins.getattr` -> a function that resolves ONLY the three nested stand-in classes on their three parents (pickle
    protocol 2, torch.save's default, reaches a nested class through getattr(parent, name));

so the reference's own classes are never imported and no code from the file or from the reference tree runs.  The result is
converted to the plain layout of nerfstyle_amd.checkpoint (dataclass -> {'__dataclass__': name, **fields}).

Not covered (external to the reference tree, "parity unpinned"): tinycudann's private `*.params` layout -- the MLP vectors
of such a file are taken in THIS build's layout (nerfstyle_amd/network.py: row-major [out, in] per layer, last layer padded to
16 rows), which is the build's own definition; and torch_ema's `ema` state is passed through as the plain dict it is."""
import torch

_CONFIG_CLASSES = ['BaseConfig', 'DatasetConfig', 'NetworkConfig', 'RendererConfig', 'TrainConfig']
_NESTED = {'NetworkConfig': ['HashGridConfig'], 'TrainConfig': ['TrainIntervalConfig'], 'DatasetConfig': ['ReplicaConfig']}


def _standin(module, qualname):
    cls = type(qualname.split('.')[-1], (), {'__slots__': ('__dict__',), '__doc__': 'inert stand-in for {}.{}'.format(module, qualname)})
    cls.__module__ = module
    cls.__qualname__ = qualname
    return cls


STANDINS = {}
for _n in _CONFIG_CLASSES:
    STANDINS['config.' + _n] = _standin('config', _n)
for _p, _kids in _NESTED.items():
    for _k in _kids:
        STANDINS['config.{}.{}'.format(_p, _k)] = _standin('config', '{}.{}'.format(_p, _k))
        setattr(STANDINS['config.' + _p], _k, STANDINS['config.{}.{}'.format(_p, _k)])
STANDINS['common.Intrinsics'] = _standin('common', 'Intrinsics')
_STANDIN_SET = set(STANDINS.values())


def _path_standin(*parts):
    """pathlib.PosixPath.__reduce__ -> (PosixPath, parts): joined to a str, no filesystem object is created"""
    parts = [str(p) for p in parts]
    if parts and parts[0] == '/':
        return '/' + '/'.join(parts[1:])
    return '/'.join(parts)


def _getattr_standin(obj, name):
    """pickle protocol < 4 writes a nested class as getattr(parent_class, name): only the nested stand-ins resolve"""
    if obj in _STANDIN_SET and isinstance(name, str) and name in _NESTED.get(obj.__name__, ()):
        return getattr(obj, name)
    raise RuntimeError('checkpoint asks for getattr({!r}, {!r}): not part of the reference checkpoint schema'.format(obj, name))


def safe_globals():
    """The allow-list for torch.load(weights_only=True): (stand-in, 'module.qualname the file names') pairs."""
    out = [(cls, path) for path, cls in STANDINS.items()]
    out += [(_path_standin, 'pathlib.PosixPath'), (_getattr_standin, 'builtins.getattr')]
    return out


def to_plain(v):
    """stand-in instances -> {'__dataclass__': name, **fields}; containers recursively; tensors to the CPU"""
    if torch.is_tensor(v):
        return v.detach().cpu()
    if type(v) in _STANDIN_SET:
        d = {'__dataclass__': type(v).__name__}
        for k, x in vars(v).items():
            d[str(k)] = to_plain(x)
        return d
    if isinstance(v, dict):
        return {k: to_plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [to_plain(x) for x in v]
    return v


def load_reference_checkpoint(path, map_location='cpu'):
    """-> the checkpoint as plain data (same layout as nerfstyle_amd.checkpoint writes).  Raises pickle.UnpicklingError if the
    file names any global outside the reference's schema."""
    with torch.serialization.safe_globals(safe_globals()):
        sd = torch.load(str(path), map_location=map_location, weights_only=True)
    return to_plain(sd)
