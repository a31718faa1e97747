"""VGG feature extractor of the stylisation stage: the counterpart of the reference's networks/fx.py
(`VGGFeatureExtractor`, `VGG16FeatureExtractor`: keys like 'relu3' / 'relu3_1' / 'conv4_2', ImageNet normalisation,
:11-96) without torchvision (absent offline): the VGG16 / VGG19 convolution stacks are built here with torchvision's layer
indexing (`features.N`), only as deep as the requested keys need, and run as stock PyTorch-ROCm convolutions (MIOpen).

Weights: torchvision's pretrained files cannot be fetched offline (fx.py:91 asks for weights='DEFAULT').  Pass
`weights_path` = a state_dict file with torchvision's key names ('features.0.weight', ...; loaded with
weights_only=True) to use real weights; otherwise the stack is initialised with seeded He-normal weights, which gives
the same shapes, FLOPs and gradient flow (throughput runs and gradient tests; not a perceptual loss)."""
import re
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

# conv layer indices per block in torchvision's `features` Sequential (fx.py:90,95)
VGG16_LAYERS = [[0, 2], [5, 7], [10, 12, 14], [17, 19, 21], [24, 26, 28]]
VGG19_LAYERS = [[0, 2], [5, 7], [10, 12, 14, 16], [19, 21, 23, 25], [28, 30, 32, 34]]
_CHANNELS = [64, 128, 256, 512, 512]
NODE_PATTERN = r'^(conv|relu)([1-5])(?:_([1-4]))?$'


class VGGFeatureExtractor(nn.Module):
    def __init__(self, keys: Union[str, List[str]], layers=VGG16_LAYERS, weights_path: Optional[str] = None, seed: int = 0):
        super().__init__()
        if isinstance(keys, str):
            keys = [keys]
        self.layers = layers
        self.keys = []              # (key, [sub keys]) in request order, as fx.py:57
        self.nodes = {}             # features index -> sub key name
        for k in keys:
            m = re.match(NODE_PATTERN, k)
            if not m:
                raise ValueError('"{}" is an invalid identifier'.format(k))
            op, block, layer = m.groups()
            is_relu = int(op == 'relu')
            b = int(block) - 1
            if layer is None:
                subs = []
                for i, idx in enumerate(layers[b]):
                    name = '{}_{}'.format(k, i + 1)
                    self.nodes[idx + is_relu] = name
                    subs.append(name)
            else:
                self.nodes[layers[b][int(layer) - 1] + is_relu] = k
                subs = [k]
            self.keys.append((k, subs))
        last = max(self.nodes)
        mods, cin = [], 3
        for b, idxs in enumerate(layers):
            for idx in idxs:
                while len(mods) < idx:
                    mods.append(nn.MaxPool2d(2, 2))           # the only gaps in the index lists are the pools
                mods.append(nn.Conv2d(cin, _CHANNELS[b], 3, padding=1))
                mods.append(nn.ReLU(inplace=False))
                cin = _CHANNELS[b]
        self.features = nn.Sequential(*mods[:last + 1])
        g = torch.Generator().manual_seed(seed)
        for mod in self.features:
            if isinstance(mod, nn.Conv2d):
                fan_in = mod.in_channels * 9
                with torch.no_grad():
                    mod.weight.normal_(0.0, (2.0 / fan_in) ** 0.5, generator=g)
                    mod.bias.zero_()
        if weights_path is not None:
            sd = torch.load(weights_path, map_location='cpu', weights_only=True)
            own = self.state_dict()
            self.load_state_dict({k: sd[k] for k in own})
        for p in self.parameters():
            p.requires_grad_(False)
        self.eval()
        self.register_buffer('mean', torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
        self.register_buffer('std', torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

    def forward(self, x: torch.Tensor, detach: bool = False) -> Dict[str, torch.Tensor]:
        """x [3,H,W] or [N,3,H,W] in [0,1] -> {key: [N, C, h, w]} (a block key concatenates its layers' maps, fx.py:81-84)"""
        if x.dim() == 3:
            x = x.unsqueeze(0)
        assert x.dim() == 4
        h = (x.float() - self.mean) / self.std
        feats = {}
        for i, mod in enumerate(self.features):
            h = mod(h)
            if i in self.nodes:
                feats[self.nodes[i]] = h.detach() if detach else h
        return {k: torch.cat([feats[s] for s in subs], dim=1) for k, subs in self.keys}


class VGG16FeatureExtractor(VGGFeatureExtractor):
    def __init__(self, keys, weights_path: Optional[str] = None, seed: int = 0):
        super().__init__(keys, VGG16_LAYERS, weights_path, seed)


class VGG19FeatureExtractor(VGGFeatureExtractor):
    def __init__(self, keys, weights_path: Optional[str] = None, seed: int = 0):
        super().__init__(keys, VGG19_LAYERS, weights_path, seed)
