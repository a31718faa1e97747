"""Timing of the stylisation stage's PyTorch part (VGG16 relu3 + semantic NNFM) at 1008x756 in several modes.
Run on the GPU box: python tools/exp_vgg.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerfstyle_amd.vgg import VGG16FeatureExtractor
from nerfstyle_amd.losses import SemanticStyleLoss
from nerfstyle_amd.stylize import StyleCriterion

dev = torch.device('cuda:0')
H, W, nc = 756, 1008, 5
g = torch.Generator(device=dev); g.manual_seed(0)
style = torch.rand(3, H, W, device=dev, generator=g)
seg = torch.randint(0, nc, (H, W), device=dev, generator=g)
target = torch.rand(3, H, W, device=dev, generator=g)
classes = torch.rand(H, W, nc, device=dev, generator=g)
fx = VGG16FeatureExtractor(['relu3']).to(dev)


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def run(mode_name, amp=None, bench=False, cl=False):
    torch.backends.cudnn.benchmark = bench
    m = fx.to(memory_format=torch.channels_last) if cl else fx
    crit = StyleCriterion(m, SemanticStyleLoss(['relu3'], clusters=seg), content_lambda=0.001, style_lambda=1.0)
    with torch.autocast('cuda', dtype=amp, enabled=amp is not None):
        crit.init_style(style, num_classes=nc)
    rgb = torch.rand(H, W, 3, device=dev, generator=g).requires_grad_(True)

    def vgg_only():
        with torch.autocast('cuda', dtype=amp, enabled=amp is not None):
            f = m(rgb.permute(2, 0, 1))['relu3']
            l = f.float().square().mean()
        l.backward()

    def full():
        with torch.autocast('cuda', dtype=amp, enabled=amp is not None):
            l = crit(rgb, target, classes, frame_key=0, it=1)[0]
        l.backward()
        return l
    t_v = timeit(vgg_only)
    t_f = timeit(full)
    print('{:28s} vgg fwd+bwd {:7.2f} ms   criterion fwd+bwd {:7.2f} ms   loss {:.6f}'.format(mode_name, t_v, t_f, float(full())), flush=True)


run('fp32')
run('fp32 benchmark', bench=True)
run('f16 autocast', amp=torch.float16)
run('f16 autocast benchmark', amp=torch.float16, bench=True)
run('bf16 autocast benchmark', amp=torch.bfloat16, bench=True)
run('f16 autocast bench chlast', amp=torch.float16, bench=True, cl=True)
