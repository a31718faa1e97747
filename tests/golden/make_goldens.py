"""Generates tests/golden/reference_python.npz from the importable Python pieces of the
reference (run in the authoring container only; /root/reference does not travel).

What is imported from /root/reference: `common`, `nerf_lib`, `utils`, `loss` -- plain
numpy/torch code.  `utils/__init__.py` imports two non-arithmetic third-party packages
that are absent here (`git` = GitPython, `torch_ema`); they are replaced by inert, empty
stand-in modules so that the import succeeds.  Nothing from the
stand-ins is ever called.  The reference's CUDA extensions (raymarching, gridencoder),
tinycudann and torchvision are NOT importable and are not touched.

The outputs are data only (inputs + expected outputs); no reference source is copied.
Also writes nerfstyle_amd/assets/llff_room_cameras.json (the LLFF 'room' camera poses and
intrinsics from the reference's datasets/nerf_llff_data/room/transforms_train.json --
a data file -- with the dataset scale 0.33 of cfgs/dataset/llff_room.yaml applied).

Usage: python tests/golden/make_goldens.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def import_reference():
    git = types.ModuleType('git')
    torch_ema = types.ModuleType('torch_ema')

    class ExponentialMovingAverage:  # inert stand-in, never instantiated here
        pass
    torch_ema.ExponentialMovingAverage = ExponentialMovingAverage
    sys.modules.setdefault('git', git)
    sys.modules.setdefault('torch_ema', torch_ema)
    sys.path.insert(0, REF)
    import common
    import nerf_lib
    import utils
    import loss
    return common, nerf_lib, utils, loss


def main():
    common, nerf_lib_mod, utils, loss = import_reference()
    out = {}

    # ---- cameras (data) -------------------------------------------------------------------
    with open(os.path.join(REF, 'datasets/nerf_llff_data/room/transforms_train.json')) as f:
        tj = json.load(f)
    poses = np.array([fr['transform_matrix'] for fr in tj['frames']], dtype=np.float32)
    poses[:, :3, 3] *= 0.33                       # base_dataset.py:64, cfgs/dataset/llff_room.yaml:4
    cams = {
        'w': int(tj['w']), 'h': int(tj['h']), 'fl_x': tj['fl_x'], 'fl_y': tj['fl_y'],
        'cx': tj['cx'], 'cy': tj['cy'], 'scale': 0.33, 'bound': 2.0, 'flip_camera': 3,
        'poses': [[[float(v) for v in row] for row in p] for p in poses.astype(np.float64)],
        'source': 'hkust-vgd/nerfstyle datasets/nerf_llff_data/room/transforms_train.json '
                  '(translation x0.33 applied)',
    }
    os.makedirs(os.path.join(REPO, 'nerfstyle_amd', 'assets'), exist_ok=True)
    with open(os.path.join(REPO, 'nerfstyle_amd', 'assets', 'llff_room_cameras.json'), 'w') as f:
        json.dump(cams, f)
    # LLFF 'fern' cameras (BASELINE config 4), same treatment: cfgs/dataset/llff_fern.yaml has scale 0.33, bound 2
    with open(os.path.join(REF, 'datasets/nerf_llff_data/fern/transforms_train.json')) as f:
        fj = json.load(f)
    fposes = np.array([fr['transform_matrix'] for fr in fj['frames']], dtype=np.float32)
    fposes[:, :3, 3] *= 0.33
    fern = {
        'w': int(fj['w']), 'h': int(fj['h']), 'fl_x': fj['fl_x'], 'fl_y': fj['fl_y'], 'cx': fj['cx'], 'cy': fj['cy'],
        'scale': 0.33, 'bound': 2.0, 'flip_camera': 3,
        'poses': [[[float(v) for v in row] for row in p] for p in fposes.astype(np.float64)],
        'source': 'hkust-vgd/nerfstyle datasets/nerf_llff_data/fern/transforms_train.json (translation x0.33 applied)',
    }
    with open(os.path.join(REPO, 'nerfstyle_amd', 'assets', 'llff_fern_cameras.json'), 'w') as f:
        json.dump(fern, f)

    # ---- generate_rays (nerf_lib.py:69-142) --------------------------------------------------
    lib = nerf_lib_mod.nerf_lib
    lib._device = torch.device('cpu')       # the public setter asserts CUDA (nerf_lib.py:35)
    lib._ready = True
    intr = common.Intrinsics(h=cams['h'], w=cams['w'], fx=cams['fl_x'], fy=cams['fl_y'],
                             cx=cams['cx'], cy=cams['cy'])
    pose = torch.tensor(poses[0])
    rays, _ = lib.generate_rays(pose, intr, camera_flip=3)
    sel = np.arange(0, cams['w'] * cams['h'], 997)
    out['rays_full_sel'] = sel
    out['rays_full_o'] = rays.origins.numpy()[sel]
    out['rays_full_d'] = rays.dirs.numpy()[sel]
    rays, _ = lib.generate_rays(pose, intr, patch=common.Box2D(200, 0, 200, 200), camera_flip=3)
    sel = np.arange(0, 200 * 200, 101)
    out['rays_patch_sel'] = sel
    out['rays_patch_o'] = rays.origins.numpy()[sel]
    out['rays_patch_d'] = rays.dirs.numpy()[sel]
    np.random.seed(69420)
    rays, _ = lib.generate_rays(pose, intr, bsize=4096, camera_flip=3)
    np.random.seed(69420)
    idx = np.random.choice(np.arange(cams['w'] * cams['h']), 4096, replace=False)
    out['rays_rand_idx'] = idx
    out['rays_rand_o'] = rays.origins.numpy()[:512]
    out['rays_rand_d'] = rays.dirs.numpy()[:512]
    out['pose0'] = poses[0]

    # ---- integrate_points (nerf_lib.py:179-219) ---------------------------------------------
    g = torch.Generator().manual_seed(1234)
    N, K = 256, 64
    dists = torch.rand(N, K, generator=g) * 0.05
    rgbs = torch.rand(N, K, 3, generator=g)
    dens = torch.rand(N, K, generator=g) * 20
    rgb_map, acc_map, trans_map = lib.integrate_points(
        dists, rgbs, dens, torch.zeros(N, 3), torch.zeros(N, 1), torch.ones(N, 1))
    out.update(ip_dists=dists.numpy(), ip_rgbs=rgbs.numpy(), ip_dens=dens.numpy(),
               ip_rgb_map=rgb_map.numpy(), ip_acc_map=acc_map.numpy(), ip_trans_map=trans_map.numpy())

    # ---- BBox.normalize (common.py:276-288) -------------------------------------------------
    bbox = common.BBox.from_radius(2.0)
    pts = (torch.rand(128, 3, generator=g) * 4 - 2)
    out['bbox_pts'] = pts.numpy()
    out['bbox_norm'] = bbox.normalize(pts).numpy()

    # ---- trunc_exp fwd/bwd (utils/__init__.py:496-513 == tcnn_nerf.py:55-69) ----------------
    x = torch.tensor([-20., -15.5, -3., 0., 0.5, 7., 14.9, 15.1, 18.], requires_grad=True)
    y = utils.trunc_exp(x)
    y.backward(torch.ones_like(y) * 0.5)
    out['texp_x'] = x.detach().numpy()
    out['texp_y'] = y.detach().numpy()
    out['texp_gx'] = x.grad.numpy()

    # ---- density2alpha / compute_psnr --------------------------------------------------------
    out['d2a'] = utils.density2alpha(dens[:4], dists[:4]).numpy()
    mse = torch.tensor(3.7e-4)
    out['psnr_in'] = mse.numpy()
    out['psnr_out'] = utils.compute_psnr(mse).numpy()

    # ---- loss.py (stays PyTorch in the build; pins the loss the renderer back-props from) -----
    fa = torch.rand(1, 8, 12, 10, generator=g)
    fb = torch.rand(1, 8, 9, 11, generator=g)
    out['loss_fa'] = fa.numpy()
    out['loss_fb'] = fb.numpy()
    f1 = torch.rand(40, 8, generator=g)
    f2 = torch.rand(30, 8, generator=g)
    out['loss_f1'] = f1.numpy()
    out['loss_f2'] = f2.numpy()
    out['loss_cosine_dists'] = loss.cosine_dists(f1, f2).numpy()
    nnfm = loss.NNFMStyleLoss(['f'])
    out['loss_nnfm'] = np.array(float(nnfm({'f': fa}, {'f': fb[:, :, :, :10]})))
    gram = loss.GramStyleLoss(['f'])
    out['loss_gram'] = np.array(float(gram({'f': fa}, {'f': fb})))

    # gradients of the NNFM loss w.r.t. the image features (what back-propagates into the renderer)
    fa_g = fa.clone().requires_grad_(True)
    nnfm({'f': fa_g}, {'f': fb[:, :, :, :10]}).backward()
    out['loss_nnfm_grad'] = fa_g.grad.numpy()

    # ---- SemanticStyleLoss (loss.py:116-214), the loss StyleTrainer uses (trainers/style.py:50-52) --------------------
    out['loss_labels'] = torch.randint(0, 3, (37, 29), generator=g).numpy()
    out['loss_labels_down'] = loss.labels_downscale(torch.tensor(out['loss_labels']), (12, 10)).numpy()
    mask = torch.tensor(out['loss_labels']) == 1
    out['loss_centroid'] = loss.compute_centroid(mask).numpy()
    # (a) without clusters: plain nearest-neighbour feature matching over all style positions
    sem = loss.SemanticStyleLoss(['f'], clusters_path=None)
    sem.init_feats({'f': fb}, num_classes=3)
    fa_g = fa.clone().requires_grad_(True)
    v = sem({'f': fa_g}, None, torch.tensor(out['loss_labels']), 0)
    v.backward()
    out['loss_sem_plain'] = np.array(float(v))
    out['loss_sem_plain_grad'] = fa_g.grad.numpy()
    # (b) with style clusters.  The constructor's clusters branch calls .cuda() (loss.py:138), so the object is built
    # without a path and given the cluster map by attribute, then the reference's own init_feats / update_matching /
    # forward run on the CPU unchanged.
    clusters = torch.randint(0, 3, (40, 44), generator=g)
    sem2 = loss.SemanticStyleLoss(['f'], clusters_path=None)
    sem2.use_matching = True
    sem2.clusters = clusters.clone()
    sem2.n_clusters = 3
    sem2.matching = None
    sem2.init_feats({'f': fb}, num_classes=3)
    fa_g = fa.clone().requires_grad_(True)
    v2 = sem2({'f': fa_g}, None, torch.tensor(out['loss_labels']), 0)
    v2.backward()
    out['loss_sem_clusters'] = clusters.numpy()
    out['loss_sem_clusters_small'] = sem2.clusters.numpy()
    out['loss_sem_matching'] = np.asarray(sem2.matching)
    out['loss_sem_match'] = np.array(float(v2))
    out['loss_sem_match_grad'] = fa_g.grad.numpy()
    out['loss_sem_style_mean'] = sem2.style_feats_mean.numpy()
    out['loss_sem_style_centroids'] = sem2.style_centroids.numpy()
    # (c) with a fixed matching given up front (--style_matching, trainers/style.py:47-49)
    sem3 = loss.SemanticStyleLoss(['f'], clusters_path=None)
    sem3.use_matching = True
    sem3.clusters = clusters.clone()
    sem3.n_clusters = 3
    sem3.matching = [2, 0, 1]
    sem3.init_feats({'f': fb}, num_classes=3)
    out['loss_sem_fixed'] = np.array(float(sem3({'f': fa}, None, torch.tensor(out['loss_labels']), 0)))

    np.savez_compressed(os.path.join(HERE, 'reference_python.npz'), **out)
    print('wrote', os.path.join(HERE, 'reference_python.npz'), sorted(out.keys()))


if __name__ == '__main__':
    main()
