"""nerfstyle_amd/losses.py (the PyTorch losses the stylisation stage back-propagates into the HIP renderer) against the
outputs of the reference's own loss.py on the same seeded inputs (tests/golden/make_goldens.py, run where the reference is
importable): values AND gradients w.r.t. the image features."""
import numpy as np
import torch


def _t(a):
    return torch.tensor(np.asarray(a))


def test_cosine_gram_helpers(golden):
    from nerfstyle_amd import losses as Ls
    d = Ls.cosine_dists(_t(golden['loss_f1']), _t(golden['loss_f2']))
    assert np.abs(d.numpy() - golden['loss_cosine_dists']).max() < 1e-6
    gram = Ls.GramStyleLoss(['f'])({'f': _t(golden['loss_fa'])}, {'f': _t(golden['loss_fb'])})
    assert abs(float(gram) - float(golden['loss_gram'])) < 1e-6 * max(1.0, abs(float(golden['loss_gram'])))
    lab = _t(golden['loss_labels'])
    assert np.array_equal(Ls.labels_downscale(lab, (12, 10)).numpy(), golden['loss_labels_down'])
    assert np.abs(Ls.compute_centroid(lab == 1).numpy() - golden['loss_centroid']).max() < 1e-6


def test_nnfm_value_and_gradient(golden):
    from nerfstyle_amd import losses as Ls
    fa = _t(golden['loss_fa']).requires_grad_(True)
    fb = _t(golden['loss_fb'])[:, :, :, :10]
    v = Ls.NNFMStyleLoss(['f'])({'f': fa}, {'f': fb})
    v.backward()
    assert abs(float(v) - float(golden['loss_nnfm'])) < 1e-6
    assert np.abs(fa.grad.numpy() - golden['loss_nnfm_grad']).max() < 1e-7 + 1e-4 * np.abs(golden['loss_nnfm_grad']).max()


def test_semantic_style_loss_plain_and_matched(golden):
    from nerfstyle_amd import losses as Ls
    labels = _t(golden['loss_labels'])
    fb = _t(golden['loss_fb'])
    # (a) no clusters
    sem = Ls.SemanticStyleLoss(['f'])
    sem.init_feats({'f': fb}, num_classes=3)
    fa = _t(golden['loss_fa']).requires_grad_(True)
    v = sem({'f': fa}, None, labels, 0)
    v.backward()
    assert abs(float(v) - float(golden['loss_sem_plain'])) < 1e-6
    assert np.abs(fa.grad.numpy() - golden['loss_sem_plain_grad']).max() < 1e-7 + 1e-4 * np.abs(golden['loss_sem_plain_grad']).max()
    # (b) clusters, matching found by the Hungarian step on the first frame
    sem2 = Ls.SemanticStyleLoss(['f'], clusters=golden['loss_sem_clusters'])
    sem2.init_feats({'f': fb}, num_classes=3)
    assert np.array_equal(sem2.clusters.numpy(), golden['loss_sem_clusters_small'])
    assert np.abs(sem2.style_feats_mean.numpy() - golden['loss_sem_style_mean']).max() < 1e-6
    assert np.abs(sem2.style_centroids.numpy() - golden['loss_sem_style_centroids']).max() < 1e-6
    fa = _t(golden['loss_fa']).requires_grad_(True)
    v2 = sem2({'f': fa}, None, labels, 0)
    v2.backward()
    assert list(sem2.matching) == [int(x) for x in golden['loss_sem_matching']]
    assert abs(float(v2) - float(golden['loss_sem_match'])) < 1e-6
    assert np.abs(fa.grad.numpy() - golden['loss_sem_match_grad']).max() < 1e-7 + 1e-4 * np.abs(golden['loss_sem_match_grad']).max()
    assert float(v2) > float(v)                       # restricting the candidates can only increase the nearest distance
    # (c) matching given up front
    sem3 = Ls.SemanticStyleLoss(['f'], clusters=golden['loss_sem_clusters'], matching=[2, 0, 1])
    sem3.init_feats({'f': fb}, num_classes=3)
    assert abs(float(sem3({'f': _t(golden['loss_fa'])}, None, labels, 0)) - float(golden['loss_sem_fixed'])) < 1e-6
    # chunked nearest-neighbour search == one-shot search
    f1 = torch.nn.functional.normalize(torch.randn(300, 8), dim=1)
    f2 = torch.nn.functional.normalize(torch.randn(170, 8), dim=1)
    a, _ = Ls.nearest_style_index(f1, f2, chunk=64)
    b, _ = Ls.nearest_style_index(f1, f2, chunk=100000)
    assert torch.equal(a, b) and torch.equal(a, torch.argmin(1 - f1 @ f2.T, dim=1))


def test_grouped_matching_equals_masked_search():
    """nearest_style_index groups positions by cluster instead of masking a full distance matrix: same arg-min as the masked
    form (loss.py:201-206), including unrestricted rows (-1), unlabelled style positions (-1) and a class whose cluster is empty."""
    from nerfstyle_amd import losses as Ls
    g = torch.Generator().manual_seed(3)
    N1, N2, C, K = 301, 257, 24, 4
    f1 = torch.nn.functional.normalize(torch.randn(N1, C, generator=g), dim=1)
    f2 = torch.nn.functional.normalize(torch.randn(N2, C, generator=g), dim=1)
    rc = torch.randint(-1, K, (N1,), generator=g)
    sc = torch.randint(-1, K - 1, (N2,), generator=g)            # cluster K-1 has no style position at all
    idx, valid = Ls.nearest_style_index(f1, f2, rc, sc, chunk=64)
    d = 1.0 - f1 @ f2.T
    allowed = (sc[None, :] == rc[:, None]) | (rc[:, None] < 0)
    ref_valid = allowed.any(dim=1)
    ref_idx = torch.argmin(d.masked_fill(~allowed, float('inf')), dim=1)
    assert torch.equal(valid, ref_valid)
    assert (~ref_valid).sum() > 0 and (rc < 0).sum() > 0
    assert torch.equal(idx[ref_valid], ref_idx[ref_valid])
    # precomputed groups give the same answer
    groups = [torch.nonzero(sc == k)[:, 0] for k in range(K)]
    idx2, valid2 = Ls.nearest_style_index(f1, f2, rc, sc, chunk=1000, style_groups=groups)
    assert torch.equal(idx2[ref_valid], ref_idx[ref_valid]) and torch.equal(valid2, ref_valid)
    # no clusters: plain nearest neighbour
    idx3, _ = Ls.nearest_style_index(f1, f2)
    assert torch.equal(idx3, torch.argmin(d, dim=1))
