"""CPU model: how many gradient records / atomic requests per sample the table scatter of the field backward needs when
the samples of a DENSE ray batch (a full frame: neighbouring pixels) are processed (a) in ray order, as round 1 does,
or (b) in Morton order of their finest-level cell with a per-wave, per-level LDS lattice tile that accumulates the
corner gradients of T^3 cells and is flushed (one record per touched corner) when the wave's sample stream leaves
the tile.  Analysis tool only (uses oracle/ for marching); not imported by the product.

usage: python tools/sorted_scatter_sim.py [patch_edge_px=96] [samples_per_wave=47000]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from oracle import oracle as O
from nerfstyle_amd.scene import load_room_cameras, synthetic_density_grid


def dense_patch_samples(edge=96, x0=400, y0=300):
    poses, intr, _ = load_room_cameras(2)
    ys, xs = np.meshgrid(np.arange(y0, y0 + edge), np.arange(x0, x0 + edge), indexing='ij')
    pix = (ys * intr.w + xs).reshape(-1)
    o, d = O.generate_rays(poses[0], intr.w, intr.h, intr.fx, intr.fy, intr.cx, intr.cy, 3, pix_indices=pix)[:2]
    grid = synthetic_density_grid(2.0, 128, n_boxes=28, seed=0)
    bits = O.packbits(grid, 0.5)
    nears, fars = O.near_far_from_aabb(o, d, np.array([-2, -2, -2, 2, 2, 2], np.float32), 0.2)
    xyzs, _, _, rays, cnt = O.march_rays_train(o, d, 2.0, bits, 2, 128, nears, fars, max_steps=1024)
    return xyzs[:int(cnt[0])], rays


def morton3(c):
    def spread(v):
        v = v.astype(np.uint64)
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        v = (v | (v << 2)) & 0x1249249249249249
        return v
    return spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2)


def level_cells(u, res):
    pos = u * np.float32(res)
    c = np.minimum(np.floor(pos), res - 1).astype(np.int64)
    return c


def lines_of(rows_sorted_by_emit):
    """requests of a record stream drained 16 per instruction under the measured merge rule"""
    n = len(rows_sorted_by_emit) // 16 * 16
    if n == 0:
        return 0
    s = rows_sorted_by_emit[:n].reshape(-1, 16)
    line = s >> 2
    # distinct lines per instruction (duplicates of the same row within an instruction are rare after merging)
    srt = np.sort(line, axis=1)
    return int((np.diff(srt, axis=1) != 0).sum() + len(srt))


def hash_row(cx, cy, cz, size, offset):
    idx = (cx.astype(np.uint64) ^ (cy.astype(np.uint64) * 2654435761 & 0xFFFFFFFF) ^ (cz.astype(np.uint64) * 805459861 & 0xFFFFFFFF)) & 0xFFFFFFFF
    return (idx % size).astype(np.int64) + offset


def main():
    edge = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    spw = int(sys.argv[2]) if len(sys.argv) > 2 else 47000
    xyz, rays = dense_patch_samples(edge, int(sys.argv[3]) if len(sys.argv) > 3 else 0, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    M = len(xyz)
    u = O.encoder_inputs(xyz, 2.0).astype(np.float32)
    pls = O.per_level_scale_from_cfg()
    off = O.grid_offsets(16, pls, 16, 19)
    res = O.grid_resolutions(16, O.grid_S(pls), 16)
    print('rays', edge * edge, 'samples', M, 'per ray', M / edge / edge)
    fin = level_cells(u, int(res[15]))
    key = morton3(fin - fin.min(0))
    order = np.argsort(key, kind='stable')
    corners = np.array([[(i >> d) & 1 for d in range(3)] for i in range(8)], np.int64)
    tot = {}
    for name, perm in (('ray order', np.arange(M)), ('morton order', order)):
        for T in ((None, 2, 3, 4) if name == 'morton order' else (None,)):
            rec_total, req_total, low_total = 0, 0, 0
            per_level = []
            for l in range(16):
                c = level_cells(u[perm], int(res[l]))
                size = int(off[l + 1] - off[l])
                wave = np.arange(M) // spw
                if T is None:
                    # run tracker: a record per corner stream whenever the cell changes (merges only same-cell runs;
                    # the real tracker also hands runs over to the neighbouring cell: slightly optimistic/pessimistic either way)
                    change = np.r_[True, (np.diff(c, axis=0) != 0).any(1) | (np.diff(wave) != 0)]
                    seg = np.cumsum(change) - 1
                else:
                    tile = c // T
                    change = np.r_[True, (np.diff(tile, axis=0) != 0).any(1) | (np.diff(wave) != 0)]
                    seg = np.cumsum(change) - 1
                # distinct corners per segment
                cc = (c[:, None, :] + corners[None]).reshape(-1, 3)
                sg = np.repeat(seg, 8)
                k = ((sg * 4099 + cc[:, 2]) * 4099 + cc[:, 1]) * 4099 + cc[:, 0]
                uk, first = np.unique(k, return_index=True)
                rec = len(uk)
                # emit order: by segment, then z, y, x (x fastest) -> rows
                cz = (uk // 1) % 4099
                rows = hash_row(uk % 4099, (uk // 4099) % 4099, (uk // 4099 ** 2) % 4099, size, int(off[l]))
                req = lines_of(rows)
                low = len(np.unique(hash_row(cc[:, 0], cc[:, 1], cc[:, 2], size, int(off[l]))))
                rec_total += rec
                req_total += req
                low_total += low
                per_level.append((rec / M, req / M, low / M))
            tag = '{} / {}'.format(name, 'run tracker' if T is None else 'LDS tile T={} ({} corners x 16 levels = {:.1f} KB/wave)'.format(T, (T + 1) ** 3, (T + 1) ** 3 * 16 * 16 / 1024))
            print('%-78s records/sample %6.2f  requests/sample %6.2f  (distinct rows/sample %.3f)' % (tag, rec_total / M, req_total / M, low_total / M))
            print('    per level records: ' + ' '.join('%.2f' % p[0] for p in per_level))


if __name__ == '__main__':
    main()
