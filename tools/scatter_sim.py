"""CPU model of the backward's atomic-request count under the merge rule measured on the box
(tools/atomic_merge_rule.hip): one request per (wave-instruction, 64-byte line), except that exact
duplicate rows inside one instruction each cost their own request.  Replays the record stream the
scatter would emit for the bench scene and compares record orders.  Test/analysis tool only (uses
oracle/ for marching and corner rows); not imported by the product."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from oracle import oracle as O
from nerfstyle_amd.scene import load_room_cameras, synthetic_density_grid


def samples(n_rays=3000, seed=0):
    poses, intr, _ = load_room_cameras(2)
    rng = np.random.default_rng(seed)
    pix = rng.choice(intr.w * intr.h, n_rays, replace=False)
    o, d = O.generate_rays(poses[0], intr.w, intr.h, intr.fx, intr.fy, intr.cx, intr.cy, 3, pix_indices=pix)[:2]
    grid = synthetic_density_grid(2.0, 128, n_boxes=28, seed=0)
    bits = O.packbits(grid, 0.5)
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = O.near_far_from_aabb(o, d, aabb, 0.2)
    out = O.march_rays_train(o, d, 2.0, bits, 2, 128, nears, fars, max_steps=1024)
    xyzs, rays = out[0], out[3]
    order = np.argsort(rays[:, 0], kind='stable')
    idx = np.concatenate([np.arange(rays[i, 1], rays[i, 1] + rays[i, 2]) for i in order])
    return xyzs[idx]


def requests(stream):
    """stream: 1-D array of rows in ring order -> requests when drained 16 records per instruction"""
    n = len(stream) // 16 * 16
    s = stream[:n].reshape(-1, 16)
    req = 0
    for ins in s:
        lines = {}
        for r in ins:
            d = lines.setdefault(r >> 2, {})
            d[r] = d.get(r, 0) + 1
        req += sum(max(d.values()) for d in lines.values())
    return req


def emit(rows, order, levels_of_call, pair_calls=(), window=512, dedupe=True):
    """rows: [L, M, 8] -> ring stream.  order 'corner': per call (4 levels at once, one per 16-lane
    group), per corner: run tails of the 16 samples, group by group.  pair_calls: calls whose x / x+1
    records of a sample are emitted adjacent."""
    L, M, _ = rows.shape
    M = M // 16 * 16
    stream = []
    seen = {}
    for t in range(0, M, 16):
        for ci, lv in enumerate(levels_of_call):
            tile = rows[lv, t:t + 16]                     # [4, 16, 8]
            if ci in pair_calls:
                for pr in range(4):
                    a, b = tile[:, :, 2 * pr], tile[:, :, 2 * pr + 1]
                    for g in range(tile.shape[0]):
                        ta = np.r_[a[g, 1:] != a[g, :-1], True]
                        tb = np.r_[b[g, 1:] != b[g, :-1], True]
                        for s in range(16):
                            for r, tl in ((a[g, s], ta[s]), (b[g, s], tb[s])):
                                if not tl:
                                    continue
                                if dedupe and r in seen and len(stream) - seen[r] <= window:
                                    continue
                                seen[r] = len(stream)
                                stream.append(r)
            else:
                for c in range(8):
                    for g in range(tile.shape[0]):
                        k = tile[g, :, c]
                        tl = np.r_[k[1:] != k[:-1], True]
                        for r in k[tl]:
                            if dedupe and r in seen and len(stream) - seen[r] <= window:
                                continue
                            seen[r] = len(stream)
                            stream.append(r)
    return np.array(stream, np.int64), M


if __name__ == '__main__':
    x = samples(int(sys.argv[1]) if len(sys.argv) > 1 else 1500)
    enc = O.encoder_inputs(x, 2.0)
    off = O.grid_offsets(16, O.per_level_scale_from_cfg(), 16, 19)
    rows = O.grid_corner_rows(enc, off, O.per_level_scale_from_cfg(), 16, 0, True).astype(np.int64)
    if rows[1].max() < off[1]:
        rows += np.asarray(off[:16], np.int64)[:, None, None]      # level-local -> arena rows
    print('samples', x.shape[0])
    cur = [[0, 2, 4, 6], [1, 3, 5, 7], [8, 10, 12, 14], [9, 11, 13, 15]]
    alt = [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    for name, calls, pc in (('current', cur, ()), ('current, pair calls 2,3', cur, (2, 3)),
                            ('regrouped', alt, ()), ('regrouped, pair call 3', alt, (3,)),
                            ('regrouped, pair calls 2,3', alt, (2, 3)), ('regrouped, all pair', alt, (0, 1, 2, 3))):
        st, M = emit(rows, 'corner', calls, pc)
        print('%-28s records/sample %.2f requests/sample %.2f' % (name, len(st) / M, requests(st) / M))
    st, M = emit(rows, 'corner', cur, (), dedupe=False)
    print('current, no dedupe: records/sample %.2f requests/sample %.2f' % (len(st) / M, requests(st) / M))
    for w in (64, 128, 256, 1024):
        st, M = emit(rows, 'corner', cur, (), window=w)
        print('current, window %d: records/sample %.2f requests/sample %.2f' % (w, len(st) / M, requests(st) / M))
    for l in range(16):
        one = rows[l:l + 1]
        st, M = emit(one, 'corner', [[0]], (), dedupe=False)
        sd, M = emit(one, 'corner', [[0]], ())
        sp, _ = emit(one, 'corner', [[0]], (0,))
        print('level %2d alone: records/sample nodedupe %.2f dedupe %.2f  requests/sample nodedupe %.2f corner-major %.2f  pair %.2f' % (
            l, len(st) / M, len(sd) / M, requests(st) / M, requests(sd) / M, requests(sp) / M))
