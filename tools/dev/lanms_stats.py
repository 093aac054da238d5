"""Development aid: how much work the bound tests of the LANMS kernel leave per target tile on the bench's pages."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ocr_amd import synth
from ocr_amd.engine import HipContext

ctx = HipContext(0)
images, maps, _ = synth.make_pages(1238, 2, 1024, 1024, 220)
dec = ctx.east_decode(torch.from_numpy(maps).cuda(), (1024, 1024), 0.35)
cnt = dec['counts'].cpu().numpy()
for b in range(2):
    bx = dec['xyxy'][b, :cnt[b]].cpu().numpy()
    n = len(bx)
    cx, cy = 0.5 * (bx[:, 0] + bx[:, 2]), 0.5 * (bx[:, 1] + bx[:, 3])
    print('n', n, 'coordinate range', bx.min(), bx.max())
    ix = np.clip((cx * 64).astype(int), 0, 63); iy = np.clip((cy * 64).astype(int), 0, 63)
    def part1by1(v):
        v = v & 0x3f
        v = (v | (v << 4)) & 0x0f0f
        v = (v | (v << 2)) & 0x3333
        v = (v | (v << 1)) & 0x5555
        return v
    def hilbert(x, y, order=6):
        d = np.zeros_like(x)
        x = x.copy(); y = y.copy()
        s_ = 1 << (order - 1)
        while s_ > 0:
            rx = ((x & s_) > 0).astype(int); ry = ((y & s_) > 0).astype(int)
            d += s_ * s_ * ((3 * rx) ^ ry)
            # rotate
            m = ry == 0
            flip = m & (rx == 1)
            x = np.where(flip, s_ - 1 - x, x); y = np.where(flip, s_ - 1 - y, y)
            x, y = np.where(m, y, x), np.where(m, x, y)
            x &= (s_ - 1); y &= (s_ - 1)     # keep the low bits
            s_ >>= 1
        return d
    for name, cell in (('row-major', iy * 64 + ix), ('hilbert', hilbert(ix, iy))):
        order = np.argsort(cell, kind='stable')
        s = bx[order]
        nt = (n + 63) // 64
        bounds = np.array([[s[t*64:(t+1)*64, 0].min(), s[t*64:(t+1)*64, 1].min(), s[t*64:(t+1)*64, 2].max(), s[t*64:(t+1)*64, 3].max()] for t in range(nt)])
        tiles_hit, boxes_hit = [], []
        for t in range(nt):
            tb = bounds[t]
            th = (bounds[:, 0] <= tb[2]) & (tb[0] <= bounds[:, 2]) & (bounds[:, 1] <= tb[3]) & (tb[1] <= bounds[:, 3])
            tiles_hit.append(th.sum())
            idx = np.nonzero(th)[0]
            k = 0
            for S in idx:
                u = s[S*64:(S+1)*64]
                k += ((u[:, 0] <= tb[2]) & (tb[0] <= u[:, 2]) & (u[:, 1] <= tb[3]) & (tb[1] <= u[:, 3])).sum()
            boxes_hit.append(k)
        # the two parts of a tile: cut before the entry with the largest jump of the centre from its predecessor
        parts = []
        for t in range(nt):
            tb_ = s[t*64:(t+1)*64]
            c = np.stack([tb_[:, 0] + tb_[:, 2], tb_[:, 1] + tb_[:, 3]], 1)
            j = np.abs(np.diff(c, axis=0)).sum(1)
            cut = (int(np.argmax(j)) + 1) if len(j) and j.max() > 0 else min(32, len(tb_))
            pa, pb = tb_[:cut], tb_[cut:]
            ra = [pa[:, 0].min(), pa[:, 1].min(), pa[:, 2].max(), pa[:, 3].max()]
            rb = [pb[:, 0].min(), pb[:, 1].min(), pb[:, 2].max(), pb[:, 3].max()] if len(pb) else [np.inf, np.inf, -np.inf, -np.inf]
            parts.append((ra, rb))
        def meet(u, r):
            return (u[:, 0] <= r[2]) & (r[0] <= u[:, 2]) & (u[:, 1] <= r[3]) & (r[1] <= u[:, 3])
        part_hit, pair_hit = [], []
        for t in range(nt):
            ra, rb = parts[t]
            m = meet(s, ra) | meet(s, rb)
            part_hit.append(m.sum())
            # boxes with a real overlap with some box of the tile
            tb_ = s[t*64:(t+1)*64]
            u = s[m]
            iw = np.minimum(u[:, None, 2], tb_[None, :, 2]) - np.maximum(u[:, None, 0], tb_[None, :, 0])
            ih = np.minimum(u[:, None, 3], tb_[None, :, 3]) - np.maximum(u[:, None, 1], tb_[None, :, 1])
            pair_hit.append(((iw > 0) & (ih > 0)).any(1).sum())
        part_hit, pair_hit = np.array(part_hit), np.array(pair_hit)
        print('   boxes meeting one of the two parts: mean %.0f median %.0f max %d; boxes overlapping a box of the tile: mean %.0f max %d' % (
            part_hit.mean(), np.median(part_hit), part_hit.max(), pair_hit.mean(), pair_hit.max()))
        tiles_hit, boxes_hit = np.array(tiles_hit), np.array(boxes_hit)
        print(name, 'tiles', nt, 'candidate tiles per target: mean %.1f max %d; boxes inside the target bounds: mean %.0f median %.0f max %d' % (
            tiles_hit.mean(), tiles_hit.max(), boxes_hit.mean(), np.median(boxes_hit), boxes_hit.max()))
        w = bounds[:, 2] - bounds[:, 0]; h = bounds[:, 3] - bounds[:, 1]
        print('   tile bounds width mean %.3f max %.3f height mean %.3f max %.3f' % (w.mean(), w.max(), h.mean(), h.max()))
