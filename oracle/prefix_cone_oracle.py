"""TEST INFRASTRUCTURE -- numpy restatement of which items of an AR run's observed prefix anybody reads (the
dependency cone that k_prefix_starts in pixelsynth_amd/csrc/lmconv.hip keeps as one start rank per stage and frame).

Not a restatement of reference code: the reference runs a full forward per sampled code (models/lmconv/sample.py:54-66)
and has no prefix pass.  The graph walked here is the reference network's (models/lmconv/model.py:132-151: up pass u0..u8,
down pass d0..d9 with the skip pairs of layers.py:155-156) and the reads are the open taps of the kernel masks
(models/lmconv/masking.py:287-370).  Parity unpinned (nothing in the reference to pin it to); what the tests check is
(a) the device table equals this one and (b) AR results are bit-identical with and without the elimination.
"""
import numpy as np

G_IN = [0, 1, 3, 4, 6, 7, 8, 9, 11, 12, 13, 15, 16, 17]
G_OUT = [1, 2, 4, 5, 7, 8, 9, 10, 12, 13, 14, 16, 17, 18]
G_SKIP = [-1] * 6 + [7, 6, 5, 4, 3, 2, 1, 0]
D_IN, D_OUT = [2, 5, 10, 14], [3, 6, 11, 15]
EXEC = [("g", 0), ("g", 1), ("d", 0), ("g", 2), ("g", 3), ("d", 1), ("g", 4), ("g", 5), ("g", 6), ("g", 7), ("d", 2),
        ("g", 8), ("g", 9), ("g", 10), ("d", 3), ("g", 11), ("g", 12), ("g", 13)]
N_EVAL = 33   # 0 u_init, 1+g conv_input of gated block g, 15+g its conv_out, 29+d dilated conv d


def _min_tap_rank(order_loc, mask, H, W, dil, npre):
    """For every location (by rank): the smallest rank among its open, in-grid, non-centre taps (npre if none)."""
    L = H * W
    rank = np.empty(L, np.int64)
    rank[order_loc] = np.arange(L)
    out = np.full(L, npre, np.int64)
    for r in range(L):
        q = int(order_loc[r])
        y, x = divmod(q, W)
        for t in range(9):
            if t == 4 or mask[t, q] == 0:
                continue
            yy, xx = y + dil * (t // 3 - 1), x + dil * (t % 3 - 1)
            if 0 <= yy < H and 0 <= xx < W:
                out[r] = min(out[r], rank[yy * W + xx])
    return out


def prefix_starts(order_loc, mask_und, mask_dil, H, W, npre):
    """order_loc (L,) location by rank; mask_und / mask_dil (9, L) by location -> (33,) first rank evaluated per stage."""
    m1 = _min_tap_rank(order_loc, mask_und, H, W, 1, npre)
    m2 = _min_tap_rank(order_loc, mask_dil, H, W, 2, npre)
    c1 = int(min(npre, m1[npre:].min())) if npre < len(m1) else npre
    c2 = int(min(npre, m2[npre:].min())) if npre < len(m2) else npre
    s1 = np.minimum.accumulate(m1[:npre][::-1])[::-1] if npre else m1[:0]
    s2 = np.minimum.accumulate(m2[:npre][::-1])[::-1] if npre else m2[:0]
    suf = lambda s, r0: npre if r0 >= npre else int(min(r0, s[r0]))
    need = [npre] * 19
    need_x = [c1] * 14
    for g in range(14):
        need[G_IN[g]] = min(need[G_IN[g]], c1)
    for d in range(4):
        need[D_IN[d]] = min(need[D_IN[d]], c2)
    out = np.full(N_EVAL, npre, np.int64)
    for kind, i in reversed(EXEC):
        if kind == "g":
            so = need[G_OUT[i]]
            out[15 + i] = so
            need_x[i] = min(need_x[i], suf(s1, so))
            need[G_IN[i]] = min(need[G_IN[i]], so)
            si = need_x[i]
            out[1 + i] = si
            need[G_IN[i]] = min(need[G_IN[i]], suf(s1, si))
            if G_SKIP[i] >= 0:
                need[G_SKIP[i]] = min(need[G_SKIP[i]], si)
        else:
            sd = need[D_OUT[i]]
            out[29 + i] = sd
            need[D_IN[i]] = min(need[D_IN[i]], suf(s2, sd))
    out[0] = need[0]
    return out


def exact_need_sets(order_loc, mask_und, mask_dil, H, W, npre):
    """The exact sets (no suffix approximation): per stage a boolean (npre,) by rank -- what must be evaluated."""
    L = H * W
    rank = np.empty(L, np.int64)
    rank[order_loc] = np.arange(L)

    def reads(ranks_set, mask, dil):   # ranks (bool by rank, full L) -> prefix ranks read through open taps (+ themselves)
        out = np.zeros(L, bool)
        for r in np.nonzero(ranks_set)[0]:
            q = int(order_loc[r])
            y, x = divmod(q, W)
            if r < npre:
                out[r] = True
            for t in range(9):
                if t == 4 or mask[t, q] == 0:
                    continue
                yy, xx = y + dil * (t // 3 - 1), x + dil * (t % 3 - 1)
                if 0 <= yy < H and 0 <= xx < W:
                    out[rank[yy * W + xx]] = True
        out[npre:] = False
        return out
    cols = np.zeros(L, bool)
    cols[npre:] = True
    c1, c2 = reads(cols, mask_und, 1), reads(cols, mask_dil, 2)
    need = [np.zeros(L, bool) for _ in range(19)]
    need_x = [c1.copy() for _ in range(14)]
    for g in range(14):
        need[G_IN[g]] |= c1
    for d in range(4):
        need[D_IN[d]] |= c2
    ev = [None] * N_EVAL
    for kind, i in reversed(EXEC):
        if kind == "g":
            so = need[G_OUT[i]].copy()
            ev[15 + i] = so
            need_x[i] |= reads(so, mask_und, 1)
            need[G_IN[i]] |= so
            si = need_x[i].copy()
            ev[1 + i] = si
            need[G_IN[i]] |= reads(si, mask_und, 1)
            if G_SKIP[i] >= 0:
                need[G_SKIP[i]] |= si
        else:
            sd = need[D_OUT[i]].copy()
            ev[29 + i] = sd
            need[D_IN[i]] |= reads(sd, mask_dil, 2)
    ev[0] = need[0].copy()
    return [e[:npre] for e in ev]
