"""CPU: the dependency-cone oracle (oracle/prefix_cone_oracle.py) -- the per-stage start ranks the prefix pass uses are a
superset of the exact sets of items anybody reads, on the generation orders of the synthetic background masks."""
import numpy as np
import pytest

from oracle import c_oracle, prefix_cone_oracle as pc
from pixelsynth_amd import synthetic as syn


def _case(name):
    info = c_oracle.masks_for_background(syn.background_masks(256)[name], 32)
    order = info["order"]
    order_loc = (order[:, 0] * 32 + order[:, 1]).astype(np.int64)
    rank = np.empty(1024, np.int64)
    rank[order_loc] = np.arange(1024)
    bg = info["bg32"].reshape(-1) > 0
    first = int(rank[bg].min()) if bg.any() else 1024
    return order_loc, info["mask_undilated"][0], info["mask_dilated"][0], first


@pytest.mark.parametrize("name", ["right_half", "ragged", "top_band", "half_plus_island"])
def test_start_ranks_cover_the_exact_cone(name):
    order_loc, mu, md, first = _case(name)
    for npre in (first, max(first - 37, 1)):
        starts = pc.prefix_starts(order_loc, mu, md, 32, 32, npre)
        exact = pc.exact_need_sets(order_loc, mu, md, 32, 32, npre)
        assert starts.shape == (pc.N_EVAL,) and (starts >= 0).all() and (starts <= npre).all()
        for sid in range(pc.N_EVAL):
            need = np.nonzero(exact[sid])[0]
            if len(need):
                assert starts[sid] <= need.min(), (name, sid)
        # the last stage's output (node d9) is read by no prefix item: nothing of the final conv_out is evaluated
        assert starts[15 + 13] == npre
        # the one-number-per-stage form gives away little against the exact sets
        kept = sum(npre - int(s) for s in starts)
        exact_n = sum(int(e.sum()) for e in exact)
        assert kept >= exact_n and kept <= 1.15 * exact_n + 64


def test_no_prefix_and_full_prefix():
    order_loc, mu, md, _ = _case("right_half")
    assert (pc.prefix_starts(order_loc, mu, md, 32, 32, 0) == 0).all()
    s = pc.prefix_starts(order_loc, mu, md, 32, 32, 1024)   # nothing is walked: nobody reads anything
    assert (s == 1024).all()
