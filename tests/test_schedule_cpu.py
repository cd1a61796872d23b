"""CPU: the host logic that lets several batches share the column launches (lmconv.model.split_tail / split_parts / merge_schedules /
fold_schedules, what z_buffermodel.outpaint_pipelined runs): whatever the depth, every column of every batch runs exactly once, a batch's
waves keep their order -- a column of wave w + 1 never shares a launch with, or precedes, a column of wave w of the same batch -- and a
launch takes no more than the capacity unless one wave of the newest batch alone exceeds it."""
import numpy as np
import pytest

from pixelsynth_amd.lmconv.model import fold_schedules, merge_schedules, split_parts, split_tail


def schedule(rs, n_waves, width, batch):
    """A batch's schedule: waves of 1 .. width columns; a column = (frame, rank) with frame = 1000 batch + wave (so that a column says
    which batch and wave it belongs to)."""
    sizes = rs.randint(1, width + 1, size=n_waves)
    ws = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    cols = np.stack([np.repeat(1000 * batch + np.arange(n_waves), sizes), np.arange(ws[-1])], 1).astype(np.int32)
    return cols, ws


def run_pipeline(batches, depth, cap, merge_max):
    """outpaint_pipelined's loop on the host: -> list of launches (arrays of columns), in order."""
    inflight, launches = [], []

    def step():
        merged = fold_schedules([b.pop(0) for b in inflight], cap)
        inflight[:] = [b for b in inflight if b]
        if merged is not None:
            cols, ws = merged
            launches.extend(cols[ws[k]:ws[k + 1]] for k in range(len(ws) - 1) if ws[k + 1] > ws[k])
    for cols, ws in batches:
        bounds = split_parts(ws, depth, merge_max)
        assert bounds[0] == 0 and bounds[-1] == len(ws) - 1 and all(a <= b for a, b in zip(bounds, bounds[1:]))
        inflight.append([(cols[ws[a]:ws[b]], ws[a:b + 1] - ws[a]) for a, b in zip(bounds[:-1], bounds[1:])])
        step()
    while inflight:
        step()
    return launches


@pytest.mark.parametrize("depth", [2, 3, 4, 6])
@pytest.mark.parametrize("width,cap", [(40, 128), (300, 1024), (1024, 1024)])
def test_batches_in_flight_keep_every_dependency(depth, width, cap):
    rs = np.random.RandomState(100 * depth + width)
    batches = [schedule(rs, int(rs.randint(1, 60)), width, b) for b in range(7)]
    launches = run_pipeline(batches, depth, cap, merge_max=min(720, cap * 45 // 64))
    allc = np.concatenate(launches)
    want = np.concatenate([c for c, _ in batches])
    assert len(allc) == len(want) and np.array_equal(allc[np.lexsort((allc[:, 1], allc[:, 0]))], want[np.lexsort((want[:, 1], want[:, 0]))])
    last_launch_of_wave, first_launch_of_wave = {}, {}
    for k, cols in enumerate(launches):
        for key in np.unique(cols[:, 0]):
            first_launch_of_wave.setdefault(int(key), k)
            last_launch_of_wave[int(key)] = k
    for key, first in first_launch_of_wave.items():
        if key % 1000 and key - 1 in last_launch_of_wave:      # wave w of a batch starts strictly behind the last launch of its wave w - 1
            assert first > last_launch_of_wave[key - 1], (key, first, last_launch_of_wave[key - 1])
    biggest_wave = max(int(np.diff(ws).max()) for _, ws in batches)
    assert max(len(c) for c in launches) <= max(cap, biggest_wave)
    own = sum(len(ws) - 1 for _, ws in batches)
    if depth > 2 and width * 3 <= cap:
        assert len(launches) < 0.75 * own          # narrow waves: the batches really share launches


def test_two_parts_are_head_and_tail_and_empty_parts_are_harmless():
    ws = np.array([0, 10, 30, 60, 100, 130, 150, 160, 164, 166], np.int32)      # grows to 40, shrinks to 2
    cut = split_tail(ws, 25)
    assert split_parts(ws, 2, 25) == [0, cut, 9] and np.diff(ws)[cut:].max() <= 25 and np.diff(ws)[cut - 1] > 25
    assert split_parts(ws, 3, 25) == [0, 3, 6, 9]
    one = split_parts(np.array([0, 5], np.int32), 4, 25)              # one wave, four parts: three of them empty
    assert one[0] == 0 and one[-1] == 1 and sorted(one) == one and len(one) == 5
    empty = (np.zeros((0, 2), np.int32), np.zeros(1, np.int32))
    cols, w = schedule(np.random.RandomState(1), 3, 5, 0)
    assert fold_schedules([], 128) is None
    m = fold_schedules([empty, (cols, w), empty], 128)
    assert np.array_equal(m[0], cols) and np.array_equal(m[1], w)
    m2 = merge_schedules(cols, w, empty[0], empty[1], 128)           # a tail with no head: launches of its own
    assert np.array_equal(m2[0], cols) and np.array_equal(m2[1], w)
