"""CPU: the host logic that lets several batches share the column launches (lmconv.model.pack_launches, what
z_buffermodel.outpaint_pipelined runs): whatever the depth, every column of every batch runs exactly once, a batch's waves keep their
order -- a column of wave w + 1 never shares a launch with, or precedes, a column of wave w of the same batch -- no launch takes more than
the capacity, and with a few batches in flight the launches are full."""
import numpy as np
import pytest

from pixelsynth_amd.lmconv.model import pack_launches


def schedule(rs, n_waves, width):
    sizes = rs.randint(1, width + 1, size=n_waves)
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)


def run_pipeline(schedules, depth, cap):
    """outpaint_pipelined's loop on the host -> (launches: lists of (batch, column) pairs, completion order, launches per call)."""
    inflight, launches, finished, per_call = [], [], [], []

    def step(drain=False):
        n0 = len(launches)
        while inflight:
            full = len(inflight) >= depth
            slices, starts = pack_launches(inflight, cap, budget=None if (drain or full) else -(-(len(inflight[-1]["ws"]) - 1) // depth))
            flat = [(inflight[k]["id"], c) for k, a, b in slices for c in range(a, b)]
            assert starts[-1] == len(flat)
            launches.extend(flat[starts[j]:starts[j + 1]] for j in range(len(starts) - 1))
            finished.extend(b["id"] for b in inflight if b["w"] >= len(b["ws"]) - 1)
            inflight[:] = [b for b in inflight if b["w"] < len(b["ws"]) - 1]
            if not drain:
                break
        per_call.append(len(launches) - n0)
    for i, ws in enumerate(schedules):
        assert len(inflight) < depth          # (a share of the handle is free when a batch arrives)
        inflight.append(dict(ws=ws, w=0, off=0, id=i))
        step()
    step(drain=True)
    return launches, finished, per_call


@pytest.mark.parametrize("depth", [2, 3, 4, 6])
@pytest.mark.parametrize("width,cap", [(40, 128), (300, 1024), (1024, 1024)])
def test_batches_in_flight_keep_every_dependency(depth, width, cap):
    rs = np.random.RandomState(100 * depth + width)
    schedules = [schedule(rs, int(rs.randint(1, 60)), width) for _ in range(9)]
    launches, finished, _ = run_pipeline(schedules, depth, cap)
    assert sorted(finished) == list(range(9))
    seen = sorted(x for l in launches for x in l)
    assert seen == sorted((i, c) for i, ws in enumerate(schedules) for c in range(ws[-1]))      # every column exactly once
    assert max(len(l) for l in launches) <= cap and min(len(l) for l in launches) >= 1
    for i, ws in enumerate(schedules):          # wave w + 1 of a batch starts strictly behind the last launch of its wave w
        wave_of = np.repeat(np.arange(len(ws) - 1), np.diff(ws))
        first, last = {}, {}
        for k, l in enumerate(launches):
            for b, c in l:
                if b == i:
                    first.setdefault(int(wave_of[c]), k)
                    last[int(wave_of[c])] = k
        assert all(first[w + 1] > last[w] for w in range(len(ws) - 2))


def test_launches_are_full_with_a_few_batches_in_flight():
    """Equal batches whose waves grow and shrink (an AR run's do): with four in flight the steady-state launches hold nearly the capacity,
    so a step needs about columns / capacity of them; one batch at a time needs a launch per wave."""
    sizes = np.concatenate([np.linspace(20, 100, 30), np.linspace(100, 4, 50)]).astype(int)
    ws = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    launches, finished, per_call = run_pipeline([ws.copy() for _ in range(16)], 4, 128)
    assert finished == list(range(16))
    steady = per_call[6:-1]
    assert max(steady) <= 1.15 * ws[-1] / 128 + 1, (steady, ws[-1] / 128)
    assert np.mean([len(l) for l in launches[len(launches) // 3: 2 * len(launches) // 3]]) > 0.93 * 128
    alone, _, _ = run_pipeline([ws.copy()], 4, 128)
    assert len(alone) == len(sizes)


def test_empty_and_single_wave_schedules():
    empty = dict(ws=np.zeros(1, np.int32), w=0, off=0)
    assert pack_launches([empty], 128)[0] == []
    one = dict(ws=np.array([0, 300], np.int32), w=0, off=0)
    slices, starts = pack_launches([one], 128)
    assert slices == [(0, 0, 128), (0, 128, 256), (0, 256, 300)] and list(starts) == [0, 128, 256, 300] and one["w"] == 1
