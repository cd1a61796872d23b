"""Input generators the fixture generator (make_golden.py, which needs the reference) and the tests share -- importable
without /root/reference."""
import numpy as np

from pixelsynth_amd import synthetic as syn


def zbuffer_cases(seeds=None):
    """inputs of the hard z-buffer fixture: (name, W, depth (B,1,W,W), cams, RT2).  W = 256 is what the reference's literal
    128 / 255 pixel mapping assumes.  seeds: {name: int}, chosen by the generator so that no two points of a frame have the same
    projected z (torch's sort is not stable: a tie could go either way) and stored in the fixture."""
    seeds = seeds or {}
    out = []
    for name, cams, yaw, pitch, lo, hi in (("demo_R", syn.demo_cameras(2), 0.6, 0.0, 1.0, 100.0), ("demo_small", syn.demo_cameras(2), 0.05, 0.02, 1.0, 100.0),
                                           ("mp3d_yaw", syn.mp3d_cameras(2), 0.4, 0.1, 0.5, 10.0), ("eps", syn.demo_cameras(2), 1.5, 0.0, 0.001, 0.02)):
        W = 256
        rs = np.random.RandomState(len(name) * 7 + 1 + 1000 * int(seeds.get(name, 0)))
        # distinct depths (a random permutation of an even ladder), so that the z order has no ties the sort could break either way
        d = np.stack([(lo + (hi - lo) * (rs.permutation(W * W) + 0.5) / (W * W)) for _ in range(2)]).reshape(2, 1, W, W).astype(np.float32)
        d[1] = syn.depth_smooth(3, 1, W, lo, hi)[0] + d[1] * 1e-3
        RTinv, RT = syn.yaw_pose(cams["P"], yaw, pitch=pitch)
        out.append((name, W, d, cams, RT))
    return out
