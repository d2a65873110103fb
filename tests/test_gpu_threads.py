"""Contexts used from several host threads at once (one context per thread, one device): the
library's process-wide state -- the cache of kernel residencies, the error text, the lazily built
tables -- must not couple them.  Every thread's planes and denoised frame equal, bit for bit, what
the same calls give when the threads run one after the other."""
import os
import threading

import numpy as np
import pytest

from conftest import PIN_SEED

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JOBS = [("cornell_crystal", 96, 54, 4), ("MultiLightGlass", 80, 45, 8), ("Box_Grid", 64, 64, 4),
        ("blur_test", 50, 30, 12), ("hierarchy", 96, 54, 4), ("bars3", 33, 17, 4)]


def _job(name, W, H, depth, out, slot, rounds):
    import ptk
    import scenes
    try:
        weights = ptk.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
        res = []
        for r in range(rounds):
            ctx = ptk.Context(0)
            if name == "hierarchy":
                ctx.set_scene(scenes.random_scene(700, W, H, seed=5, max_depth=depth), want_features=1)
            else:
                raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", name + ".scene"), W, H,
                                                max_depth=depth)
                ctx.set_constants(raw, want_features=1)
            ctx.render(3, ptk.SEED_REFERENCE, PIN_SEED)
            planes = ctx.read_planes().copy()
            exposure = ctx.auto_exposure(ptk.EXPOSURE_CPU)[0]   # (the fp32 result; the raw sum's last bit may depend on the order of its atomics)
            d = ctx.denoise(weights, n=1 + r % 3, variant=0)
            post, _ = ctx.post_process(metric=1, exposure=1.0, g=2.2)
            res.append((planes, exposure, d["denoised"].copy(), d["variances"].copy(), post.copy()))
            ctx.close()
        out[slot] = res
    except Exception as exc:   # reported by the test
        out[slot] = exc


def _equal(a, b):
    if isinstance(a, np.ndarray):
        return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
    return a == b or (a != a and b != b)


def test_contexts_on_six_threads_give_what_they_give_one_after_the_other():
    rounds = 4
    serial = [None] * len(JOBS)
    for i, j in enumerate(JOBS):
        _job(*j, serial, i, rounds)
    together = [None] * len(JOBS)
    threads = [threading.Thread(target=_job, args=(*j, together, i, rounds)) for i, j in enumerate(JOBS)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
        assert not t.is_alive()
    for i, j in enumerate(JOBS):
        assert not isinstance(serial[i], Exception), serial[i]
        assert not isinstance(together[i], Exception), together[i]
        for r in range(rounds):
            for a, b in zip(serial[i][r], together[i][r]):
                assert _equal(a, b), (j, r)
