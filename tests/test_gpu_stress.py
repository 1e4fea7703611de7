"""Randomised sequences of C-ABI calls on one context (single device and a 3-band group): uploads, target changes,
bursts of draws with different transforms and flags, presents into several host images, reads, syncs, timing and
pipelining switches — in any order, under every stream-ordering mode.  Every image the host gets to see (swr_read_*,
swr_present + swr_present_wait) is compared bit for bit with the oracle's frame for that draw.  The context runs three
host threads and three HIP streams per band; this is the test that they never hand out a frame built
from another frame's working set, a half-written framebuffer or a stale scene."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DT, NC = 1, 2


def _mat(rng):
    m = np.eye(4, dtype=np.float32)
    m[0, 0] = rng.uniform(0.6, 1.2); m[1, 1] = rng.uniform(0.6, 1.2)
    m[3, 0] = rng.uniform(-0.2, 0.2); m[3, 1] = rng.uniform(-0.2, 0.2)
    return np.ascontiguousarray(m).reshape(16)


# SWR_STRESS_SEEDS=N adds N more random sequences (seeds 100..) per band count: a longer soak, one pytest run
_EXTRA = [(1 + 2 * (k % 2), 100 + k) for k in range(int(os.environ.get("SWR_STRESS_SEEDS", "0")))]


@pytest.mark.parametrize("env", [{}, {"SWR_LANES": "0"}, {"SWR_LANES": "0", "SWR_EVENT_WAITS": "1"}, {"SWR_LANES": "0", "SWR_HOST_THREADS": "1"}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()) or "default(lanes)")
@pytest.mark.parametrize("n,seed", [(1, 11), (1, 12), (1, 14), (3, 13), (2, 15)] + _EXTRA)
def test_random_call_sequences(swr, oracle, monkeypatch, env, n, seed):
    S = swr.scenes
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(seed)
    scenes = [S.random_soup(4000, 320, 192, 0xA110 + i, r_ndc=0.07, flags=DT, margin=1.1) for i in range(2)]
    sizes = [(320, 192), (256, 160), (192, 96)]
    pool = {}                                   # (W, H) -> list of (colour HostImage, depth HostImage)

    def images(W, H):
        if (W, H) not in pool:
            pool[(W, H)] = [(swr.HostImage((H, W, 4), np.uint8), swr.HostImage((H, W), np.float32)) for _ in range(3)]
        return pool[(W, H)]

    def expect(scene, m, W, H, flags, prim):
        c, d, _, code = oracle.render(scene.vertices, scene.indices, m, W, H, flags & ~NC, primitive_type=prim)
        assert code == 0
        return c, d

    def check(got_c, got_d, want, flags, what):
        wc, wd = want
        if got_c is not None and not (flags & NC):
            assert np.array_equal(got_c, wc), f"{what}: colour differs"
        if got_d is not None:
            assert got_d.tobytes() == wd.tobytes(), f"{what}: depth differs"

    with swr.Context(0, device_count=n) as ctx:
        scene = scenes[0]
        ctx.scene_upload(scene.vertices, scene.indices)
        W, H = sizes[0]
        ctx.target_set(W, H)
        last = None                              # (matrix, flags, prim) of the last draw on this scene / target
        pending = []                             # presents not yet waited for: (image index, frame description)
        for step in range(400):
            op = rng.choice(["draw", "draw", "draw", "burst", "present", "wait", "read", "sync", "upload", "target",
                             "timing", "pipeline", "points"])
            if op in ("draw", "points"):
                prim = 2 if op == "points" else 0            # SWR_PRIMITIVE_VERTICES = 2
                flags = int(rng.choice([0, DT, DT | NC])) if prim == 0 else 0
                m = _mat(rng)
                ctx.draw(m, flags, prim)
                last = (m, flags, prim)
            elif op == "burst":
                for _ in range(int(rng.integers(2, 9))):
                    m = _mat(rng)
                    ctx.draw(m, DT)
                    last = (m, DT, 0)
            elif op == "present" and last is not None and len(pending) < 3:
                used = {i for i, _ in pending}
                i = next(k for k in range(3) if k not in used)
                ci, di = images(W, H)[i]
                ctx.present(ci, di)
                pending.append((i, (scene, last, W, H)))
            elif op == "wait":
                ctx.present_wait()
                for i, (sc, (m, flags, prim), w, h) in pending:
                    ci, di = images(w, h)[i]
                    check(ci.array, di.array, expect(sc, m, w, h, flags, prim), flags, f"step {step}: presented frame")
                pending = []
            elif op == "read" and last is not None:
                m, flags, prim = last
                want = expect(scene, m, W, H, flags, prim)
                check(None, ctx.read_depth(), want, flags, f"step {step}: read_depth")
                if not (flags & NC):
                    check(ctx.read_color(), None, want, flags, f"step {step}: read_color")
            elif op == "sync":
                ctx.sync()
            elif op in ("upload", "target") and not pending:     # both invalidate what a pending present would copy
                if op == "upload":
                    scene = scenes[int(rng.integers(0, 2))]
                    ctx.scene_upload(scene.vertices, scene.indices)
                else:
                    W, H = sizes[int(rng.integers(0, len(sizes)))]
                    ctx.target_set(W, H)
                last = None
            elif op == "timing":
                ctx.timing_enable(int(rng.integers(0, 3)))
            elif op == "pipeline":
                ctx.pipeline_enable(bool(rng.integers(0, 2)))
        ctx.present_wait()
        for i, (sc, (m, flags, prim), w, h) in pending:
            ci, di = images(w, h)[i]
            check(ci.array, di.array, expect(sc, m, w, h, flags, prim), flags, "final wait: presented frame")
    for lst in pool.values():
        for a, b in lst:
            a.free(); b.free()


@pytest.mark.parametrize("n", [1, 3])
def test_destroy_with_frames_and_presents_in_flight(swr, oracle, n):
    """swr_context_destroy right behind a burst of draws and an un-waited present: the helper threads are drained, the
    streams synchronised and the page-locked image is complete (the copy was enqueued before the destroy returned)."""
    S = swr.scenes
    s = S.random_soup(20000, 640, 352, 0xD35, r_ndc=0.05, flags=DT, margin=1.1)
    ci, di = swr.HostImage((s.height, s.width, 4), np.uint8), swr.HostImage((s.height, s.width), np.float32)
    for rep in range(4):
        ctx = swr.Context(0, device_count=n)
        ctx.scene_upload(s.vertices, s.indices)
        ctx.target_set(s.width, s.height)
        for _ in range(12):
            ctx.draw(s.transform, DT)
        ctx.present(ci, di)
        ctx.close()                                # no sync, no present_wait
        rc, rd, _, code = oracle.render_scene(s)
        assert code == 0 and np.array_equal(ci.array, rc) and di.array.tobytes() == rd.tobytes(), f"rep {rep}"
        ci.array[:] = 0; di.array[:] = 0
    ci.free(); di.free()


@pytest.mark.parametrize("level", [1, 2])
def test_long_timed_bursts_reuse_working_sets_correctly(swr, oracle, level):
    """More timed frames than the 64-deep event ring, alternating two transforms, no wait: the ring is drained in the
    middle of swr_draw (a full sync AFTER the frame has been numbered but BEFORE it is posted).  Round 2 had a bug here:
    the sync recorded the not-yet-posted frame as complete, so the binning three frames later did not wait for its
    raster and overwrote the working set under it.  Every presented frame is checked."""
    S = swr.scenes
    s = S.random_soup(60000, 1280, 704, 0x71ED, r_ndc=0.03, flags=DT, margin=1.1)
    W, H = s.width, s.height
    mats = []
    for k in range(2):
        m = np.eye(4, dtype=np.float32)
        m[0, 0] = 0.9 - 0.2 * k; m[1, 1] = 0.8 + 0.25 * k; m[3, 0] = 0.1 * k - 0.05
        mats.append(np.ascontiguousarray(m).reshape(16))
    want = [oracle.render(s.vertices, s.indices, m, W, H, DT)[:2] for m in mats]
    imgs = [(swr.HostImage((H, W, 4), np.uint8), swr.HostImage((H, W), np.float32)) for _ in range(4)]
    with swr.Context() as ctx:
        ctx.scene_upload(s.vertices, s.indices)
        ctx.target_set(W, H)
        ctx.timing_enable(level)
        for rep in range(3):
            shown = []
            for k in range(150):
                ctx.draw(mats[k & 1], DT)
                if k in (60, 63, 66, 149):                    # around the ring drain, and the last frame
                    ci, di = imgs[len(shown)]
                    ctx.present(ci, di)
                    shown.append(k & 1)
            ctx.present_wait()
            for (ci, di), which in zip(imgs, shown):
                assert np.array_equal(ci.array, want[which][0]), f"rep {rep}: colour of a frame with transform {which}"
                assert di.array.tobytes() == want[which][1].tobytes(), f"rep {rep}: depth of a frame with transform {which}"
        ctx.timing_enable(0)
    for a, b in imgs:
        a.free(); b.free()
