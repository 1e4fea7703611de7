"""One context, N bands (swr_config.device_count) and the host-visible gather (swr_present / swr_present_wait,
page-locked host images) — SURVEY.md §8(e), the north star's "framebuffer shards by tile rows across the GPUs of one
node ... final image gathered with pinned hipMemcpyAsync".

The reference has one synchronous draw call that leaves the pixels host-visible on return
(GpuRenderer.swift:35,73,87; Metal+Extensions.swift:57-67; caller App.swift:185); here that ONE call (swr_render) drives
every band.  On a box with fewer GPUs than bands several bands share a GPU — the same code path (sub-contexts, host
threads, copy streams), which is how the 8-band layout of BASELINE configs 4 and 5 is tested at full size on one GPU.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT, NC = 1, 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def same(c, d, rc, rd, what=""):
    if rc is not None:
        bad = np.nonzero((c != rc).any(axis=-1))
        assert bad[0].size == 0, f"{what}: {bad[0].size} colour pixels differ, first at (y,x)=({bad[0][0]},{bad[1][0]})"
    bad = np.nonzero(d.view(np.uint32) != rd.view(np.uint32))
    assert bad[0].size == 0, f"{what}: {bad[0].size} depth values differ, first at (y,x)=({bad[0][0]},{bad[1][0]})"


@pytest.fixture(scope="module")
def cfg4(swr):
    return swr.scenes.cfg4_soup()          # the headline workload: 1 M triangles, 3840x2160, depth-only


def test_cfg4_full_size_through_one_render_call_on_8_bands(swr, oracle, cfg4):
    """cfg4 at full size through ONE swr_render call on a context of 8 bands, bit-exact."""
    s = cfg4
    rc_c, rc_d, st, code = oracle.render_scene(s, oracle.TINV_PER_TRIANGLE)
    assert code == 0 and st.triangles_skipped == 0 and st.fragments > 30_000_000
    with swr.Context(0, device_count=8) as ctx:
        c, d = ctx.render(s.vertices, s.indices, s.transform, s.width, s.height, s.flags)
        bands = ctx.bands()
    assert len(bands) == 8 and bands[0][1] == 0 and bands[-1][2] == s.height
    assert all(a[2] == b[1] for a, b in zip(bands, bands[1:]))                 # contiguous tile-row bands
    assert [(a, b) for _, a, b in bands] == [swr.band_rows(s.height, 8, k) for k in range(8)]   # 68 tile rows over 8 bands
    same(None, d, None, rc_d, "cfg4 x 8 bands")
    # and with the colour image on, same context shape
    with swr.Context(0, device_count=8) as ctx:
        c, d = ctx.render(s.vertices, s.indices, s.transform, s.width, s.height, DT)
    rc_c, rc_d, _, _ = oracle.render(s.vertices, s.indices, s.transform, s.width, s.height, DT | oracle.TINV_PER_TRIANGLE)
    same(c, d, rc_c, rc_d, "cfg4 colour x 8 bands")


def test_cfg5_full_size_on_8_bands(swr, oracle):
    """BASELINE config 5 (262 144 triangles, 7680x4320, z-test, colour) is the other config assigned to 8 GPUs."""
    s = swr.scenes.cfg5_sponza_scale()
    rc_c, rc_d, st, code = oracle.render_scene(s, oracle.TINV_PER_TRIANGLE)
    assert code == 0
    with swr.Context(0, device_count=8) as ctx:
        c, d = ctx.render(s.vertices, s.indices, s.transform, s.width, s.height, s.flags)
    same(c, d, rc_c, rc_d, "cfg5 x 8 bands")


@pytest.mark.parametrize("n", [2, 3, 5])
@pytest.mark.parametrize("flags", [0, DT, 4])
def test_group_matches_oracle_on_soups(swr, oracle, n, flags):
    s = swr.scenes.random_soup(4000, 700, 450, 900 + n, r_ndc=0.08, flags=flags, margin=1.15)
    with swr.Context(0, device_count=n) as ctx:
        c, d = ctx.render(s.vertices, s.indices, s.transform, s.width, s.height, flags)
    if flags == 4:
        rc_c, rc_d, _, code = oracle.render_metal(s.vertices, s.indices, s.transform, s.width, s.height)
    else:
        rc_c, rc_d, _, code = oracle.render(s.vertices, s.indices, s.transform, s.width, s.height, flags)
    assert code == 0
    same(c, d, rc_c, rc_d, f"soup x {n} bands flags={flags}")


def test_more_bands_than_tile_rows_gives_empty_bands(swr, oracle):
    """a band with row_begin == row_end must be a no-op, not a zero-sized grid launch."""
    s = swr.scenes.random_soup(300, 200, 40, 5, r_ndc=0.3, flags=DT)          # 40 rows = 2 tile rows
    with swr.Context(0, device_count=8) as ctx:
        c, d = ctx.render(s.vertices, s.indices, s.transform, 200, 40, DT)
        bands = ctx.bands()
    assert sum(1 for _, a, b in bands if a == b) == 6
    rc_c, rc_d, _, _ = oracle.render(s.vertices, s.indices, s.transform, 200, 40, DT)
    same(c, d, rc_c, rc_d, "empty bands")
    # the single-device context with an empty band of its own
    with swr.Context() as ctx:
        ctx.scene_upload(s.vertices, s.indices)
        ctx.target_set(200, 40, 32, 32)
        ctx.draw(s.transform, DT)
        ctx.sync()
        img = np.full((40, 200), 7.0, np.float32)
        ctx.read_depth(img)
        assert (img == 7.0).all()                                            # nothing written outside the (empty) band


@pytest.mark.parametrize("n", [1, 4])
def test_present_streams_frames_into_pinned_host_images(swr, oracle, n):
    """The resident frame loop with host-visible frames: draw -> present (async D2H, colour and depth together) ->
    next draw renders into the other device framebuffer; two page-locked host image sets alternate."""
    S = swr.scenes
    s = S.cfg2_teapot_scale(640, 360, nu=24, nv=30)
    W, H = 640, 360
    times = [0.1 * k for k in range(7)]
    imgs = [(swr.HostImage((H, W, 4), np.uint8), swr.HostImage((H, W), np.float32)) for _ in range(2)]
    with swr.Context(0, device_count=n) as ctx:
        ctx.scene_upload(s.vertices, s.indices)
        ctx.target_set(W, H)
        got = []
        for k, t in enumerate(times):
            ci, di = imgs[k & 1]
            ctx.draw(S.app_transform(t), DT)
            ctx.present(ci, di)
            if k >= 1:                       # frame k-1 sits in the other image set; wait covers every enqueued copy
                ctx.present_wait()
                pc, pd = imgs[(k - 1) & 1]
                got.append((pc.array.copy(), pd.array.copy()))
        ctx.present_wait()
        got.append((imgs[(len(times) - 1) & 1][0].array.copy(), imgs[(len(times) - 1) & 1][1].array.copy()))
        # a burst without any wait in between: only the last frame must be intact in its image
        for k, t in enumerate(times):
            ctx.draw(S.app_transform(t), DT)
            ctx.present(*imgs[0])
        ctx.present_wait()
        last = (imgs[0][0].array.copy(), imgs[0][1].array.copy())
    for k, t in enumerate(times):
        rc_c, rc_d, _, _ = oracle.render(s.vertices, s.indices, S.app_transform(t), W, H, DT)
        same(got[k][0], got[k][1], rc_c, rc_d, f"present frame {k} (n={n})")
    same(last[0], last[1], rc_c, rc_d, "burst, last frame")
    for a, b in imgs:
        a.free(); b.free()


def test_present_into_pageable_and_registered_memory(swr, oracle):
    s = swr.scenes.random_soup(2500, 1280, 720, 31, r_ndc=0.06, flags=DT)
    rc_c, rc_d, _, _ = oracle.render(s.vertices, s.indices, s.transform, 1280, 720, DT)
    with swr.Context(0, device_count=3) as ctx:
        ctx.scene_upload(s.vertices, s.indices)
        ctx.target_set(1280, 720)
        ctx.draw(s.transform, DT)
        c = np.full((720, 1280, 4), 0xEE, np.uint8)                           # pageable: staged through pinned chunks
        d = np.full((720, 1280), -5.0, np.float32)
        ctx.present(c, d)
        ctx.present_wait()
        same(c, d, rc_c, rc_d, "pageable destination")
        c2 = np.zeros((720, 1280, 4), np.uint8)
        d2 = np.zeros((720, 1280), np.float32)
        swr.host_register(c2); swr.host_register(d2)                          # hipHostRegister: true async DMA
        try:
            ctx.draw(s.transform, DT)
            ctx.present(c2, d2)
            ctx.present_wait()
        finally:
            swr.host_unregister(c2); swr.host_unregister(d2)
        same(c2, d2, rc_c, rc_d, "registered destination")
        # depth-only frame: the colour pointer is ignored
        ctx.draw(s.transform, DT | NC)
        c3 = np.full((720, 1280, 4), 0x11, np.uint8)
        ctx.present(c3, d)
        ctx.present_wait()
        assert (c3 == 0x11).all()
        same(None, d, None, rc_d, "depth-only present")


def test_present_bookkeeping_of_the_binding(swr):
    """binding.Context.present keeps every destination alive (and a HostImage un-freeable) while a copy may be in flight:
    one entry per image however many frames present it, and nothing left behind by a present the library refused."""
    s = swr.scenes.random_soup(500, 320, 200, 5, r_ndc=0.1, flags=DT)
    ci, di = swr.HostImage((200, 320, 4), np.uint8), swr.HostImage((200, 320), np.float32)
    with swr.Context(0) as ctx:
        with pytest.raises(swr.SwrError):
            ctx.present(ci, di)                       # no target yet: SWR_ERR_NO_SCENE, nothing enqueued
        assert not ctx._present_refs and not ci._busy and not di._busy
        ctx.scene_upload(s.vertices, s.indices)
        ctx.target_set(320, 200)
        with pytest.raises(swr.SwrError):
            ctx.present(None, None)
        assert not ctx._present_refs
        for _ in range(50):
            ctx.draw(s.transform, DT)
            ctx.present(ci, di)
        assert len(ctx._present_refs) == 2 and ci._busy and di._busy
        with pytest.raises(swr.SwrError):
            ci.free()                                 # a copy into it is (or may be) in flight
        ctx.present_wait()
        assert not ctx._present_refs and not ci._busy and not di._busy
    ci.free(); di.free()


def big_scene(swr, ntri=220):
    """A few hundred near-full-screen triangles at 4K: ~4 000 tiles each, far more (triangle,tile) pairs than the
    initial bin capacity (2 x triangles + 65 536)."""
    return swr.scenes.random_soup(ntri, 3840, 2160, 77, r_ndc=1.4, flags=DT, margin=0.3)


def crowded_scene(swr, ntri=20000):
    """ntri small triangles inside two neighbouring tiles of a 1280x720 frame: far more entries than the initial tile
    region of the fixed-stride bins (k_bin) holds."""
    s = swr.scenes.random_soup(ntri, 1280, 720, 555, r_ndc=0.01, flags=DT, margin=1.0)
    v = s.vertices.copy()
    v[:, 0] = 0.30 + (v[:, 0] * 0.5 + 0.5) * 0.07
    v[:, 1] = 0.10 + (v[:, 1] * 0.5 + 0.5) * 0.06
    s.vertices = np.ascontiguousarray(v)
    return s


@pytest.mark.parametrize("bins", ["fixed", "exact"])
def test_bin_overflow_is_repaired_or_reported(swr, oracle, bins):
    """Both bin layouts: k_bin's fixed tile regions (a crowded tile overflows its region) and the exact-size bins of the
    four-kernel path (more (triangle,tile) pairs than the list holds)."""
    if bins == "exact":
        s = big_scene(swr)
    else:
        s = crowded_scene(swr)
    rc_c, rc_d, _, _ = oracle.render(s.vertices, s.indices, s.transform, s.width, s.height, DT | NC | oracle.TINV_PER_TRIANGLE)
    # (1) draw + present + wait: the overflowing frame is redrawn with grown bins and copied again
    with swr.Context() as ctx:
        if bins == "exact":
            ctx.debug_set(swr.binding.DEBUG_BIN_MODE, swr.binding.BIN_MODE_EXACT)
        ctx.scene_upload(s.vertices, s.indices)
        ctx.target_set(s.width, s.height)
        d = swr.HostImage((s.height, s.width), np.float32)
        ctx.draw(s.transform, DT | NC)
        ctx.present(None, d)
        ctx.present_wait()
        same(None, d.array, None, rc_d, "overflow repaired at present_wait")
        assert ctx.timings()["tile_pairs"] > (2 * s.triangles + 65536 if bins == "exact" else s.triangles)
        d.free()
    # (2) an un-waited burst of PRESENTED frames: the earlier overflowing frame was rastered empty and copied to the
    # host like that -> reported once, bins grown, the last frame repaired
    with swr.Context() as ctx:
        if bins == "exact":
            ctx.debug_set(swr.binding.DEBUG_BIN_MODE, swr.binding.BIN_MODE_EXACT)
        ctx.scene_upload(s.vertices, s.indices)
        ctx.target_set(s.width, s.height)
        d = swr.HostImage((s.height, s.width), np.float32)
        for _ in range(2):
            ctx.draw(s.transform, DT | NC)
            ctx.present(None, d)
        with pytest.raises(swr.SwrError) as e:
            ctx.present_wait()
        assert e.value.code == -8                                             # SWR_ERR_FRAME_DROPPED
        ctx.present_wait()                                                    # the burst's last frame was redrawn and copied again
        same(None, d.array, None, rc_d, "last frame of the burst, repaired")
        ctx.draw(s.transform, DT | NC)
        ctx.sync()
        same(None, ctx.read_depth(), None, rc_d, "after the reported drop")
        d.free()
    # (2b) the same burst without any present: nobody can have seen the empty frame -> no error, last frame repaired
    with swr.Context() as ctx:
        if bins == "exact":
            ctx.debug_set(swr.binding.DEBUG_BIN_MODE, swr.binding.BIN_MODE_EXACT)
        ctx.scene_upload(s.vertices, s.indices)
        ctx.target_set(s.width, s.height)
        ctx.draw(s.transform, DT | NC)
        ctx.draw(s.transform, DT | NC)
        ctx.sync()
        same(None, ctx.read_depth(), None, rc_d, "burst without presents")
    # (3) the same through a group and swr_render
    with swr.Context(0, device_count=2) as ctx:
        if bins == "exact":
            ctx.debug_set(swr.binding.DEBUG_BIN_MODE, swr.binding.BIN_MODE_EXACT)
        _, d = ctx.render(s.vertices, s.indices, s.transform, s.width, s.height, DT | NC)
        same(None, d, None, rc_d, "overflow inside a group")


def test_product_library_has_no_ablation_switch(swr, oracle):
    """SWR_DEBUG_VARIANT selected timing-only k_raster instantiations with invalid results; the
    product library no longer contains them (only `make ablation` builds them, into another .so)."""
    code = (
        "import numpy as np, swr_amd\n"
        "s = swr_amd.scenes.random_soup(3000, 640, 360, 5, r_ndc=0.05, flags=1)\n"
        "with swr_amd.Context() as ctx:\n"
        "    c, d = ctx.render(s.vertices, s.indices, s.transform, 640, 360, 1)\n"
        "np.save(r'%s', d)\n")
    out = os.path.join(ROOT, "gpurun_out", "variant4_depth.npy") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) \
        else "/tmp/variant4_depth.npy"
    r = subprocess.run([sys.executable, "-c", code % out], env={**os.environ, "SWR_DEBUG_VARIANT": "4"}, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    s = swr.scenes.random_soup(3000, 640, 360, 5, r_ndc=0.05, flags=1)
    _, rd, _, _ = oracle.render(s.vertices, s.indices, s.transform, 640, 360, 1)
    assert np.load(out).tobytes() == rd.tobytes()
    blob = open(swr.library_path(), "rb").read()
    assert b"SWR_DEBUG_VARIANT" not in blob


def test_group_resident_path_and_timings(swr, oracle):
    s = swr.scenes.random_soup(5000, 1024, 768, 12, r_ndc=0.05, flags=DT)
    with swr.Context(0, device_count=4) as ctx:
        ctx.scene_upload(s.vertices, s.indices)
        ctx.target_set(1024, 768)
        ctx.timing_enable(1)
        ctx.timing_reset()
        for _ in range(12):
            ctx.draw(s.transform, DT)
        sums, n = ctx.timing_totals()
        assert n == 12 and sums["raster_ms"] > 0
        t = ctx.timings()
        assert t["tiles"] == 16 * 24 and t["tile_pairs"] > 5000
        ctx.timing_enable(0)
        c, d = ctx.read_color(), ctx.read_depth()
        # a band of a group: the group's own target may itself be a band of a larger image
        ctx.target_set(1024, 768, 256, 640)
        ctx.draw(s.transform, DT)
        c2 = np.zeros_like(c); d2 = np.full_like(d, 3.0)
        ctx.read_color(c2); ctx.read_depth(d2)
    rc_c, rc_d, _, _ = oracle.render(s.vertices, s.indices, s.transform, 1024, 768, DT)
    same(c, d, rc_c, rc_d, "group resident")
    assert np.array_equal(c2[256:640], rc_c[256:640]) and d2[256:640].tobytes() == rc_d[256:640].tobytes()
    assert (d2[:256] == 3.0).all() and (d2[640:] == 3.0).all() and not c2[:256].any()


@pytest.mark.parametrize("env", [{}, {"SWR_LANES": "0"}, {"SWR_LANES": "0", "SWR_EVENT_WAITS": "1"}, {"SWR_LANES": "0", "SWR_HOST_THREADS": "1"},
                                 {"SWR_LANES": "0", "SWR_BIND_EVENTS": "0"}, {"SWR_PIPELINE": "0"},
                                 {"SWR_LANES": "0", "SWR_EVENT_WAITS": "1", "SWR_BIND_EVENTS": "0"}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()) or "default(lanes)")
@pytest.mark.parametrize("n", [1, 3])
def test_every_frame_of_an_unwaited_burst_is_intact_under_every_ordering_mode(swr, oracle, monkeypatch, env, n):
    """Eight frames with eight different transforms, drawn and presented into eight host image sets without a single
    wait in between: three working sets and the device framebuffers are re-used while earlier frames are still in
    flight.  Frame lanes (the default: every frame on its own stream, nothing else orders it), or the two-stream pipeline of
    rounds 1-3 (SWR_LANES=0) ordered by the helper threads' event polls, by event waits on the streams
    (SWR_EVENT_WAITS=1 / SWR_HOST_THREADS=1), with recorded instead of kernel-bound events, or one stream (SWR_PIPELINE=0);
    the pixels must not depend on which (include/swr.h: the knobs of INTEGRATION.md §7 only change timing)."""
    S = swr.scenes
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    s = S.random_soup(30000, 640, 352, 0xB0B5, r_ndc=0.05, flags=DT, margin=1.1)
    W, H = s.width, s.height
    mats = []
    for k in range(8):                            # column-major: scale about the centre + a shift, w = 1
        m = np.eye(4, dtype=np.float32)
        m[0, 0] = 0.75 + 0.05 * k; m[1, 1] = 1.1 - 0.04 * k
        m[3, 0] = 0.03 * k - 0.1; m[3, 1] = 0.05 - 0.02 * k
        mats.append(np.ascontiguousarray(m).reshape(16))
    imgs = [(swr.HostImage((H, W, 4), np.uint8), swr.HostImage((H, W), np.float32)) for _ in mats]
    with swr.Context(0, device_count=n) as ctx:
        ctx.scene_upload(s.vertices, s.indices)
        ctx.target_set(W, H)
        for rep in range(3):                      # three bursts: the second and third start with everything warm
            for m, (ci, di) in zip(mats, imgs):
                ctx.draw(m, DT)
                ctx.present(ci, di)
            ctx.present_wait()
    for k, m in enumerate(mats):
        rc_c, rc_d, _, code = oracle.render(s.vertices, s.indices, m, W, H, DT)
        assert code == 0
        same(imgs[k][0].array, imgs[k][1].array, rc_c, rc_d, f"burst frame {k} ({env or 'default'}, n={n})")
    for a, b in imgs:
        a.free(); b.free()


@pytest.mark.gpu
@pytest.mark.parametrize("bands", [2, 5])
def test_full_size_one_shot_render_through_bands_equals_one_context(swr, bands):
    """cfg4 at full size (1 M triangles, 4K) through swr_render WITHOUT a scene identity on a banded context: every band uploads
    the scene for one frame (index order: nothing to cull by, exact-size bins) — the image equals the one a single context draws
    from its sorted, resident scene, bit for bit (order-independent visibility keys)."""
    sc = swr.scenes.cfg4_soup(depth_only=False)
    with swr.Context() as one:
        c_ref, d_ref = one.render(sc.vertices, sc.indices, sc.transform, sc.width, sc.height, 1, scene_id=5)
    with swr.Context(0, device_count=bands) as grp:
        c, d = grp.render(sc.vertices, sc.indices, sc.transform, sc.width, sc.height, 1, scene_id=0)
        assert grp.render_timings()["scene_cached"] == 0
        assert np.array_equal(c, c_ref) and d.tobytes() == d_ref.tobytes()
        c, d = grp.render(sc.vertices, sc.indices, sc.transform, sc.width, sc.height, 1, scene_id=9)
        assert np.array_equal(c, c_ref) and d.tobytes() == d_ref.tobytes()
