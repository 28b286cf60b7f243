"""16-bit UNet (throughput paths SH_UNET_BF16 and SH_UNET_F16): logits against the float64 evaluation of the same
network on the same image, within the tolerance stated here per element type; mask identical wherever the logit is not
within that type's error band of zero; downstream landmarks stay close to the f32-path result."""
import numpy as np
import pytest

from oracle import unet as o_unet
from shoulder_amd import _lib

pytestmark = pytest.mark.gpu
LOGIT_ABS_TOL = 0.08      # bf16 activations/weights (8 significant bits) through 23 layers; measured max ~0.03
BAND = 0.10               # pixels with |logit_f64| > BAND must get the same mask value
# per element type: (enumerator, logit tolerance, mask band, plane-point bound mm, axis bound mm); f16 carries 11 significant bits
DTYPES = {"bf16": (_lib.UNET_BF16, LOGIT_ABS_TOL, BAND, 0.3, 1.0), "f16": (_lib.UNET_F16, 0.012, 0.015, 0.06, 0.3)}


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_bf16_logits_and_mask(engine, oracle_bones, unet_weights, name):
    dtype, LOGIT_ABS_TOL, BAND, plane_mm, axis_mm = DTYPES[name]
    h = oracle_bones("humerus_left")
    engine.upload([(h.verts, h.faces)])
    engine.store("obb_transform", h.T_obb[None])
    stages = _lib.STAGE_ALL & ~_lib.STAGE_OBB
    engine.set_params(unet_dtype=_lib.UNET_F32)
    lm32 = engine.run(stages)[0].copy()
    try:
        engine.set_params(unet_dtype=dtype)
        engine.set_keep_products(True)      # (the 16-bit network scales the raw image itself; "anp.image" is written on request)
        lm16 = engine.run(stages)[0].copy()
        engine.set_keep_products(False)
        img = engine.fetch("anp.image", np.float32, (1, 512, 512))[0]
        lg = engine.fetch("anp.logits", np.float32, (1, 512, 512))[0]
    finally:
        engine.set_params(unet_dtype=_lib.UNET_F32)
    ref = o_unet.forward_f64(unet_weights, img)
    err = np.abs(lg - ref)
    print("%s logits: max abs err %.4f, mean %.5f; plane point moves %.4f mm, central axis %.4f mm" % (
        name, err.max(), err.mean(), np.abs(lm16["anp_plane_point"] - lm32["anp_plane_point"]).max(),
        np.abs(lm16["anp_axis_central"] - lm32["anp_axis_central"]).max()))
    assert err.max() < LOGIT_ABS_TOL
    sure = np.abs(ref) > BAND
    assert ((lg > 0) == (ref > 0))[sure].all()
    assert lm16["status"] == 0
    # groove / canal do not depend on the network
    np.testing.assert_array_equal(lm16["groove_axis"], lm32["groove_axis"])
    # the neck plane moves by far less than a pixel of the 512x512 image (~0.3 mm)
    assert np.abs(lm16["anp_plane_point"] - lm32["anp_plane_point"]).max() < plane_mm
    assert np.abs(lm16["anp_axis_central"] - lm32["anp_axis_central"]).max() < axis_mm
    assert abs(int(lm16["n_anp"]) - int(lm32["n_anp"])) < 0.1 * int(lm32["n_anp"])


@pytest.mark.parametrize("H,W", [(256, 512), (512, 512)])
def test_unet_alone_config5(engine, unet_weights, H, W):
    """BASELINE configs[4] / SURVEY 8(d) config 5: the network alone on [B,1,256,512] and [B,1,512,512] inputs U(0,1), seed
    1234, through sh_unet_infer: fp16 MFMA conv (SH_UNET_F16) and bf16 against the float64 evaluation within each type's
    tolerance; the f32 path bit-exact against the C fma-chain restatement."""
    rng = np.random.default_rng(1234)
    img = rng.random((2, H, W), dtype=np.float32)
    engine.set_params(unet_dtype=_lib.UNET_F32)
    try:
        lo32 = engine.unet_infer(img)
        engine.set_params(unet_dtype=_lib.UNET_BF16)
        lo_bf = engine.unet_infer(img)
        engine.set_params(unet_dtype=_lib.UNET_F16)
        lo_h = engine.unet_infer(img)
    finally:
        engine.set_params(unet_dtype=_lib.UNET_F32)
    ref = np.stack([o_unet.forward_chain(unet_weights, img[i]) for i in range(2)])
    assert lo32.dtype == np.float32 and lo32.shape == img.shape
    assert np.array_equal(lo32, ref)                      # one fma chain per output: bit for bit
    ref64 = o_unet.forward_f64(unet_weights, img[0])
    e_bf, e_h = float(np.abs(lo_bf[0] - ref64).max()), float(np.abs(lo_h[0] - ref64).max())
    print(f"{H}x{W}: max |logit - f64|  bf16 {e_bf:.4f}  f16 {e_h:.5f}  f32 {float(np.abs(lo32[0] - ref64).max()):.2e}")
    assert e_bf < DTYPES["bf16"][1] and e_h < DTYPES["f16"][1]
    assert float(np.abs(lo_bf - ref).max()) < 0.08 and float(np.abs(lo_h - ref).max()) < DTYPES["f16"][1]


def test_unet_infer_rejects_bad_shapes(engine):
    with pytest.raises(Exception):
        engine.unet_infer(np.zeros((1, 250, 512), np.float32))      # not a multiple of 16 << depth
    with pytest.raises(ValueError):
        engine.unet_infer(np.zeros((512, 512), np.float32))


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_production_network_against_the_reference_network(engine, name):
    """The production 16-bit network -- level 0 as three fused ping-pong kernels (k_unet16_pp.h: image -> enc0a -> LDS -> enc0b -> skip0 +
    pool; up0 inside dec0a; dec0b + head), the >= 64-channel convs on the persistent LDS-DMA kernel with loader waves (k_unet16_ldr.h,
    pools in its epilogue), the up-convolutions with register-resident pixels -- against the REFERENCE network of the same library
    (SHOULDER_UNET_REFERENCE=1 when the context is created: layer by layer on the generic two-barrier kernels, nothing fused).  Same
    weights, same rounding points except two: the fused first conv runs on the matrix cores with ET weights and a hi + lo split image,
    and the fused head sums a pixel's 32 products from unrounded activations.  So the skip tensors agree to a few ulps of the element
    type (an activation that rounds the other way moves the outputs behind it), the logits far inside the type's tolerance -- incl.
    ragged last work ranges (5 images of 256 x 256 over 256 workgroups), odd batches and the image borders (zero padding of every conv)."""
    from conftest import engine_with_env
    rng = np.random.default_rng(7)
    tof = (lambda u: (u.astype(np.uint32) << 16).view(np.float32)) if name == "bf16" else (lambda u: u.view(np.float16).astype(np.float32))
    ulp = 2.0 ** -8 if name == "bf16" else 2.0 ** -11
    tol_l, band_l = (0.06, 0.08) if name == "bf16" else (0.008, 0.01)      # logits of these noise images reach +-4
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        with engine_with_env(SHOULDER_UNET_REFERENCE=1) as ref:
            ref.set_params(unet_dtype=DTYPES[name][0])
            for shape in ((5, 256, 256), (2, 256, 512), (3, 512, 512)):
                img = rng.random(shape, dtype=np.float32)
                n_px = shape[0] * shape[1] * shape[2]
                a = engine.unet_infer(img)
                b = ref.unet_infer(img)
                assert float(np.abs(a - b).max()) < tol_l, (shape, float(np.abs(a - b).max()))
                band = np.abs(b) > band_l
                assert np.array_equal((a > 0)[band], (b > 0)[band])
                for lvl, ch in ((0, 32), (1, 64), (2, 128), (3, 256)):
                    n = (n_px >> (2 * lvl)) * ch
                    ta = tof(engine.fetch("unet16.skip%d" % lvl, np.uint16)[:n].copy())
                    tb = tof(ref.fetch("unet16.skip%d" % lvl, np.uint16)[:n].copy())
                    d = np.abs(ta - tb)
                    scale = max(1.0, float(np.abs(tb).max()))
                    assert np.isfinite(ta).all()
                    # ET-rounded first-conv weights and activations that round the other way move the outputs behind them by a few ulps of
                    # their tensor's scale
                    assert float(d.max()) <= 16 * ulp * scale and float(d.mean()) < ulp * scale / 2, (shape, lvl, float(d.max()), float(d.mean()))
                s0a = tof(engine.fetch("unet16.skip0", np.uint16)[:n_px * 32].copy()).reshape(shape + (32,))
                s0b = tof(ref.fetch("unet16.skip0", np.uint16)[:n_px * 32].copy()).reshape(shape + (32,))
                for sl in (np.s_[:, 0], np.s_[:, -1], np.s_[:, :, 0], np.s_[:, :, -1]):      # borders: no stale halo data
                    assert float(np.abs(s0a - s0b)[sl].max()) <= 8 * ulp * max(1.0, float(np.abs(s0b).max()))
    finally:
        engine.set_params(unet_dtype=_lib.UNET_F32)


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_production_network_is_deterministic(engine, name):
    """The persistent kernels hand their work out in tickets, keep tiles in flight across barriers behind counted waits and, at level 0,
    let two groups of waves take turns on the matrix pipe: a tile read before it landed, or overwritten while it was read, would show
    as a result that changes from run to run or with the batch an image is part of.  Same logits bit for bit, run after run, alone
    and inside larger batches, at both image sizes (one item per workgroup and many)."""
    rng = np.random.default_rng(37)
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        for H, W, n in ((256, 512, 1), (256, 512, 3), (512, 512, 5), (512, 512, 16), (512, 512, 40)):
            img = rng.random((n, H, W), dtype=np.float32)
            a = engine.unet_infer(img)
            for _ in range(3):
                assert np.array_equal(a, engine.unet_infer(img))
            assert np.array_equal(a[:1], engine.unet_infer(img[:1]))              # an image's logits do not depend on its batch
            if n > 2:
                assert np.array_equal(a[n - 2:], engine.unet_infer(img[n - 2:]))
    finally:
        engine.set_params(unet_dtype=_lib.UNET_F32)


def test_packed_weights_follow_the_parameter_block(unet_weights):
    """The 16-bit paths pack their conv weights once per parameter block: a second sh_load_unet, and a block written through
    the device pointer + sh_param_block_commit, must both be seen by the next pass (no stale packed copy)."""
    from shoulder_amd import unet_spec
    from shoulder_amd.engine import Engine
    rng = np.random.default_rng(23)
    img = rng.random((2, 256, 512), dtype=np.float32)
    e = Engine(0)
    try:
        e.load_rfc()
        e.load_unet(unet_weights, unet_spec.BASE, unet_spec.DEPTH)
        e.set_params(unet_dtype=_lib.UNET_BF16)
        a = e.unet_infer(img)
        w2 = {k: v.copy() for k, v in unet_weights.items()}
        w2["dec0b_w"] = w2["dec0b_w"] * np.float32(0.5)      # an MFMA layer's weights: only the packed copy carries them
        e.load_unet(w2, unet_spec.BASE, unet_spec.DEPTH)
        b = e.unet_infer(img)
        assert not np.array_equal(a, b)
        f = Engine(0)
        try:
            f.load_rfc()
            f.load_unet(w2, unet_spec.BASE, unet_spec.DEPTH)
            f.set_params(unet_dtype=_lib.UNET_BF16)
            np.testing.assert_array_equal(b, f.unet_infer(img))
            # the first engine's block written over the second one's (what a broadcast does), then committed
            f.load_unet(unet_weights, unet_spec.BASE, unet_spec.DEPTH)
            f.unet_infer(img)
            _, n = e.param_block()
            _, n2 = f.param_block()
            assert n == n2
            f.store("params", e.fetch("params", np.uint8, (n,)))
            f.param_block_commit()
            np.testing.assert_array_equal(b, f.unet_infer(img))
        finally:
            f.close()
    finally:
        e.close()


@pytest.mark.parametrize("base,depth,H,W", [(96, 2, 64, 64), (160, 1, 32, 64), (64, 3, 128, 256), (256, 1, 32, 32)])
def test_other_widths_and_depths(base, depth, H, W):
    """Networks other than the default 4 x base-32 one (a user's ONNX import may have any base % 32 == 0 up to 256): both paths
    against the float64 evaluation.  Regression: the first-conv and head kernels used to stage their weights in LDS arrays sized
    for 64 channels, so base 96 / 160 gave garbage on the f32 path."""
    from oracle import unet as ounet
    from shoulder_amd import unet_spec
    from shoulder_amd.engine import Engine
    rng = np.random.default_rng(base + depth)
    w = unet_spec.make_teacher_weights(seed=9, base=base, depth=depth)
    x = rng.uniform(0, 1, (2, H, W)).astype(np.float32)
    want = np.stack([ounet.forward_f64(w, xi) for xi in x])
    e = Engine(0)
    try:
        e.load_unet(w, base, depth)
        e.set_params(unet_dtype=_lib.UNET_F32)
        assert np.abs(e.unet_infer(x) - want).max() < 5e-5
        e.set_params(unet_dtype=_lib.UNET_BF16)
        assert np.abs(e.unet_infer(x) - want).max() < 0.08
        e.set_params(unet_dtype=_lib.UNET_F16)
        assert np.abs(e.unet_infer(x) - want).max() < 0.012
        with pytest.raises(Exception):
            e.load_unet(unet_spec.make_teacher_weights(seed=1, base=288, depth=1), 288, 1)      # above SH_UNET_MAXBASE
    finally:
        e.close()


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_first_kernel_scales_the_raw_image_itself(engine, oracle_bones, name):
    """In the 16-bit network k_enc0_pp reads the UNSCALED image ("anp.raw", f64) and applies the MinMaxScaler arithmetic of
    k_anp_scale where it loads its patches (no "anp.image" pass): the logits are those of the same network fed the f32 image, bit
    for bit, and a run that also writes the image (sh_set_keep_products) gives the same record."""
    h = oracle_bones("humerus_left_flipped")
    engine.reset_params()
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        engine.upload([(h.verts, h.faces)])
        engine.store("anp.image", np.full((1, 512, 512), -3.0, np.float32))
        lm = engine.run(_lib.STAGE_ALL).copy()
        assert (engine.fetch("anp.image", np.float32, (1, 512, 512)) == -3.0).all()      # never written ...
        lg = engine.fetch("anp.logits", np.float32, (1, 512, 512)).copy()
        engine.set_keep_products(True)
        lm_keep = engine.run(_lib.STAGE_ALL).copy()
        img = engine.fetch("anp.image", np.float32, (1, 512, 512)).copy()                   # ... unless asked for
        assert lm_keep.tobytes() == lm.tobytes() and lm["status"][0] == 0
        assert 0.0 <= img.min() and img.max() <= 1.0 and img.max() > 0.99
        np.testing.assert_array_equal(engine.unet_infer(img), lg)
    finally:
        engine.set_keep_products(False)
        engine.reset_params()


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_up_convolution_inside_the_decoder_conv_is_bit_identical(engine, name):
    """SHOULDER_UP_INSIDE=1 (off by default: faster alone, slower inside the two-lane region -- DESIGN.md section 9): up1 computed by the
    loader waves of dec1a (k_conv3_ldr16<.., UPL = 4>) instead of a launch of its own.  Same arithmetic in the same order as
    k_upconv16g, so the conv sees the same 16-bit values: logits bit for bit, borders and odd batches included."""
    from conftest import engine_with_env
    rng = np.random.default_rng(11)
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        with engine_with_env(SHOULDER_UP_INSIDE=1) as alt:
            alt.set_params(unet_dtype=DTYPES[name][0])
            for shape in ((5, 256, 256), (3, 512, 512)):
                img = rng.random(shape, dtype=np.float32)
                np.testing.assert_array_equal(alt.unet_infer(img), engine.unet_infer(img))
    finally:
        engine.set_params(unet_dtype=_lib.UNET_F32)
