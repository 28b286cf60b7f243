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
def test_fused_ends_match_layerwise(engine, monkeypatch, name):
    """bf16 path.  Pools fused into conv epilogues are bit-identical to the layer-by-layer kernels (max commutes with the
    bf16 rounding).  The fused first conv runs on the matrix cores with bf16 weights and a hi+lo split image, the fused head
    sums a pixel's 32 products in another order from unrounded activations: both agree with the layer-by-layer network to
    bf16-activation precision."""
    rng = np.random.default_rng(99)
    img = rng.random((2, 256, 256), dtype=np.float32)
    n1 = 2 * 128 * 128 * 64
    bf = (lambda u: (u.astype(np.uint32) << 16).view(np.float32)) if name == "bf16" else (lambda u: u.view(np.float16).astype(np.float32))
    tol_l, band_l, ulp = (0.06, 0.08, 0.02) if name == "bf16" else (0.008, 0.01, 0.0025)
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        fused = engine.unet_infer(img)
        skip_f = engine.fetch("unet16.skip1", np.uint16)[:n1].copy()
        monkeypatch.setenv("SHOULDER_UNET_FUSE_FIRST", "0")
        pools = engine.unet_infer(img)
        skip_q = engine.fetch("unet16.skip1", np.uint16)[:n1].copy()
        monkeypatch.setenv("SHOULDER_UNET_UNFUSED", "1")
        plain = engine.unet_infer(img)
        skip_p = engine.fetch("unet16.skip1", np.uint16)[:n1].copy()
    finally:
        monkeypatch.delenv("SHOULDER_UNET_UNFUSED", raising=False)
        monkeypatch.delenv("SHOULDER_UNET_FUSE_FIRST", raising=False)
        engine.set_params(unet_dtype=_lib.UNET_F32)
    assert np.array_equal(skip_q, skip_p)                  # fused pools: exact
    d1 = np.abs(bf(skip_f) - bf(skip_p))
    assert float(d1.max()) <= ulp * float(np.abs(bf(skip_p)).max()) + 1e-3      # fused first conv: a few ulps of the element type at level 1
    for a in (fused, pools):
        assert float(np.abs(a - plain).max()) < tol_l      # logits of this noise image reach +-3
        band = np.abs(plain) > band_l
        assert np.array_equal((a > 0)[band], (plain > 0)[band])


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_level0_fused_matches_two_barrier_kernel(engine, monkeypatch, name):
    """k_enc0_fused16 (image -> enc0a -> LDS -> enc0b -> skip0 + pool, persistent, weights resident in LDS) against the
    two-barrier kernel k_conv_mfma16<EK, 9, 2, UF_FIRST | UF_POOL>: the same arithmetic (hi + lo split image, ET weights,
    f32 accumulate) with the first conv's 18 products summed in another order, so skip0 agrees to an ulp or two of the
    element type and the logits far inside the type's tolerance -- incl. a ragged last work range (5 images of 256 x 256:
    640 items over 256 workgroups) and the image borders (zero padding of both convs)."""
    rng = np.random.default_rng(7)
    tof = (lambda u: (u.astype(np.uint32) << 16).view(np.float32)) if name == "bf16" else (lambda u: u.view(np.float16).astype(np.float32))
    ulp = 2.0 ** -8 if name == "bf16" else 2.0 ** -11
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        for shape in ((5, 256, 256), (2, 256, 512)):
            img = rng.random(shape, dtype=np.float32)
            n0 = shape[0] * shape[1] * shape[2] * 32
            monkeypatch.setenv("SHOULDER_UNET_L0", "1")
            a = engine.unet_infer(img)
            s0a = tof(engine.fetch("unet16.skip0", np.uint16)[:n0].copy()).reshape(shape + (32,))
            monkeypatch.setenv("SHOULDER_UNET_L0", "0")
            b = engine.unet_infer(img)
            s0b = tof(engine.fetch("unet16.skip0", np.uint16)[:n0].copy()).reshape(shape + (32,))
            d = np.abs(s0a - s0b)
            scale = max(1.0, float(np.abs(s0b).max()))
            # a first-conv activation that rounds the other way (1 ulp) moves a second-conv output by a few ulps of the tensor's scale
            assert float(d.max()) <= 8 * ulp * scale and float(d.mean()) < ulp * scale / 8, (shape, float(d.max()), float(d.mean()))
            for sl in (np.s_[:, 0], np.s_[:, -1], np.s_[:, :, 0], np.s_[:, :, -1]):      # borders: no stale halo data
                assert float(d[sl].max()) <= 8 * ulp * scale
            assert float(np.abs(a - b).max()) < DTYPES[name][1] / 4
    finally:
        monkeypatch.delenv("SHOULDER_UNET_L0", raising=False)
        engine.set_params(unet_dtype=_lib.UNET_F32)


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_resident_weights_match_staged_weights(engine, monkeypatch, name):
    """k_conv3_dma16<..., WRES = 1> keeps the weights of the one-group layers (32->64, 64->64, 64->32, 32->32) in LDS for the
    whole launch instead of staging them with every step: same operations in the same order, bit-identical logits and tensors."""
    rng = np.random.default_rng(13)
    img = rng.random((3, 256, 512), dtype=np.float32)
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        monkeypatch.setenv("SHOULDER_DMA_WRES", "1")
        a = engine.unet_infer(img)
        sa = [engine.fetch(f"unet16.skip{i}", np.uint16).copy() for i in (0, 1)]
        monkeypatch.setenv("SHOULDER_DMA_WRES", "0")
        b = engine.unet_infer(img)
        sb = [engine.fetch(f"unet16.skip{i}", np.uint16).copy() for i in (0, 1)]
        assert np.array_equal(a, b) and all(np.array_equal(x, y) for x, y in zip(sa, sb))
    finally:
        monkeypatch.delenv("SHOULDER_DMA_WRES", raising=False)
        engine.set_params(unet_dtype=_lib.UNET_F32)


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_fused_upconv_dec0a_matches_two_launches(engine, monkeypatch, name):
    """k_dec0a_up16 computes up0's half of dec0a's input tile on the matrix cores inside the conv (k_unet16_dec0.h) instead of
    reading it from HBM: the values are rounded exactly as k_upconv16 rounds them, so the logits are the same bit for bit
    (image borders included: 256 x 512 and 512 x 512, several images)."""
    rng = np.random.default_rng(19)
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        for H, W, n in ((256, 512, 3), (512, 512, 2)):
            img = rng.random((n, H, W), dtype=np.float32)
            monkeypatch.setenv("SHOULDER_UNET_DEC0", "0")
            a = engine.unet_infer(img)
            monkeypatch.setenv("SHOULDER_UNET_DEC0", "1")
            b = engine.unet_infer(img)
            assert np.isfinite(b).all()
            assert np.array_equal(a, b), (H, W, float(np.abs(a - b).max()), int((a != b).sum()))
    finally:
        monkeypatch.delenv("SHOULDER_UNET_DEC0", raising=False)
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


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_work_tickets_match_fixed_shares(engine, monkeypatch, name):
    """k_conv3_dma16 hands its items out in tickets from a global counter (which workgroup computes an item depends on the run);
    the fixed equal shares of SHOULDER_DMA_TICKETS=0 give the same tensors bit for bit, at both image sizes, run after run."""
    rng = np.random.default_rng(17)
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        for H, W, n in ((256, 512, 3), (512, 512, 5)):
            img = rng.random((n, H, W), dtype=np.float32)
            monkeypatch.setenv("SHOULDER_DMA_TICKETS", "0")
            a = engine.unet_infer(img)
            sa = [engine.fetch(f"unet16.skip{i}", np.uint16).copy() for i in (0, 1, 2, 3)]
            monkeypatch.setenv("SHOULDER_DMA_TICKETS", "1")
            for _ in range(2):
                b = engine.unet_infer(img)
                sb = [engine.fetch(f"unet16.skip{i}", np.uint16).copy() for i in (0, 1, 2, 3)]
                assert np.array_equal(a, b) and all(np.array_equal(x, y) for x, y in zip(sa, sb))
    finally:
        monkeypatch.delenv("SHOULDER_DMA_TICKETS", raising=False)
        engine.set_params(unet_dtype=_lib.UNET_F32)


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_loader_wave_conv_bit_identical(engine, monkeypatch, name):
    """k_conv3_ldr16 (k_unet16_ldr.h: 4 compute waves + 4 loader waves per workgroup) against k_conv3_dma16 (every wave stages
    and multiplies): same chunk and tap order per output element -> the same tensors bit for bit, with tickets and with fixed
    shares, at both image sizes, run after run (the hand-off is one barrier per step: a race would show as a changing result)."""
    rng = np.random.default_rng(23)
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        for H, W, n in ((256, 512, 3), (512, 512, 5)):
            img = rng.random((n, H, W), dtype=np.float32)
            monkeypatch.setenv("SHOULDER_DMA_LDR", "0")
            a = engine.unet_infer(img)
            sa = [engine.fetch(f"unet16.skip{i}", np.uint16).copy() for i in (0, 1, 2, 3)]
            monkeypatch.setenv("SHOULDER_DMA_LDR", "1")
            for tickets in ("1", "0", "1"):
                monkeypatch.setenv("SHOULDER_DMA_TICKETS", tickets)
                b = engine.unet_infer(img)
                sb = [engine.fetch(f"unet16.skip{i}", np.uint16).copy() for i in (0, 1, 2, 3)]
                assert np.array_equal(a, b) and all(np.array_equal(x, y) for x, y in zip(sa, sb))
    finally:
        monkeypatch.delenv("SHOULDER_DMA_LDR", raising=False)
        monkeypatch.delenv("SHOULDER_DMA_TICKETS", raising=False)
        engine.set_params(unet_dtype=_lib.UNET_F32)


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_small_workgroup_dec0b_bit_identical(engine, monkeypatch, name):
    """k_dec0b_head_occ (k_unet16_occ.h: 4-wave workgroups on 32 x 8 tiles, two per CU, weight fragments in registers, tickets of four
    tiles) against k_conv3_dma16<.., UF_HEAD, 2, 1, 2> (one 8-wave workgroup per CU on 32 x 16 tiles): the same operations in the same
    order per output -> the same logits bit for bit, at both image sizes, an odd batch, run after run (a race between the LDS-DMA of
    the next tile and the reads of the current one would show as a changing result)."""
    rng = np.random.default_rng(31)
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        for H, W, n in ((256, 512, 3), (512, 512, 5)):
            img = rng.random((n, H, W), dtype=np.float32)
            monkeypatch.setenv("SHOULDER_DEC0B_OCC", "0")
            a = engine.unet_infer(img)
            monkeypatch.setenv("SHOULDER_DEC0B_OCC", "1")
            for _ in range(3):
                assert np.array_equal(a, engine.unet_infer(img))
    finally:
        monkeypatch.delenv("SHOULDER_DEC0B_OCC", raising=False)
        engine.set_params(unet_dtype=_lib.UNET_F32)


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_register_resident_upconv_bit_identical(engine, monkeypatch, name):
    """k_upconv16r (k_unet16_l0.h: a wave keeps its 64 source pixels x Cin in MFMA fragments, the weights stream through LDS one
    (group, phase) slice at a time; up1 and up2) against k_upconv16 (tile staged per chunk): same accumulation order -> logits and
    a decoder tensor bit-identical, at both image sizes and an odd batch."""
    rng = np.random.default_rng(29)
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        for H, W, n in ((256, 512, 3), (512, 512, 5)):
            img = rng.random((n, H, W), dtype=np.float32)
            monkeypatch.setenv("SHOULDER_UPCONV_REG", "0")
            a = engine.unet_infer(img)
            ua = engine.fetch("unet16.b", np.uint16).copy()
            monkeypatch.setenv("SHOULDER_UPCONV_REG", "1")
            for _ in range(2):
                b = engine.unet_infer(img)
                ub = engine.fetch("unet16.b", np.uint16).copy()
                assert np.array_equal(a, b) and np.array_equal(ua, ub)
    finally:
        monkeypatch.delenv("SHOULDER_UPCONV_REG", raising=False)
        engine.set_params(unet_dtype=_lib.UNET_F32)


@pytest.mark.parametrize("name", ["bf16", "f16"])
def test_row_upconv_matches_per_phase_kernel(engine, monkeypatch, name):
    """k_upconv16 (16x16 source tile x 32 channels x both column phases of a row parity per workgroup: full output lines per
    wave) sums every output in the order of the per-phase two-barrier kernel: logits and a decoder tensor are bit-identical."""
    rng = np.random.default_rng(11)
    img = rng.random((3, 256, 512), dtype=np.float32)
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        monkeypatch.setenv("SHOULDER_UNET_UPCONV", "1")
        a = engine.unet_infer(img)
        ua = engine.fetch("unet16.b", np.uint16).copy()
        monkeypatch.setenv("SHOULDER_UNET_UPCONV", "0")
        b = engine.unet_infer(img)
        ub = engine.fetch("unet16.b", np.uint16).copy()
        assert np.array_equal(a, b) and np.array_equal(ua, ub)
    finally:
        monkeypatch.delenv("SHOULDER_UNET_UPCONV", raising=False)
        engine.set_params(unet_dtype=_lib.UNET_F32)


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
    """In the 16-bit network k_enc0_fused16 reads the UNSCALED image ("anp.raw", f64) and applies the MinMaxScaler arithmetic of
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
def test_three_buffer_dec0b_bit_identical(engine, monkeypatch, name):
    """k_dec0b_head3 (k_unet16_dec0b3.h: three halo buffers, the tile two items ahead in flight, counted waits over a fixed number of
    vector-memory operations per step) against k_conv3_dma16<.., UF_HEAD, 2, 1, 2> (two buffers): the same operations in the same
    order -> the same logits bit for bit, at both image sizes, batches of one item per workgroup and of many, run after run (a tile
    read before it landed, or overwritten while it was read, would show as a changing result)."""
    rng = np.random.default_rng(37)
    engine.set_params(unet_dtype=DTYPES[name][0])
    try:
        for H, W, n in ((256, 512, 1), (256, 512, 3), (512, 512, 5), (512, 512, 16)):
            img = rng.random((n, H, W), dtype=np.float32)
            monkeypatch.setenv("SHOULDER_DEC0B3", "0")
            a = engine.unet_infer(img)
            monkeypatch.setenv("SHOULDER_DEC0B3", "1")
            for _ in range(3):
                assert np.array_equal(a, engine.unet_infer(img))
    finally:
        monkeypatch.delenv("SHOULDER_DEC0B3", raising=False)
        engine.set_params(unet_dtype=_lib.UNET_F32)
