"""CPU tests of the Radiance .hdr decode: known-answer RGBE values, the oracle against a numpy restatement on
run-length coded and flat files, header grammar / rejection cases (stb_image's), and the product's host-side header parser
(vqhip_hdr_parse_header needs no GPU) against the oracle's."""
import numpy as np
import pytest

from tests import oracle_lib as O
from vqengine_amd import capi, synth


def _np_decode(rgbe):
    e = rgbe[..., 3].astype(np.int32)
    f = np.ldexp(np.float32(1.0), e - 136).astype(np.float32)
    out = np.empty(rgbe.shape[:2] + (4,), np.float32)
    out[..., :3] = rgbe[..., :3].astype(np.float32) * f[..., None]
    out[e == 0, :3] = 0.0
    out[..., 3] = 1.0
    return out


def test_rgbe_known_answers():
    """stbi__hdr_convert: rgb * 2^(e-136); e == 0 -> black; alpha 1."""
    px = np.array([[[128, 64, 32, 129], [255, 255, 255, 136], [1, 0, 0, 1], [200, 100, 50, 0], [128, 0, 0, 255], [1, 2, 3, 9],
                    [255, 0, 128, 128], [77, 77, 77, 120]]], np.uint8)
    img = O.hdr_decode(synth.hdr_file_bytes(px))
    exp = [(1.0, 0.5, 0.25), (255.0, 255.0, 255.0), (2.0 ** -135, 0, 0), (0, 0, 0), (128 * 2.0 ** 119, 0, 0),
           (2.0 ** -127, 2.0 ** -126, 3 * 2.0 ** -127), (255 / 256, 0, 0.5), (77 * 2.0 ** -16,) * 3]
    for i, e in enumerate(exp):
        assert tuple(img[0, i, :3]) == tuple(np.float32(v) for v in e), (i, img[0, i], e)
    assert np.all(img[..., 3] == 1.0)


@pytest.mark.parametrize("shape,rle", [((37, 200), True), ((37, 200), False), ((5, 7), True), ((3, 8), True), ((2, 300), True)])
def test_oracle_decode_matches_numpy(shape, rle):
    h, w = shape
    r = np.random.default_rng(w * 131 + h)
    rgb = synth.equirect(w, h)[..., :3] if w >= 16 else r.random((h, w, 3)) * 100
    rgbe = synth.float_to_rgbe(rgb)
    rgbe[0, :min(w, 40)] = (9, 9, 9, 130)                      # long runs
    rgbe[h - 1, ::3, 3] = 0                                    # zero exponents
    rgbe[h // 2] = r.integers(0, 256, (w, 4), dtype=np.uint8)  # incompressible row, every exponent incl. denormal scales
    data = synth.hdr_file_bytes(rgbe, rle=rle)
    if rle and w >= 8:
        assert len(data) < w * h * 4 + 200 + 4 * h + 2 * h * (w // 64 + 4)
    img = O.hdr_decode(data)
    n, idx = O.bits_equal(img, _np_decode(rgbe))
    assert n == 0, (n, idx)
    # Ward encoding error bound: one part in 128 of the largest channel
    if w >= 16:
        dec = img[1:h // 2, :, :3].astype(np.float64)
        src = rgb[1:h // 2]
        assert np.all(np.abs(dec - src) <= src.max(-1, keepdims=True) / 128.0 + 1e-30)


def test_first_scanline_without_marker_is_flat_data():
    """stb_image: a first scanline not starting 02 02 means the whole image is stored flat, even for width >= 8."""
    r = np.random.default_rng(3)
    rgbe = r.integers(3, 256, (4, 16, 4), dtype=np.uint8)      # first byte != 2
    data = synth.hdr_file_bytes(rgbe, rle=False)
    n, _ = O.bits_equal(O.hdr_decode(data), _np_decode(rgbe))
    assert n == 0


def _mut(data, old, new):
    assert old in data
    return data.replace(old, new, 1)


def test_header_grammar_and_rejections():
    rgbe = synth.float_to_rgbe(synth.equirect(32, 4)[..., :3])
    good = synth.hdr_file_bytes(rgbe)
    assert capi.hdr_parse_header(good)[:2] == (32, 4)
    assert capi.hdr_parse_header(synth.hdr_file_bytes(rgbe, magic=b"#?RGBE"))[:2] == (32, 4)
    O.hdr_decode(synth.hdr_file_bytes(rgbe, magic=b"#?RGBE"))
    bad = {
        "magic": _mut(good, b"#?RADIANCE", b"#?RADIANCF"),
        "format": _mut(good, b"FORMAT=32-bit_rle_rgbe", b"FORMAT=32-bit_rle_xyze"),
        "layout": _mut(good, b"-Y 4 +X 32", b"+Y 4 +X 32"),
        "layout2": _mut(good, b"-Y 4 +X 32", b"-Y 4 -X 32"),
        "no_blank": _mut(good, b"FORMAT=32-bit_rle_rgbe\n\n", b"FORMAT=32-bit_rle_rgbe\n"),
    }
    for name, data in bad.items():
        with pytest.raises(capi.VQHipError):
            capi.hdr_parse_header(data)
        with pytest.raises(ValueError):
            O.hdr_decode(data)
    # data errors: truncated file, wrong scanline length, zero-length run, run overrunning the scanline
    w, h, off = capi.hdr_parse_header(good)
    for data in (good[:-5], good[:off] + bytes((2, 2, 0, 33)) + good[off + 4:], good[:off + 4] + bytes((128,)) + good[off + 5:],
                 good[:off + 4] + bytes((128 + 33, 7)) + good[off + 6:]):
        with pytest.raises(ValueError):
            O.hdr_decode(data)


def test_product_header_parser_matches_oracle():
    lib = O.load()
    import ctypes as C
    for (h, w) in ((1, 1), (512, 1024), (4096, 8192), (7, 33000)):
        head = b"#?RADIANCE\n# x\nGAMMA=1\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=2\n\n" + b"-Y %d +X %d\n" % (h, w) + b"\x02\x02"
        pw, ph, poff = capi.hdr_parse_header(head)
        ow, oh, ooff = C.c_int(), C.c_int(), C.c_size_t()
        assert lib.vqo_hdr_parse_header(head, len(head), C.byref(ow), C.byref(oh), C.byref(ooff)) == 0
        assert (pw, ph, poff) == (ow.value, oh.value, ooff.value) == (w, h, len(head) - 2)


def test_downsize_oracle_is_the_block_mean_and_rejects_other_ratios():
    """vqo_hdr_downsize_rgba32f (the fallback of EnvironmentMap.cpp:142-209 restated as a k x k mean — PARITY UNPINNED, the reference's
    resampler is stb_image_resize in the absent submodule): against a float64 block mean; non-integer / anisotropic ratios are refused."""
    img = synth.equirect(256, 128)
    for k in (1, 2, 4, 8):
        got = O.hdr_downsize(img, 256 // k, 128 // k)
        want = img.astype(np.float64).reshape(128 // k, k, 256 // k, k, 4).mean((1, 3))
        assert np.allclose(got[..., :3], want[..., :3], rtol=2e-6, atol=0) and (got[..., 3] == 1).all()
    for ow, oh in ((100, 50), (128, 32), (512, 256)):
        with pytest.raises(ValueError):
            O.hdr_downsize(img, ow, oh)
