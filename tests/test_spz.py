"""CPU tests of the SPZ codec, mirroring the reference's tests/e2e/spz.rs:
write -> read equality, versions 1..=3, SH degrees 0..=3, fractional bits 8/12/16, SH quantize
bits, header errors — with the product checked byte-for-byte against the oracle restatement and
both against the numpy golden vectors (tests/golden/make_golden_spz.py) and examples/model.spz."""
import ast
import gzip
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SEEDS = list(range(15)) + [42, 123]
# tests/e2e/spz.rs:14-20 ASSERT_GAUSSIAN_OPTIONS
POS_EPS, ROT_EPS, COLOR_TOL, SH_EPS, SCALE_EPS = 1.0, 1e-1, 2, 1e-1, 1.0


@pytest.fixture(scope="module")
def gspz():
    return np.load(os.path.join(HERE, "golden", "golden_spz_v1.npz"))


@pytest.fixture(scope="module")
def fixture_gaussians(golden, ob):
    """the Gaussians the golden encode cases were made from (golden_v1.npz: seeds + the unit-test Gaussian)"""
    g = np.zeros(len(golden["pos"]), dtype=ob.GAUSSIAN_DTYPE)
    for k in ("rot", "pos", "color", "scale"):
        g[k] = golden[k]
    g["sh"] = golden["sh"].reshape(len(g), 45)
    return g


def _cases(gspz):
    return [ast.literal_eval(str(c)) for c in gspz["encode_cases"]]


def _opts(gs, case):
    kw = dict(case)
    if "sh_bits" in kw:
        kw["sh_quantize_bits"] = kw.pop("sh_bits")
    return kw


def _ulps(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b).max() if a.size else 0


def _assert_gaussian(a, b, sh_coeffs=15):
    """tests/common/assert.rs:65-118 with the SPZ tolerances"""
    assert np.abs(a["rot"] - b["rot"]).max() <= ROT_EPS
    assert np.abs(a["pos"] - b["pos"]).max() <= POS_EPS
    assert np.abs(a["color"].astype(int) - b["color"].astype(int)).max() <= COLOR_TOL
    exp_sh = b["sh"].copy()
    exp_sh[:, 3 * sh_coeffs:] = 0
    assert np.abs(a["sh"] - exp_sh).max() <= SH_EPS
    assert np.abs(a["scale"] - b["scale"]).max() <= SCALE_EPS


def _check_decoded(got, gspz, prefix):
    assert np.array_equal(got["pos"], gspz[prefix + "pos"])
    assert np.array_equal(got["rot"], gspz[prefix + "rot"])
    assert np.array_equal(got["color"], gspz[prefix + "color"])
    assert np.array_equal(got["sh"], gspz[prefix + "sh"].reshape(got["sh"].shape))
    assert _ulps(got["scale"], gspz[prefix + "scale"]) <= 2   # numpy's expf vs libm's


# ---- oracle pinned to the golden vectors -----------------------------------------------------------

def test_oracle_decodes_model_spz_as_golden(ob, gspz):
    raw = gzip.decompress(open(os.path.join(HERE, "golden", "model.spz"), "rb").read())
    got = ob.spz_decode_raw(raw)
    assert len(got) == 9
    _check_decoded(got, gspz, "model_")


def test_oracle_encode_matches_golden_bytes(ob, gspz, fixture_gaussians):
    g = fixture_gaussians
    for i, case in enumerate(_cases(gspz)):
        kw = _opts(None, case)
        payload = ob.spz_encode_raw(g, **kw)
        assert payload == gspz[f"enc{i}_bytes"].tobytes(), case
        _check_decoded(ob.spz_decode_raw(payload), gspz, f"enc{i}_")


def test_oracle_header_errors(ob):
    hdr = np.array([0x5053474E, 2, 0], dtype="<u4").tobytes() + bytes([3, 12, 0, 0])
    assert len(ob.spz_decode_raw(hdr)) == 0
    for bad, code in ((b"\0\0\0\0" + hdr[4:], -1), (hdr[:4] + np.uint32(999).tobytes() + hdr[8:], -2),
                      (hdr[:12] + bytes([4, 12, 0, 0]), -3), (hdr[:8], -4)):
        with pytest.raises(ValueError) as e:
            ob.spz_decode_raw(bad)
        assert e.value.args[0] == code


# ---- product == oracle == golden ------------------------------------------------------------------

def test_product_decodes_model_spz(gs, ob, gspz):
    """examples/model.spz (reference data file): 9 points, version 2, SH degree 3, 12 fractional bits"""
    path = os.path.join(HERE, "golden", "model.spz")
    got = gs.SpzGaussians.read_from_file(path)
    h = got.header
    assert (h.magic, h.version, h.num_points, h.sh_degree, h.fractional_bits, h.flags) == (0x5053474E, 2, 9, 3, 12, 0)
    assert list(gspz["model_header"]) == [2, 9, 3, 12, 0]
    _check_decoded(got.gaussians, gspz, "model_")
    raw = gzip.decompress(open(path, "rb").read())
    assert got.gaussians.tobytes() == ob.spz_decode_raw(raw).tobytes()
    assert gs.SpzGaussians.read_decompressed(raw).gaussians.tobytes() == got.gaussians.tobytes()
    # the same scene as examples/model.ply, up to SPZ quantisation
    ply = gs.PlyGaussians.read_from_file(os.path.join(HERE, "golden", "model.ply")).iter_gaussian()
    assert np.abs(got.gaussians["pos"] - ply["pos"]).max() <= 2.0 ** -12
    assert np.abs(got.gaussians["rot"] - ply["rot"]).max() <= 1.0 / 127.5
    assert np.abs(got.gaussians["scale"] / ply["scale"] - 1).max() <= 0.04
    assert np.abs(got.gaussians["color"].astype(int) - ply["color"].astype(int)).max() <= COLOR_TOL
    assert np.abs(got.gaussians["sh"] - ply["sh"]).max() <= 1.0 / 128


def test_product_encode_matches_oracle_and_golden(gs, ob, gspz, fixture_gaussians):
    g = fixture_gaussians
    for i, case in enumerate(_cases(gspz)):
        kw = _opts(gs, case)
        payload = gs.SpzGaussians.write_gaussians_decompressed(g, gs.spz_options(**kw))
        assert payload == ob.spz_encode_raw(g, **kw), case
        assert payload == gspz[f"enc{i}_bytes"].tobytes(), case
        got = gs.SpzGaussians.read_decompressed(payload)
        assert got.gaussians.tobytes() == ob.spz_decode_raw(payload).tobytes()
        _check_decoded(got.gaussians, gspz, f"enc{i}_")


def test_gzip_framing_interoperates(gs, ob):
    """write_to / read_from use a gzip member (flate2 GzEncoder/GzDecoder, spz.rs:945-959)"""
    g = ob.given_gaussians([42, 123])
    z = gs.SpzGaussians.write_gaussians(g)
    raw = gs.SpzGaussians.write_gaussians_decompressed(g)
    assert z[:2] == b"\x1f\x8b" and gzip.decompress(z) == raw
    a = gs.SpzGaussians.read_from(z)
    b = gs.SpzGaussians.read_from(gzip.compress(raw, 1))
    assert a.gaussians.tobytes() == b.gaussians.tobytes() == ob.spz_decode_raw(raw).tobytes()


# ---- mirrors of tests/e2e/spz.rs ------------------------------------------------------------------

def test_len_and_is_empty(gs, ob):
    """spz.rs:59-64"""
    got = gs.SpzGaussians.read_from(gs.SpzGaussians.write_gaussians(ob.given_gaussians([42, 123])))
    assert len(got) == 2 and got.header.num_points == 2
    empty = gs.SpzGaussians.read_from(gs.SpzGaussians.write_gaussians(ob.given_gaussians([])))
    assert len(empty) == 0


def test_write_to_file_and_read_from_file_should_be_equal(gs, ob, tmp_path):
    """spz.rs:67-93: a second write of what was read reproduces the file"""
    g = ob.given_gaussians([42, 123])
    p = tmp_path / "a.spz"
    p.write_bytes(gs.SpzGaussians.write_gaussians(g))
    back = gs.SpzGaussians.read_from_file(str(p))
    assert len(back) == 2
    again = gs.SpzGaussians.read_from(gs.SpzGaussians.write_gaussians(back.gaussians))
    assert np.abs(again.gaussians["pos"] - back.gaussians["pos"]).max() == 0
    _assert_gaussian(again.gaussians, back.gaussians)


@pytest.mark.parametrize("version", [1, 2, 3])
def test_roundtrip_versions(gs, ob, version):
    """spz.rs:110-121,185-199"""
    g = ob.given_gaussians([42, 123])
    got = gs.SpzGaussians.read_from(gs.SpzGaussians.write_gaussians(g, gs.spz_options(version=version)))
    assert got.header.version == version and len(got) == len(g)
    _assert_gaussian(got.gaussians, g)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_roundtrip_sh_degrees(gs, ob, deg):
    """spz.rs:123-134,202-233: coefficients past the header degree read back as zero"""
    g = ob.given_gaussians([42, 123])
    got = gs.SpzGaussians.read_from(gs.SpzGaussians.write_gaussians(g, gs.spz_options(sh_degree=deg)))
    assert got.header.sh_degree == deg
    ncoef = {0: 0, 1: 3, 2: 8, 3: 15}[deg]
    _assert_gaussian(got.gaussians, g, sh_coeffs=ncoef)
    assert not got.gaussians["sh"][:, 3 * ncoef:].any()


@pytest.mark.parametrize("bits", [8, 12, 16])
def test_roundtrip_fractional_bits(gs, ob, bits):
    """spz.rs:136-147,235-249"""
    g = ob.given_gaussians([42, 123])
    got = gs.SpzGaussians.read_from(gs.SpzGaussians.write_gaussians(g, gs.spz_options(fractional_bits=bits)))
    assert got.header.fractional_bits == bits
    _assert_gaussian(got.gaussians, g)
    assert np.abs(got.gaussians["pos"] - g["pos"]).max() <= 2.0 ** -(bits + 1) + 1e-7


@pytest.mark.parametrize("qbits", [(0, 0, 0), (4, 4, 4), (5, 5, 5), (8, 8, 8), (0, 1, 2), (2, 4, 6), (4, 5, 5)])
def test_roundtrip_sh_quantize_bits(gs, ob, qbits):
    """spz.rs:252-274"""
    g = ob.given_gaussians([42, 123])
    got = gs.SpzGaussians.read_from(gs.SpzGaussians.write_gaussians(g, gs.spz_options(sh_quantize_bits=qbits)))
    _assert_gaussian(got.gaussians, g)


def test_invalid_version_message(gs, ob):
    """spz.rs:277-293"""
    with pytest.raises(gs.SpzError) as e:
        gs.SpzGaussians.write_gaussians(ob.given_gaussians([42]), gs.spz_options(version=999))
    assert str(e.value) == "Unsupported SPZ version: 999, expected one of 1..=3"
    hdr = np.array([0x5053474E, 999, 0], dtype="<u4").tobytes() + bytes([3, 12, 0, 0])
    with pytest.raises(gs.SpzError) as e:
        gs.SpzGaussians.read_decompressed(hdr)
    assert str(e.value) == "Unsupported SPZ version: 999, expected one of 1..=3"


def test_invalid_magic_message(gs):
    """spz.rs:472-490"""
    hdr = np.array([0x12345678, 2, 0], dtype="<u4").tobytes() + bytes([3, 12, 0, 0])
    with pytest.raises(gs.SpzError) as e:
        gs.SpzGaussians.read_from(gzip.compress(hdr))
    assert str(e.value) == "Invalid SPZ magic number: 12345678, expected 5053474E"


def test_truncated_and_garbage_inputs(gs, ob):
    raw = gs.SpzGaussians.write_gaussians_decompressed(ob.given_gaussians([42, 123]))
    with pytest.raises(gs.SpzError):
        gs.SpzGaussians.read_decompressed(raw[:-1])
    with pytest.raises(gs.SpzError):
        gs.SpzGaussians.read_decompressed(raw[:10])
    with pytest.raises(gs.SpzError):
        gs.SpzGaussians.read_from(b"not a gzip stream")
    with pytest.raises(gs.GsError):
        gs.SpzGaussians.write_gaussians(ob.given_gaussians([42]), gs.spz_options(sh_quantize_bits=(9, 9, 9)))


def test_antialiased_flag(gs, ob):
    raw = gs.SpzGaussians.write_gaussians_decompressed(ob.given_gaussians([1]), gs.spz_options(antialiased=True))
    assert raw[14] == 1 and gs.SpzGaussians.read_decompressed(raw).header.flags == 1


def test_extremes_saturate_like_rust_casts(gs, ob):
    """`as u8` / `as i32` saturate; scale 0 -> ln = -inf -> 0; huge positions wrap into 24 bits like the reference"""
    g = ob.given_gaussians([7, 8, 9])
    g["scale"][0] = [0.0, 1e30, 1e-30]
    g["sh"][1, :6] = [5.0, -5.0, 0.999, -1.0, 1.0, 0.0]
    g["pos"][2] = [3000.0, -3000.0, 1e12]
    g["rot"][2] = [0.0, 0.0, 0.0, -2.0]
    for v in (1, 2, 3):
        kw = dict(version=v)
        a = gs.SpzGaussians.write_gaussians_decompressed(g, gs.spz_options(**kw))
        assert a == ob.spz_encode_raw(g, **kw)
        assert gs.SpzGaussians.read_decompressed(a).gaussians.tobytes() == ob.spz_decode_raw(a).tobytes()


def test_spz_to_device_buffer(gs, ob):
    """the decoded Gaussians feed GaussianPod.pack directly (the loader -> pack path of SURVEY §3.1)"""
    got = gs.SpzGaussians.read_from_file(os.path.join(HERE, "golden", "model.spz"))
    pod = gs.GaussianPod(0, 0)   # GaussianPodWithShSingleCov3dRotScaleConfigs
    packed = pod.from_gaussian(got.gaussians)
    assert packed.tobytes() == ob.pack(0, 0, got.gaussians).tobytes()


def test_gzip_bomb_and_truncation_are_rejected(gs):
    """A gzip member that inflates far past the size its own SPZ header declares must be refused
    before it eats memory, and a truncated member must not be accepted (the inflate
    buffer used to double without bound and lengths were cast to 32 bits)."""
    import gzip
    import struct
    hdr = struct.pack("<IIIBBBB", 0x5053474E, 2, 0, 0, 12, 0, 0)        # version 2, zero points
    bomb = gzip.compress(hdr + b"\0" * (64 << 20), compresslevel=9)     # 64 MiB of zeros behind an empty scene
    assert len(bomb) < 1 << 20
    with pytest.raises(gs.SpzError):
        gs.SpzGaussians.read_from(bomb)
    ok = gzip.compress(hdr)
    assert len(gs.SpzGaussians.read_from(ok)) == 0
    with pytest.raises(gs.SpzError):
        gs.SpzGaussians.read_from(ok[:-6])
    with pytest.raises(gs.SpzError):
        gs.SpzGaussians.read_from(b"not a gzip stream at all")
