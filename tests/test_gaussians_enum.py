"""The unified `Gaussians` representation, mirroring tests/e2e/gaussian.rs."""
import numpy as np
import pytest

SEEDS = [42, 123]


def _given(ob):
    return ob.given_gaussians(SEEDS)


def test_collect_internal_iter_is_identity(gs, ob):
    """gaussian.rs:7-33"""
    g = _given(ob)
    a = gs.Gaussians.from_gaussians_iter(g, gs.GaussiansSource.Internal)
    b = gs.Gaussians(g)
    assert a.source() == b.source() == gs.GaussiansSource.Internal
    assert a.iter_gaussian().tobytes() == b.iter_gaussian().tobytes() == g.tobytes()


def test_collect_ply_and_spz_iter_within_format_tolerance(gs, ob):
    """gaussian.rs:35-105"""
    g = _given(ob)
    ply = gs.Gaussians.from_gaussians_iter(g, gs.GaussiansSource.Ply).iter_gaussian()
    assert np.abs(ply["pos"] - g["pos"]).max() < 1e-4 and np.abs(ply["rot"] - g["rot"]).max() < 1e-4
    assert np.abs(ply["color"].astype(int) - g["color"].astype(int)).max() <= 1
    spz = gs.Gaussians.from_gaussians_iter(g, gs.GaussiansSource.Spz).iter_gaussian()
    assert np.abs(spz["pos"] - g["pos"]).max() <= 1.0 and np.abs(spz["rot"] - g["rot"]).max() <= 0.1
    assert np.abs(spz["sh"] - g["sh"]).max() <= 0.1 and np.abs(spz["scale"] - g["scale"]).max() <= 1.0
    assert np.abs(spz["color"].astype(int) - g["color"].astype(int)).max() <= 2


@pytest.mark.parametrize("source", ["Internal", "Ply", "Spz"])
def test_source_len_is_empty(gs, ob, source):
    """gaussian.rs:107-138"""
    g = _given(ob)
    x = gs.Gaussians.from_gaussians_iter(g, source)
    assert x.source() == source and len(x) == len(g) and not x.is_empty()
    assert gs.Gaussians.from_gaussians_iter(g[:0], source).is_empty()


@pytest.mark.parametrize("source,ext", [("Ply", ".ply"), ("Spz", ".spz")])
def test_write_read_file_and_buffer_equal(gs, ob, tmp_path, source, ext):
    """gaussian.rs:148-170,199-221: what is read back equals what was written (Spz: the same columns)"""
    x = gs.Gaussians.from_gaussians_iter(_given(ob), source)
    path = str(tmp_path / ("g" + ext))
    x.write_to_file(path)
    y = gs.Gaussians.read_from_file(path, source)
    assert len(y) == len(x) and y == x
    z = gs.Gaussians.read_from(x.write_to(), source)
    assert z == x and z.iter_gaussian().tobytes() == x.iter_gaussian().tobytes()


def test_internal_cannot_be_read_or_written(gs, ob, tmp_path):
    """gaussian.rs:172-197,223-247: the reference's messages"""
    x = gs.Gaussians(_given(ob))
    with pytest.raises(ValueError, match="cannot write Internal Gaussians to file"):
        x.write_to_file(str(tmp_path / "x.bin"))
    with pytest.raises(ValueError, match="cannot write Internal Gaussians to buffer"):
        x.write_to()
    with pytest.raises(ValueError, match="cannot read Internal Gaussians from file"):
        gs.Gaussians.read_from_file(str(tmp_path / "x.bin"), gs.GaussiansSource.Internal)
    with pytest.raises(ValueError, match="cannot read Internal Gaussians from buffer"):
        gs.Gaussians.read_from(b"", gs.GaussiansSource.Internal)


def test_spz_read_write_preserves_reference_file_columns(gs):
    """reading examples/model.spz and writing it back keeps the payload byte for byte"""
    import gzip
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model.spz")
    s = gs.SpzGaussians.read_from_file(path)
    raw = gzip.decompress(open(path, "rb").read())
    assert s.write_decompressed() == raw
    assert gzip.decompress(s.write_to()) == raw
    assert gs.SpzGaussians.read_from(s.write_to()) == s


def test_c_abi_gaussians_read_write(gs):
    """gs_gaussians_read / gs_gaussians_write (GaussiansSource behind the C ABI, src/gaussian.rs:394-548):
    same bytes / Gaussians as the per-format entry points, Internal refused with the reference's messages."""
    import ctypes as C
    import os
    L = gs._capi.load()
    gold = os.path.join(os.path.dirname(__file__), "golden")
    for name, src, ref in (("model.ply", 1, lambda b: gs.gaussian_from_ply(gs.PlyGaussians.read_from(b).pods)),
                           ("model.spz", 2, lambda b: gs.SpzGaussians.read_from(b).iter_gaussian())):
        raw = open(os.path.join(gold, name), "rb").read()
        buf = np.frombuffer(raw, dtype=np.uint8)
        n = C.c_size_t()
        assert L.gs_gaussians_read(buf.ctypes.data, buf.size, src, None, 0, C.byref(n)) == 0
        out = np.zeros(n.value, dtype=gs.GAUSSIAN_DTYPE)
        assert L.gs_gaussians_read(buf.ctypes.data, buf.size, src, out.ctypes.data, n.value, C.byref(n)) == 0
        exp = np.ascontiguousarray(ref(raw), dtype=gs.GAUSSIAN_DTYPE)
        assert out.tobytes() == exp.tobytes()
        # write -> read round trip through the same pair of entry points
        size = C.c_size_t()
        assert L.gs_gaussians_write(out.ctypes.data, len(out), src, None, 0, C.byref(size)) == 0
        enc = np.zeros(size.value, dtype=np.uint8)
        assert L.gs_gaussians_write(out.ctypes.data, len(out), src, enc.ctypes.data, enc.size, C.byref(size)) == 0
        if src == 1:
            assert enc.tobytes() == gs.PlyGaussians.from_gaussians(out).write_to()
        else:
            assert enc.tobytes() == gs.SpzGaussians.write_gaussians(out)
        assert L.gs_gaussians_read(enc.ctypes.data, enc.size, src, None, 0, C.byref(n)) == 0 and n.value == len(out)
    info = gs._capi.ErrorInfo()
    n = C.c_size_t()
    assert L.gs_gaussians_read(buf.ctypes.data, buf.size, 0, None, 0, C.byref(n)) == -1
    L.gs_last_error(C.byref(info))
    assert info.message == b"cannot read Internal Gaussians from buffer"
    assert L.gs_gaussians_write(out.ctypes.data, len(out), 0, None, 0, C.byref(n)) == -1
    L.gs_last_error(C.byref(info))
    assert info.message == b"cannot write Internal Gaussians to buffer"
