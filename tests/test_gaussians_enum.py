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
