"""On-disk artefacts either side of the hot path: `train.csv` through History
and raw + XDMF2 field dumps, in the reference's formats (reference src/odil/history.py,
src/odil/io.py, tests/test_io.py).  When the reference checkout is present (build container only)
the files are also read back with the reference's own reader."""

import os
import sys

import numpy as np
import pytest

import odil_amd as odil


def test_raw_xmf_round_trip(tmp_path):
    """reference tests/test_io.py:12-31."""
    nx, ny, nz = 3, 4, 5
    spacing = (4 / nx, 5 / ny, 6 / nz)
    for dtype in [np.float32, np.float64]:
        xmf = str(tmp_path / "data.xdmf2")
        src = np.linspace(0, 1, nx * ny * nz).reshape((nz, ny, nx)).astype(dtype)
        odil.write_raw_with_xmf(src, xmf, spacing=spacing, name="foo")
        u, meta = odil.read_raw_with_xmf(xmf)
        assert meta["count"] == src.shape and meta["name"] == "foo" and meta["cell"]
        assert meta["precision"] == np.dtype(dtype).itemsize
        np.testing.assert_array_almost_equal(meta["spacing"], spacing, decimal=8)
        assert u.dtype == dtype and np.array_equal(u, src)
    # 2-D node-centred field: leading unit axis, third spacing filled with the smallest
    xmf = str(tmp_path / "sub" / "n.xmf")
    os.makedirs(os.path.dirname(xmf))
    odil.write_raw_with_xmf(np.arange(12.0).reshape(3, 4), xmf, spacing=(0.5, 0.25), cell=False, name="u")
    u, meta = odil.read_raw(xmf)
    assert u.shape == (1, 3, 4) and not meta["cell"] and meta["spacing"] == (0.5, 0.25, 0.25)
    assert os.path.exists(str(tmp_path / "sub" / "n.raw"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/odil"), reason="reference checkout not present")
def test_files_are_readable_by_the_reference_reader(tmp_path):
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_io", "/root/reference/src/odil/io.py")
    ref_io = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_io)
    src = np.random.default_rng(0).standard_normal((5, 4, 3)).astype(np.float32)
    xmf = str(tmp_path / "a.xmf")
    odil.write_raw_with_xmf(src, xmf, spacing=(0.1, 0.2, 0.3), name="vel", cell=False)
    u, meta = ref_io.read_raw_with_xmf(xmf)
    assert np.array_equal(u, src) and meta["name"] == "vel" and not meta["cell"]
    np.testing.assert_allclose(meta["spacing"], (0.1, 0.2, 0.3))
    # and the other way round
    xmf2 = str(tmp_path / "b.xmf")
    ref_io.write_raw_with_xmf(src.astype(np.float64), xmf2, spacing=(1, 2, 3), name="p")
    u2, meta2 = odil.read_raw_with_xmf(xmf2)
    assert np.array_equal(u2, src.astype(np.float64)) and meta2["spacing"] == (1.0, 2.0, 3.0) and meta2["cell"]


def test_history_csv_format(tmp_path):
    path = str(tmp_path / "train.csv")
    h = odil.History(csvpath=path, warmup=1)
    h.append("epoch", 0)
    h.append("loss", np.float64(2.5))
    h.write()
    assert open(path).read() == ""  # warm-up row is held back: late columns may still appear
    h.append("epoch", 10)
    h.append("loss", np.array(1.25))
    h.append("tmax", 3.0)  # new column: back-filled with a zero of its type
    h.write()
    h.append("epoch", 20)
    h.append("loss", None)  # None repeats a zero of the column's type
    h.append("tmax", np.float32(0.5))
    h.write()
    assert open(path).read() == "epoch,loss,tmax\n0,2.5,0.0\n10,1.25,3.0\n20,0.0,0.5\n"
    h.append("epoch", 30)
    with pytest.raises(RuntimeError, match="Missing values for columns: loss,tmax,"):
        h.write()
    h2 = odil.History()
    h.data["epoch"].pop()
    h.save(str(tmp_path / "h.pickle"))
    h2.load(str(tmp_path / "h.pickle"))
    assert h2.count == 3 and h2.get("tmax") == [0.0, 3.0, 0.5]
    h.append("extra", 1.0)
    h.append("epoch", 40), h.append("loss", 0.1), h.append("tmax", 0.1)
    with pytest.raises(RuntimeError, match="Unexpected keys in history"):
        h.write()


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/odil"), reason="reference checkout not present")
def test_history_matches_reference_bytes(tmp_path):
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_history", "/root/reference/src/odil/history.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    paths = [str(tmp_path / "a.csv"), str(tmp_path / "b.csv")]
    for cls, path in zip([ref.History, odil.History], paths):
        h = cls(csvpath=path, warmup=1)
        for epoch in range(4):
            h.append("epoch", epoch * 5)
            h.append("frame", epoch // 2)
            h.append("norm_fu", np.array(0.3 / (epoch + 1)))
            h.append("loss", np.float64(0.09) / (epoch + 1) ** 2)
            h.append("walltime", float(np.round(0.0123 * epoch, 3)))
            if epoch:
                h.append("error_u", np.float32(1.5) / epoch)
            h.write()
        h.close()
    assert open(paths[0], "rb").read() == open(paths[1], "rb").read()


@pytest.mark.parametrize("axis,world", [(0, 2), (1, 3), (2, 2)])
def test_slab_parallel_dump_equals_whole_array_dump(tmp_path, axis, world):
    """write_raw_slab: every rank writes its planes of the global field at its own offsets of ONE .raw file; the
    result is byte-identical to the single-process dump of the whole array (reference src/odil/io.py:145-167),
    for a cut along any axis, ranks in any order."""
    from odil_amd import io

    rng = np.random.default_rng(3)
    u = rng.standard_normal((4, 6, 8)).astype(np.float32)
    io.write_raw_with_xmf(u, str(tmp_path / "whole.xmf"), spacing=(0.1, 0.2, 0.3), name="u")
    n = u.shape[axis] // world
    for rank in reversed(range(world)):
        part = np.take(u, np.arange(rank * n, (rank + 1) * n), axis=axis)
        io.write_raw_slab(part, str(tmp_path / "slab.xmf"), rank, world, axis=axis, spacing=(0.1, 0.2, 0.3), name="u")
    assert open(tmp_path / "whole.raw", "rb").read() == open(tmp_path / "slab.raw", "rb").read()
    assert open(tmp_path / "whole.xmf").read().replace("whole.raw", "slab.raw") == open(tmp_path / "slab.xmf").read()
    got, meta = io.read_raw_with_xmf(str(tmp_path / "slab.xmf"))
    assert np.array_equal(got, u) and meta["name"] == "u"


def test_slab_parallel_dump_with_a_barrier_replaces_a_stale_file(tmp_path):
    """With `barrier` rank 0 truncates the shared file before anybody writes and the description appears after
    everybody has: a stale dump of the same size cannot leak planes, whatever order the ranks arrive in."""
    import threading

    from odil_amd import io

    world, axis = 3, 1
    rng = np.random.default_rng(4)
    stale = rng.standard_normal((4, 6, 8)).astype(np.float32)
    u = rng.standard_normal((4, 6, 8)).astype(np.float32)
    path = str(tmp_path / "slab.xmf")
    io.write_raw_with_xmf(stale, path, spacing=(0.1, 0.2, 0.3), name="u")
    n = u.shape[axis] // world
    bar = threading.Barrier(world)
    seen = []

    def rank_main(rank):
        part = np.take(u, np.arange(rank * n, (rank + 1) * n), axis=axis)
        if rank == 0:  # (the truncating rank arrives last)
            import time

            time.sleep(0.2)
        io.write_raw_slab(part, path, rank, world, axis=axis, spacing=(0.1, 0.2, 0.3), name="u", barrier=bar.wait)
        seen.append(rank)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert sorted(seen) == list(range(world))
    got, _ = io.read_raw_with_xmf(path)
    assert np.array_equal(got, u)
