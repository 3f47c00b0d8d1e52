"""Field dumps in the reference's on-disk format: a flat binary `.raw` file plus an XDMF 2
description (reference src/odil/io.py:7-167; `tests/test_io.py`).  Arrays are written as
(Nz, Ny, Nx) in C order; the XDMF lists sizes in that order and origin / spacing reversed
(z first), as ParaView expects for `ORIGIN_DXDYDZ`.  Device tensors are copied to the host once."""

import os
import re

import numpy as np


def _host(u):
    return u.detach().cpu().numpy() if hasattr(u, "detach") else np.asarray(u)


def write_raw_xmf(xmfpath, rawpath, count, spacing=(1, 1, 1), name=None, precision=8, cell=True):
    """XDMF 2 metadata for the binary file `rawpath` holding a scalar of shape count = (Nz, Ny, Nx)."""
    name = name or "data"
    count = [int(c) for c in count]
    sizes = " ".join(str(c) for c in count)
    nodes = " ".join(str(c + 1 if cell else c) for c in count)
    rev = lambda v: " ".join(str(x) for x in reversed(list(v)))
    item = '<DataItem Name="{}" Dimensions="3" NumberType="Float" Precision="8" Format="XML">'
    lines = [
        '<?xml version="1.0" ?>',
        '<!DOCTYPE Xdmf SYSTEM "Xdmf.dtd" []>',
        '<Xdmf Version="2.0">',
        " <Domain>",
        '   <Grid Name="mesh" GridType="Uniform">',
        '     <Topology TopologyType="3DCORECTMesh" Dimensions="{}"/>'.format(nodes),
        '     <Geometry GeometryType="ORIGIN_DXDYDZ">',
        "       " + item.format("Origin"),
        "         " + rev([0, 0, 0]),
        "       </DataItem>",
        "       " + item.format("Spacing"),
        "         " + rev(spacing),
        "       </DataItem>",
        "     </Geometry>",
        '     <Attribute Name="{}" AttributeType="Scalar" Center="{}">'.format(name, "Cell" if cell else "Node"),
        '       <DataItem ItemType="HyperSlab" Dimensions="{}" Type="HyperSlab">'.format(sizes),
        '           <DataItem Dimensions="3 3" Format="XML">',
        "             " + rev([0, 0, 0]),
        "             " + rev([1, 1, 1]),
        "             " + sizes,
        "           </DataItem>",
        '           <DataItem Dimensions="{}" Seek="0" Precision="{}" NumberType="{}" Format="Binary">'.format(
            sizes, precision, "Double" if precision == 8 else "Float"),
        "             " + rawpath,
        "           </DataItem>",
        "       </DataItem>",
        "     </Attribute>",
        "   </Grid>",
        " </Domain>",
        "</Xdmf>",
    ]
    with open(xmfpath, "w") as f:
        f.write("\n".join(lines) + "\n")


def write_raw_with_xmf(u, xmfpath, rawpath=None, spacing=(1, 1, 1), cell=True, name=None):
    """Writes `u` (host array or device tensor; 1-D / 2-D arrays get leading unit axes) as
    `.raw` + XDMF 2.  spacing = (hx, hy, hz).  Returns xmfpath."""
    u = np.ascontiguousarray(_host(u))
    if u.dtype not in (np.float32, np.float64):
        u = u.astype(np.float64)
    while u.ndim < 3:
        u = u[None]
    if u.ndim != 3:
        raise ValueError("Expected at most 3 dimensions, got shape {}".format(u.shape))
    spacing = list(spacing)
    while len(spacing) < 3:
        spacing.append(min(spacing))
    if rawpath is None:
        rawpath = os.path.splitext(xmfpath)[0] + ".raw"
    rel = os.path.relpath(rawpath, start=os.path.dirname(xmfpath) or ".")
    write_raw_xmf(xmfpath, rel, u.shape, spacing, name or "data", 4 if u.dtype == np.float32 else 8, cell)
    u.tofile(rawpath)
    return xmfpath


def write_raw_slab(u_owned, xmfpath, rank, world, axis=0, rawpath=None, spacing=(1, 1, 1), cell=True, name=None,
                   barrier=None):
    """Slab-parallel dump: every rank of a slab decomposition (odil_amd/slab.py, slab_traced.py) writes ITS planes
    of one global field into the one `.raw` file the reference's reader expects (reference src/odil/io.py:145-167
    writes the whole array from one process), at its own offsets; rank 0 also writes the XDMF 2 description of the
    GLOBAL array.  `u_owned`: this rank's owned part (device tensor or host array, <= 3-D, the global array cut
    into `world` equal parts along `axis`).  A device tensor is copied to the host once (its owned planes only,
    through pinned memory).  `barrier`: a callable that synchronises the ranks (`torch.distributed.barrier`).  With
    it rank 0 creates / TRUNCATES the file first (planes of an older run of the same size cannot survive a rank that
    fails or is late), everybody writes after a barrier, and the description appears only after a second barrier
    -- a reader that sees the `.xmf` sees a complete `.raw`.  Without it (ranks emulated one after the other in one
    process) the file is sized by whoever comes first and never truncated."""
    import torch

    if hasattr(u_owned, "detach"):
        t = u_owned.detach()
        if t.is_cuda:
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            host.copy_(t, non_blocking=True)
            torch.cuda.current_stream(t.device).synchronize()
            t = host
        u = t.numpy()
    else:
        u = np.asarray(u_owned)
    if u.dtype not in (np.float32, np.float64):
        u = u.astype(np.float64)
    lead = 3 - u.ndim
    if lead < 0:
        raise ValueError("Expected at most 3 dimensions, got shape {}".format(u.shape))
    u = u.reshape((1,) * lead + u.shape)
    axis3 = axis + lead
    count = list(u.shape)
    n = count[axis3]
    count[axis3] = n * world
    spacing = list(spacing)
    while len(spacing) < 3:
        spacing.append(min(spacing))
    if rawpath is None:
        rawpath = os.path.splitext(xmfpath)[0] + ".raw"
    nbytes = int(np.prod(count)) * u.dtype.itemsize
    if barrier is not None:
        if rank == 0:
            fd = os.open(rawpath, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
            try:
                os.ftruncate(fd, nbytes)
            finally:
                os.close(fd)
            if os.path.exists(xmfpath):
                os.unlink(xmfpath)  # (the description of the previous dump goes before its data does)
        barrier()
    else:
        fd = os.open(rawpath, os.O_RDWR | os.O_CREAT, 0o644)
        try:
            if os.fstat(fd).st_size != nbytes:
                os.ftruncate(fd, nbytes)  # same size from every rank: whoever comes first
        finally:
            os.close(fd)
    out = np.memmap(rawpath, dtype=u.dtype, mode="r+", shape=tuple(count))
    sel = [slice(None)] * 3
    sel[axis3] = slice(rank * n, (rank + 1) * n)
    out[tuple(sel)] = u
    out.flush()
    del out
    if barrier is not None:
        barrier()
    if rank == 0:
        rel = os.path.relpath(rawpath, start=os.path.dirname(xmfpath) or ".")
        write_raw_xmf(xmfpath, rel, count, spacing, name or "data", 4 if u.dtype == np.float32 else 8, cell)
    return xmfpath


def parse_raw_xmf(xmfpath):
    """Metadata of a file written by write_raw_xmf: rawpath, count (Nz, Ny, Nx), spacing (hx, hy, hz),
    name, precision (bytes per value), cell."""
    with open(xmfpath) as f:
        text = " ".join(f.read().split())
    binary = re.search(r'<DataItem Dimensions="(\d+) (\d+) (\d+)"[^>]*Precision="(\d+)"[^>]*Format="Binary">\s*(\S+)', text)
    attr = re.search(r'<Attribute Name="([^"]*)" AttributeType="Scalar" Center="([A-Za-z]*)">', text)
    spacing = re.search(r'<DataItem Name="Spacing"[^>]*>\s*([^<]*?)\s*<', text)
    if not (binary and attr and spacing):
        raise RuntimeError("Not a raw+XDMF2 scalar description: " + xmfpath)
    if attr.group(2) not in ("Cell", "Node"):
        raise RuntimeError("Unknown Center='{}'".format(attr.group(2)))
    return {
        "rawpath": os.path.join(os.path.dirname(xmfpath), binary.group(5)),
        "count": tuple(int(binary.group(i)) for i in (1, 2, 3)),
        "spacing": tuple(float(v) for v in reversed(spacing.group(1).split())),
        "name": attr.group(1),
        "precision": int(binary.group(4)),
        "cell": attr.group(2) == "Cell",
    }


def read_raw_with_xmf(xmfpath):
    meta = parse_raw_xmf(xmfpath)
    dtype = {4: np.float32, 8: np.float64}[meta["precision"]]
    return np.fromfile(meta["rawpath"], dtype).reshape(meta["count"]), meta


read_raw = read_raw_with_xmf
