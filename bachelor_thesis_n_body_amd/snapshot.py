"""CONCEPT-layout snapshot I/O without h5py.

The reference's scripts read particle state from CONCEPT's HDF5 snapshots through h5py:
    f['components/matter/pos'][:]  (N,3) f64      SnapVsPK/Power_compare.py:143-144, FraP_kTilDelta.py:55-60
    f['components/matter/mom'][:]  (N,3) f64      Different_Sizes/Plotter.py:223-224
    f.attrs['boxsize']                             TheFeedingPipeline/FraP_kTilDelta.py:63-68
h5py is not available in this image, so this module carries a minimal reader for the subset of HDF5 those files
use (superblock v0, v1 object headers, old-style groups = v1 B-tree + local heap + symbol nodes, contiguous or
compact dataset layout, fixed-size little-endian numeric types, v1-v3 attribute messages) and a writer that emits
the same subset, so that engine state round-trips through files the thesis scripts can open.

`read_positions` / `read_boxsize` keep the names and meaning of TheFeedingPipeline/FraP_kTilDelta.py:55-68.
"""
from __future__ import annotations

import struct
from typing import Optional, Any, Dict, List, Tuple

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5Error(ValueError):
    pass


def _pad8(n: int) -> int:
    return (n + 7) & ~7


class _File:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.b = fh.read()
        if self.b[:8] != _SIG:
            raise HDF5Error(f"{path}: not an HDF5 file")
        ver = self.b[8]
        if ver not in (0, 1):
            raise HDF5Error(f"{path}: superblock version {ver} not supported (only the v0/v1 layout CONCEPT/h5py writes)")
        if self.b[13] != 8 or self.b[14] != 8:
            raise HDF5Error("only 8-byte offsets/lengths are supported")
        o = 24 + (4 if ver == 1 else 0)
        self.base = self.u64(o)
        o += 32
        # root group symbol table entry
        self.root_header = self.u64(o + 8)

    # primitives
    def u8(self, o): return self.b[o]
    def u16(self, o): return struct.unpack_from("<H", self.b, o)[0]
    def u32(self, o): return struct.unpack_from("<I", self.b, o)[0]
    def u64(self, o): return struct.unpack_from("<Q", self.b, o)[0]

    # -- object headers --------------------------------------------------------------------------
    def messages(self, addr: int) -> List[Tuple[int, int, int]]:
        """[(type, data offset, size)] of a version-1 object header, continuation blocks followed."""
        if self.b[addr:addr + 4] == b"OHDR":
            raise HDF5Error("version-2 object headers are not supported")
        if self.u8(addr) != 1:
            raise HDF5Error(f"object header version {self.u8(addr)} not supported")
        nmsg, hsize = self.u16(addr + 2), self.u32(addr + 8)
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            o, size = blocks.pop(0)
            end = o + size
            while o + 8 <= end and len(out) < nmsg:
                t, s = self.u16(o), self.u16(o + 2)
                d = o + 8
                out.append((t, d, s))
                if t == 0x10:
                    blocks.append((self.u64(d), self.u64(d + 8)))
                o = d + s
        return out

    # -- groups ------------------------------------------------------------------------------------
    def _heap_name(self, heap_addr: int, off: int) -> str:
        if self.b[heap_addr:heap_addr + 4] != b"HEAP":
            raise HDF5Error("bad local heap")
        data = self.u64(heap_addr + 24)
        e = self.b.index(b"\0", data + off)
        return self.b[data + off:e].decode()

    def _walk_btree(self, node: int, heap: int, out: Dict[str, int]) -> None:
        if self.b[node:node + 4] == b"SNOD":
            n = self.u16(node + 6)
            for i in range(n):
                e = node + 8 + 40 * i
                out[self._heap_name(heap, self.u64(e))] = self.u64(e + 8)
            return
        if self.b[node:node + 4] != b"TREE":
            raise HDF5Error("bad group B-tree node")
        used = self.u16(node + 6)
        o = node + 24
        for i in range(used):
            child = self.u64(o + 8 + 16 * i)
            self._walk_btree(child, heap, out)

    def links(self, header: int) -> Dict[str, int]:
        for t, d, _s in self.messages(header):
            if t == 0x11:
                out: Dict[str, int] = {}
                self._walk_btree(self.u64(d), self.u64(d + 8), out)
                return out
        raise HDF5Error("object is not an old-style group")

    def resolve(self, path: str) -> int:
        h = self.root_header
        for part in [p for p in path.split("/") if p]:
            ln = self.links(h)
            if part not in ln:
                raise KeyError(f"{path!r}: no member {part!r} (have {sorted(ln)})")
            h = ln[part]
        return h

    # -- datatypes / dataspaces --------------------------------------------------------------------
    def _dtype(self, o: int):
        cls, ver = self.u8(o) & 0x0F, self.u8(o) >> 4
        bits0 = self.u8(o + 1)
        size = self.u32(o + 4)
        if ver not in (1, 2, 3):
            raise HDF5Error(f"datatype version {ver}")
        if cls == 0:
            return np.dtype(("<" if not bits0 & 1 else ">") + ("i" if bits0 & 8 else "u") + str(size)), size
        if cls == 1:
            return np.dtype(("<" if not bits0 & 1 else ">") + "f" + str(size)), size
        if cls == 3:
            return np.dtype(f"S{size}"), size
        raise HDF5Error(f"datatype class {cls} not supported")

    def _dataspace(self, o: int) -> Tuple[int, ...]:
        ver, rank = self.u8(o), self.u8(o + 1)
        if ver == 1:
            d = o + 8
        elif ver == 2:
            d = o + 4
        else:
            raise HDF5Error(f"dataspace version {ver}")
        return tuple(self.u64(d + 8 * i) for i in range(rank))

    def attrs(self, header: int) -> Dict[str, Any]:
        out: Dict[str, Any] = {}
        for t, d, _s in self.messages(header):
            if t != 0x0C:
                continue
            ver = self.u8(d)
            nsz, tsz, ssz = self.u16(d + 2), self.u16(d + 4), self.u16(d + 6)
            o = d + 8 + (1 if ver == 3 else 0)
            pad = _pad8 if ver == 1 else (lambda n: n)
            name = self.b[o:o + nsz].split(b"\0")[0].decode()
            o += pad(nsz)
            try:
                dt, _ = self._dtype(o)
            except HDF5Error:
                continue            # e.g. variable-length strings: not needed by the thesis scripts' numeric reads
            shape = self._dataspace(o + pad(tsz))
            o += pad(tsz) + pad(ssz)
            count = int(np.prod(shape)) if shape else 1
            val = np.frombuffer(self.b, dt, count, o).reshape(shape)
            if dt.kind == "S":
                val = np.array([v.split(b"\0")[0].decode() for v in val.ravel()]).reshape(shape)
            out[name] = val[()] if not shape else val.copy()
        return out

    def dataset(self, header: int) -> np.ndarray:
        dt = shape = None
        layout = None
        for t, d, _s in self.messages(header):
            if t == 0x03:
                dt, _ = self._dtype(d)
            elif t == 0x01:
                shape = self._dataspace(d)
            elif t == 0x08:
                layout = d
        if dt is None or shape is None or layout is None:
            raise HDF5Error("object is not a dataset")
        ver = self.u8(layout)
        if ver != 3:
            raise HDF5Error(f"data layout version {ver} not supported")
        cls = self.u8(layout + 1)
        count = int(np.prod(shape)) if shape else 1
        if cls == 1:      # contiguous
            addr = self.u64(layout + 2)
            if addr == _UNDEF:
                return np.zeros(shape, dt)
            return np.frombuffer(self.b, dt, count, self.base + addr).reshape(shape).copy()
        if cls == 0:      # compact
            return np.frombuffer(self.b, dt, count, layout + 4).reshape(shape).copy()
        raise HDF5Error("chunked datasets are not supported (CONCEPT snapshots are contiguous)")


def read_dataset(path: str, name: str) -> np.ndarray:
    f = _File(path)
    return f.dataset(f.resolve(name))


def read_attrs(path: str, name: str = "/") -> Dict[str, Any]:
    f = _File(path)
    return f.attrs(f.resolve(name))


def read_positions(filename: str) -> np.ndarray:
    """(N,3) f64 positions — TheFeedingPipeline/FraP_kTilDelta.py:55-60."""
    return read_dataset(filename, "components/matter/pos")


def read_momenta(filename: str) -> np.ndarray:
    """(N,3) f64 momenta — Different_Sizes/Plotter.py:223-224."""
    return read_dataset(filename, "components/matter/mom")


def read_boxsize(filename: str) -> float:
    """Root attribute `boxsize` — TheFeedingPipeline/FraP_kTilDelta.py:63-68."""
    return float(read_attrs(filename)["boxsize"])


# ------------------------------------------------------------------------------------------------ writer
class _Writer:
    """Emits superblock v0 + old-style groups + contiguous f64 datasets + scalar f64/fixed-string attributes."""

    def __init__(self):
        self.buf = bytearray()

    def alloc(self, n: int, align: int = 8) -> int:
        while len(self.buf) % align:
            self.buf.append(0)
        o = len(self.buf)
        self.buf.extend(b"\0" * n)
        return o

    def put(self, o: int, data: bytes) -> None:
        self.buf[o:o + len(data)] = data

    @staticmethod
    def _dtype_msg(dt: np.dtype) -> bytes:
        if dt.kind == "f" and dt.itemsize == 8:
            # class 1 (float) v1; LE, IEEE: bit offset 0, precision 64, exp loc 52 size 11, mant loc 0 size 52, bias 1023
            return struct.pack("<BBBBI", 0x11, 0x20, 0x3F, 0x00, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
        if dt.kind == "S":
            return struct.pack("<BBBBI", 0x13, 0x00, 0x00, 0x00, dt.itemsize)   # null-terminated ASCII
        raise HDF5Error(f"writer: unsupported dtype {dt}")

    @staticmethod
    def _space_msg(shape) -> bytes:
        b = struct.pack("<BBB5x", 1, len(shape), 0)
        for s in shape:
            b += struct.pack("<Q", s)
        return b

    def _attr_msg(self, name: str, value) -> bytes:
        if isinstance(value, str):
            raw = value.encode() + b"\0"
            dt, shape, data = np.dtype(f"S{len(raw)}"), (), raw
        else:
            arr = np.asarray(value, "<f8")
            dt, shape, data = arr.dtype, arr.shape, arr.tobytes()
        nm = name.encode() + b"\0"
        t, s = self._dtype_msg(dt), self._space_msg(shape)
        body = struct.pack("<BBHHH", 1, 0, len(nm), len(t), len(s))
        body += nm.ljust(_pad8(len(nm)), b"\0") + t.ljust(_pad8(len(t)), b"\0") + s.ljust(_pad8(len(s)), b"\0") + data
        return body

    def object_header(self, msgs: List[Tuple[int, bytes]]) -> int:
        enc = b""
        for t, body in msgs:
            body = body.ljust(_pad8(len(body)), b"\0")
            enc += struct.pack("<HHB3x", t, len(body), 0) + body
        o = self.alloc(16 + len(enc))
        self.put(o, struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(enc)) + enc)
        return o

    def dataset(self, arr: np.ndarray, attrs=None, data_addr: Optional[int] = None) -> int:
        """Object header of a contiguous dataset. With `data_addr` the raw data is NOT stored here: the caller places it at that
        file offset (write_snapshot keeps all metadata in front and the arrays behind, as in the reference's files)."""
        arr = np.ascontiguousarray(arr, "<f8")
        if data_addr is None:
            data = self.alloc(arr.nbytes)
            self.put(data, arr.tobytes())
        else:
            data = int(data_addr)
        msgs = [(0x01, self._space_msg(arr.shape)), (0x03, self._dtype_msg(arr.dtype)),
                (0x08, struct.pack("<BBQQ", 3, 1, data, arr.nbytes))]
        for k, v in (attrs or {}).items():
            msgs.append((0x0C, self._attr_msg(k, v)))
        return self.object_header(msgs)

    def group(self, members: Dict[str, int], attrs=None) -> Tuple[int, int, int]:
        names = sorted(members)
        if len(names) > 8:
            raise HDF5Error("writer: at most 8 members per group")
        heap_data = bytearray(b"\0" * 8)       # offset 0 = empty name, as libhdf5 lays it out
        offs = {}
        for n in names:
            offs[n] = len(heap_data)
            heap_data += n.encode() + b"\0"
            while len(heap_data) % 8:
                heap_data.append(0)
        seg = max(_pad8(len(heap_data)) + 16, 88)
        heap_data = heap_data.ljust(seg, b"\0")
        free = _pad8(len(heap_data.rstrip(b"\0")) + 1)
        if seg - free >= 16:
            heap_data[free:free + 16] = struct.pack("<QQ", 1, seg - free)   # free block: next = 1 (none), size
        else:
            free = _UNDEF
        dseg = self.alloc(seg)
        self.put(dseg, bytes(heap_data))
        heap = self.alloc(32)
        self.put(heap, b"HEAP" + struct.pack("<B3xQQQ", 0, seg, free, dseg))
        snod = self.alloc(8 + 40 * 8)            # 2*leaf K (K = 4) entries
        ent = b""
        for n in names:
            ent += struct.pack("<QQII16x", offs[n], members[n], 0, 0)
        self.put(snod, b"SNOD" + struct.pack("<BBH", 1, 0, len(names)) + ent)
        tree = self.alloc(24 + 8 * (2 * 16 + 1) + 8 * 2 * 16)   # internal K = 16
        last = offs[names[-1]] if names else 0
        self.put(tree, b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, _UNDEF, _UNDEF) + struct.pack("<QQQ", 0, snod, last))
        msgs = [(0x11, struct.pack("<QQ", tree, heap))]
        for k, v in (attrs or {}).items():
            msgs.append((0x0C, self._attr_msg(k, v)))
        return self.object_header(msgs), tree, heap


def write_snapshot(filename: str, pos: np.ndarray, mom: np.ndarray, boxsize: float, mass: float = 1.0,
                   unit_length: str = "Mpc", unit_time: str = "Gyr", unit_mass: str = "10**(10)*m_sun",
                   extra_attrs=None) -> None:
    """Write `components/matter/{pos,mom}` (N,3) f64 + root attrs `boxsize`, `unit length|time|mass` in the layout
    the thesis scripts read."""
    pos, mom = np.asarray(pos, np.float64), np.asarray(mom, np.float64)
    if pos.ndim != 2 or pos.shape[1] != 3 or mom.shape != pos.shape:
        raise ValueError("pos and mom must both have shape (N, 3)")
    pos_le, mom_le = np.ascontiguousarray(pos, "<f8"), np.ascontiguousarray(mom, "<f8")
    root_attrs = {"boxsize": float(boxsize), "unit length": unit_length, "unit time": unit_time, "unit mass": unit_mass}
    root_attrs.update(extra_attrs or {})
    # File layout of the reference's snapshots (a 32^3 file is 1 583 104 B = 10 240 + 2 x 786 432): all metadata in the
    # first META bytes, `pos` at byte META, `mom` directly behind it, nothing after. META = 10 240 unless the attributes need more.
    meta = 10240
    while True:
        w = _Writer()
        w.alloc(96)                                   # superblock v0 + root symbol-table entry
        h_pos = w.dataset(pos_le, data_addr=meta)
        h_mom = w.dataset(mom_le, data_addr=meta + pos_le.nbytes)
        g_matter, _, _ = w.group({"pos": h_pos, "mom": h_mom}, {"mass": float(mass), "N": float(pos.shape[0])})
        g_comp, _, _ = w.group({"matter": g_matter})
        g_root, tree, heap = w.group({"components": g_comp}, root_attrs)
        if len(w.buf) <= meta:
            break
        meta = (len(w.buf) + 2047) // 2048 * 2048
    eof = meta + pos_le.nbytes + mom_le.nbytes
    sb = _SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
    sb += struct.pack("<QQQQ", 0, _UNDEF, eof, _UNDEF)
    sb += struct.pack("<QQII", 0, g_root, 1, 0) + struct.pack("<QQ", tree, heap)
    w.put(0, sb)
    with open(filename, "wb") as fh:
        fh.write(bytes(w.buf).ljust(meta, b"\0"))
        fh.write(pos_le.tobytes())
        fh.write(mom_le.tobytes())
