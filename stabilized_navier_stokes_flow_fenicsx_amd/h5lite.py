"""Minimal HDF5 writer: groups + contiguous little-endian datasets, nothing else.

The reference stores the heavy data of its XDMF output in HDF5 (``XDMFFile.write_mesh`` / ``write_function``,
NavierStokes/NavierStokesChannelFlow.py:333-341) and its own post-processing opens that file with h5py and reads
``h5f["Function"][name]["0"]`` (NavierStokes/streamtrace.py:87-96).  Neither h5py nor PyTables exists offline, so
this module writes the container itself, straight from the HDF5 file-format specification (version 1.x objects,
which every libhdf5 since 1.0 reads):

  superblock v0  ->  root group (object header v1 + symbol-table message)
  group          =   object header v1 { symbol table message } + B-tree v1 node (one leaf level) + local heap
                     + symbol table node(s) ("SNOD")
  dataset        =   object header v1 { dataspace v1, datatype (IEEE f64 / two's-complement ints, LE),
                                        fill value v2 (never written), layout v3 contiguous }

Limits (by design): no chunking, compression, attributes, links other than hard links, or datasets above 2^63
bytes; at most ``2 * LEAF_K`` entries per symbol-table node and ``2 * INTERNAL_K`` nodes per group
(= 256 children per group).  ``tests/h5read_min.py`` walks the same structures independently and the test-suite
checks the files bit for bit; where libhdf5 happens to be installed it is used as a second reader.
"""
from __future__ import annotations

import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K = 4            # symbol-table node holds up to 2*LEAF_K entries
INTERNAL_K = 16       # B-tree node holds up to 2*INTERNAL_K children
_SIG = b"\x89HDF\r\n\x1a\n"


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


class _Group:
    def __init__(self):
        self.children = {}          # name -> _Group | np.ndarray


class H5Writer:
    """``w = H5Writer(); w.dataset("/Mesh/mesh/geometry", array); w.write(path)``"""

    def __init__(self):
        self.root = _Group()

    def dataset(self, path: str, array) -> None:
        a = np.asarray(array)
        if a.dtype.kind not in "fiu" or a.dtype.itemsize not in (1, 2, 4, 8):
            raise TypeError(f"unsupported dtype {a.dtype}")
        a = np.ascontiguousarray(a.astype(a.dtype.newbyteorder("<"), copy=False))
        parts = [p for p in path.split("/") if p]
        g = self.root
        for p in parts[:-1]:
            nxt = g.children.setdefault(p, _Group())
            if not isinstance(nxt, _Group):
                raise ValueError(f"{p} is a dataset")
            g = nxt
        if parts[-1] in g.children:
            raise ValueError(f"{path} exists")
        g.children[parts[-1]] = a

    # ---- serialisation ------------------------------------------------------------------------------------
    def write(self, filename: str) -> None:
        self.buf = bytearray(96)                       # superblock placeholder
        root_hdr, root_btree, root_heap = self._emit_group(self.root)
        eof = len(self.buf)
        sb = bytearray()
        sb += _SIG
        sb += bytes([0, 0, 0, 0, 0, 8, 8, 0])          # versions, size of offsets / lengths
        sb += struct.pack("<HHI", LEAF_K, INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)        # base, free-space info, end of file, driver info
        sb += struct.pack("<QQII", 0, root_hdr, 1, 0)  # root symbol-table entry: name offset, header, cache type 1
        sb += struct.pack("<QQ", root_btree, root_heap)
        assert len(sb) == 96
        self.buf[0:96] = sb
        with open(filename, "wb") as fh:
            fh.write(self.buf)

    def _alloc(self, data: bytes) -> int:
        self.buf += b"\0" * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    @staticmethod
    def _message(mtype: int, body: bytes, flags: int = 0) -> bytes:
        body = _pad8(body)
        return struct.pack("<HHB3x", mtype, len(body), flags) + body

    def _object_header(self, messages) -> int:
        body = b"".join(messages)
        hdr = struct.pack("<BxHII4x", 1, len(messages), 1, len(body))       # version 1, #messages, refcount, size
        return self._alloc(hdr + body)

    def _emit_dataset(self, a: np.ndarray) -> int:
        data_addr = self._alloc(a.tobytes()) if a.size else UNDEF
        rank = a.ndim
        space = struct.pack("<BBB5x", 1, rank, 0) + b"".join(struct.pack("<Q", d) for d in a.shape)
        size = a.dtype.itemsize
        if a.dtype.kind == "f":
            # class 1 (floating point) version 1; bit field: little endian, mantissa normalisation 2 (implied msb),
            # sign location in byte 1
            if size == 8:
                dt = struct.pack("<BBBBI", 0x11, 0x20, 63, 0, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            elif size == 4:
                dt = struct.pack("<BBBBI", 0x11, 0x20, 31, 0, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
            else:
                raise TypeError("float16 is not supported")
        else:
            signed = 0x08 if a.dtype.kind == "i" else 0x00
            dt = struct.pack("<BBBBI", 0x10, signed, 0, 0, size) + struct.pack("<HH", 0, 8 * size)
        fill = struct.pack("<BBBB", 2, 2, 0, 0)        # version 2, allocate at create (2), write time 0, undefined
        layout = struct.pack("<BBQQ", 3, 1, data_addr, a.nbytes)          # version 3, class 1 = contiguous
        return self._object_header([
            self._message(0x0001, space),
            self._message(0x0003, dt, flags=1),         # constant
            self._message(0x0005, fill),
            self._message(0x0008, layout),
        ])

    def _emit_group(self, g: _Group):
        names = sorted(g.children)                     # symbol-table entries are ordered by name
        if len(names) > 4 * LEAF_K * INTERNAL_K:
            raise ValueError("too many entries in one group for this writer")
        child_addr = {}
        child_cache = {}
        for nm in names:
            c = g.children[nm]
            if isinstance(c, _Group):
                hdr, bt, hp = self._emit_group(c)
                child_addr[nm] = hdr
                child_cache[nm] = (bt, hp)
            else:
                child_addr[nm] = self._emit_dataset(c)
        # local heap: offset 0 holds the empty string, then the names (each null-terminated, 8-byte aligned)
        heap_data = bytearray(8)
        off = {}
        for nm in names:
            off[nm] = len(heap_data)
            heap_data += _pad8(nm.encode() + b"\0")
        # a free block at the tail keeps libhdf5 happy when it wants to grow the heap (block: next = 1, size)
        free_off = len(heap_data)
        heap_data += struct.pack("<QQ", 1, 16)
        data_addr = self._alloc(bytes(heap_data))
        heap = b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, data_addr)
        heap_addr = self._alloc(heap)
        # symbol table nodes (<= 2*LEAF_K entries each) under ONE leaf-level B-tree node
        cap = 2 * LEAF_K
        chunks = [names[i:i + cap] for i in range(0, len(names), cap)]
        snods = []
        for ch in chunks:
            b = bytearray(b"SNOD" + struct.pack("<BxH", 1, len(ch)))
            for nm in ch:
                if nm in child_cache:
                    bt, hp = child_cache[nm]
                    b += struct.pack("<QQII", off[nm], child_addr[nm], 1, 0) + struct.pack("<QQ", bt, hp)
                else:
                    b += struct.pack("<QQII", off[nm], child_addr[nm], 0, 0) + bytes(16)
            b += bytes(40 * (cap - len(ch)))
            snods.append(self._alloc(bytes(b)))
        # B-tree v1 node, type 0 (group), level 0: keys are heap offsets of the LAST name in each child
        bt = bytearray(b"TREE" + struct.pack("<BBH", 0, 0, len(snods)) + struct.pack("<QQ", UNDEF, UNDEF))
        bt += struct.pack("<Q", 0)                                  # key 0: the empty string
        for ch, ad in zip(chunks, snods):
            bt += struct.pack("<QQ", ad, off[ch[-1]])
        bt += bytes((2 * INTERNAL_K - len(snods)) * 16)             # unused child/key slots
        btree_addr = self._alloc(bytes(bt))
        hdr = self._object_header([self._message(0x0011, struct.pack("<QQ", btree_addr, heap_addr))])
        return hdr, btree_addr, heap_addr


# ---- read-back of the same subset (the product's own consumer: streamtrace.read_mesh_and_function) ------------
def read_datasets(filename: str) -> dict:
    """{"/path/to/dataset": ndarray} of every contiguous dataset reachable through old-style groups."""
    with open(filename, "rb") as fh:
        b = fh.read()
    if b[:8] != _SIG or b[8] != 0 or b[13] != 8 or b[14] != 8:
        raise ValueError(f"{filename}: not an HDF5 file this reader understands (superblock v0, 8-byte offsets)")
    out = {}

    def messages(addr):
        ver, nmsg, _, size = struct.unpack_from("<BxHII", b, addr)
        if ver != 1:
            raise ValueError("object header version")
        p, res = addr + 16, []
        while p < addr + 16 + size and len(res) < nmsg:
            t, n = struct.unpack_from("<HH", b, p)
            res.append((t, b[p + 8:p + 8 + n]))
            p += 8 + n
        return res

    def visit(addr, prefix):
        msgs = messages(addr)
        st = [m for t, m in msgs if t == 0x0011]
        if not st:
            d = dict(msgs)
            sp, dt, lay = d[0x0001], d[0x0003], d[0x0008]
            shape = struct.unpack_from("<" + "Q" * sp[1], sp, 8)
            size = struct.unpack_from("<I", dt, 4)[0]
            kind = "f" if (dt[0] & 15) == 1 else ("i" if dt[1] & 8 else "u")
            if lay[0] != 3 or lay[1] != 1:
                raise ValueError(f"{prefix}: only contiguous datasets")
            addr_d, _ = struct.unpack_from("<QQ", lay, 2)
            n = int(np.prod(shape)) if shape else 1
            out[prefix] = (np.zeros(shape, f"<{kind}{size}") if addr_d == UNDEF else
                           np.frombuffer(b, dtype=f"<{kind}{size}", count=n, offset=addr_d).reshape(shape).copy())
            return
        btree, heap = struct.unpack_from("<QQ", st[0], 0)
        data = struct.unpack_from("<Q", b, heap + 24)[0]

        def walk(node):
            level, used = struct.unpack_from("<BH", b, node + 5)
            for k in range(used):
                child = struct.unpack_from("<Q", b, node + 32 + 16 * k)[0]
                if level:
                    walk(child)
                    continue
                for s in range(struct.unpack_from("<H", b, child + 6)[0]):
                    noff, ohdr = struct.unpack_from("<QQ", b, child + 8 + 40 * s)
                    e = b.index(b"\0", data + noff)
                    visit(ohdr, prefix + "/" + b[data + noff:e].decode())

        walk(btree)

    visit(struct.unpack_from("<Q", b, 64)[0], "")
    return out
