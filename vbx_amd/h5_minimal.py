"""Reader for the HDF5 subset of the reference's ``transform.h5`` files (vbhmm.py:125-129 opens them with h5py):
superblock version 0, root group as a symbol table (version-1 B-tree + local heap), version-1 object headers,
simple dataspaces, little-endian IEEE floating-point datatypes, contiguous layout.  That is what h5py writes by
default for a handful of small float arrays.  Anything else raises ``ValueError`` naming what was met -- install
h5py for general files (``kaldi_formats.read_xvec_transform`` prefers it when present).

Format: "HDF5 File Format Specification Version 2.0", sections III.A (superblock), III.B/III.C (B-tree, symbol
table), III.D (local heap), IV.A (object headers and messages 0x0001, 0x0003, 0x0008, 0x0010).
"""
from __future__ import annotations

import struct

import numpy as np

_SIGNATURE = b'\x89HDF\r\n\x1a\n'
_UNDEF = 0xFFFFFFFFFFFFFFFF


class _File:
    def __init__(self, raw: bytes):
        self.raw = raw
        if raw[:8] != _SIGNATURE:
            raise ValueError('not an HDF5 file')
        version = raw[8]
        if version != 0:
            raise ValueError(f'HDF5 superblock version {version} (only version 0 is handled; use h5py)')
        self.size_offsets, self.size_lengths = raw[13], raw[14]
        if (self.size_offsets, self.size_lengths) != (8, 8):
            raise ValueError('HDF5 file with offsets / lengths that are not 8 bytes wide')
        self.base = self.u64(24)
        # root group symbol table entry starts after the four addresses of the superblock
        root = 24 + 4 * 8
        self.root_header = self.u64(root + 8)
        cache_type = self.u32(root + 16)
        if cache_type != 1:
            raise ValueError('HDF5 root group without cached symbol table addresses')
        self.root_btree, self.root_heap = self.u64(root + 24), self.u64(root + 32)

    def u8(self, o): return self.raw[o]
    def u16(self, o): return struct.unpack_from('<H', self.raw, o)[0]
    def u32(self, o): return struct.unpack_from('<I', self.raw, o)[0]
    def u64(self, o): return struct.unpack_from('<Q', self.raw, o)[0]

    # ---- group traversal -------------------------------------------------------------------------------
    def heap_data(self, addr):
        if self.raw[addr:addr + 4] != b'HEAP':
            raise ValueError('HDF5 local heap signature missing')
        return self.u64(addr + 24)                                # address of the data segment

    def symbol_nodes(self, addr):
        """Leaf symbol-table nodes below the B-tree node at ``addr``."""
        if self.raw[addr:addr + 4] != b'TREE':
            raise ValueError('HDF5 B-tree signature missing')
        node_type, level, used = self.u8(addr + 4), self.u8(addr + 5), self.u16(addr + 6)
        if node_type != 0:
            raise ValueError('HDF5 B-tree of a chunked dataset where a group was expected')
        pos = addr + 8 + 2 * 8                                    # skip the sibling addresses
        for _ in range(used):
            pos += 8                                              # key
            child = self.u64(pos)
            pos += 8
            if level > 0:
                yield from self.symbol_nodes(child)
            else:
                yield child

    def links(self):
        """name -> object header address for the root group."""
        heap = self.heap_data(self.root_heap)
        out = {}
        for node in self.symbol_nodes(self.root_btree):
            if self.raw[node:node + 4] != b'SNOD':
                raise ValueError('HDF5 symbol table node signature missing')
            count = self.u16(node + 6)
            for k in range(count):
                e = node + 8 + k * 40
                name_off, header = self.u64(e), self.u64(e + 8)
                end = self.raw.index(b'\x00', heap + name_off)
                out[self.raw[heap + name_off:end].decode()] = header
        return out

    # ---- datasets ----------------------------------------------------------------------------------------
    def messages(self, addr):
        """(type, body offset, size) of every message of the version-1 object header at ``addr``."""
        if self.u8(addr) != 1:
            raise ValueError(f'HDF5 object header version {self.u8(addr)} (only version 1 is handled; use h5py)')
        total = self.u16(addr + 2)
        blocks = [(addr + 16, self.u32(addr + 8))]                # (start, size) of message blocks
        seen = 0
        while blocks and seen < total:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and seen < total:
                mtype, msize = self.u16(pos), self.u16(pos + 2)
                body = pos + 8
                if mtype == 0x0010:                               # continuation
                    blocks.append((self.u64(body), self.u64(body + 8)))
                else:
                    yield mtype, body, msize
                seen += 1
                pos = body + msize

    def dataset(self, addr):
        shape = dtype = data = None
        for mtype, body, _size in self.messages(addr):
            if mtype == 0x0001:                                   # dataspace
                version, rank = self.u8(body), self.u8(body + 1)
                first = body + (8 if version == 1 else 4)
                shape = tuple(self.u64(first + 8 * k) for k in range(rank))
            elif mtype == 0x0003:                                 # datatype
                cls, bits0 = self.u8(body) & 0x0F, self.u8(body + 1)
                size = self.u32(body + 4)
                if cls != 1 or (bits0 & 1) or size not in (4, 8):
                    raise ValueError('HDF5 dataset that is not little-endian float32 / float64')
                dtype = '<f4' if size == 4 else '<f8'
            elif mtype == 0x0008:                                 # layout
                version = self.u8(body)
                if version != 3 or self.u8(body + 1) != 1:
                    raise ValueError('HDF5 dataset whose layout is not contiguous (version 3); use h5py')
                data = (self.u64(body + 2), self.u64(body + 10))
        if shape is None or dtype is None or data is None or data[0] == _UNDEF:
            raise ValueError('HDF5 dataset without dataspace, datatype or allocated contiguous storage')
        count = int(np.prod(shape)) if shape else 1
        if count * np.dtype(dtype).itemsize != data[1]:
            raise ValueError('HDF5 dataset size does not match its dataspace')
        return np.frombuffer(self.raw, dtype=dtype, count=count, offset=self.base + data[0]).reshape(shape).copy()


def read_datasets(path, names=None):
    """``{name: ndarray}`` for the datasets of the root group (all of them, or ``names``)."""
    with open(path, 'rb') as fd:
        f = _File(fd.read())
    links = f.links()
    wanted = list(links) if names is None else list(names)
    missing = [n for n in wanted if n not in links]
    if missing:
        raise KeyError(f'{path}: no dataset named {missing} (has {sorted(links)})')
    return {n: f.dataset(links[n]) for n in wanted}
