"""Test-only writer of LMDB data files (there is no `lmdb` package in this image): lays records out the way LMDB 0.9 does on disk
(data version 1) so that `vk_lmdb_*` can be exercised on trees of any depth.  Not a general LMDB implementation: one bulk load of
sorted records, no free list, no sub-databases.

Layout written (little-endian, 64-bit):
  page header (16 B)   pgno u64 | pad u16 | flags u16 | lower u16, upper u16  (overflow pages: page count u32 instead of lower / upper)
  meta page            header | magic 0xBEEFC0DE u32 | version 1 u32 | address u64 | mapsize u64 | MDB_db free | MDB_db main | last_pg u64 | txnid u64
  MDB_db (48 B)        pad u32 (free db: the page size) | flags u16 | depth u16 | branch, leaf, overflow pages u64 x3 | entries u64 | root u64
  node                 lo u16 | hi u16 | flags u16 | ksize u16 | key | data      leaf: lo | hi << 16 = data size, F_BIGDATA 0x01 -> data = overflow pgno u64
                                                                                 branch: lo | hi << 16 | flags << 32 = child pgno, first key empty
  node offsets (u16) grow up from byte 16, nodes grow down from the end of the page; node sizes are rounded up to even."""
import struct

P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 1, 2, 4, 8
F_BIGDATA = 1
INVALID = 0xFFFFFFFFFFFFFFFF


def _even(n):
    return (n + 1) & ~1


def _page(psize, pgno, flags, nodes):
    """nodes: list of bytes (already even-sized)."""
    buf = bytearray(psize)
    upper = psize
    ptrs = []
    for nd in nodes:
        upper -= len(nd)
        buf[upper:upper + len(nd)] = nd
        ptrs.append(upper)
    lower = 16 + 2 * len(nodes)
    assert lower <= upper, "page overfull"
    struct.pack_into("<QHHHH", buf, 0, pgno, 0, flags, lower, upper)
    for i, off in enumerate(ptrs):
        struct.pack_into("<H", buf, 16 + 2 * i, off)
    return buf


def write_lmdb(path, records, psize=4096, max_keys=None):
    """records: dict or iterable of (key bytes, value bytes).  `max_keys` caps the entries per page (forces deeper trees)."""
    items = sorted(dict(records).items())
    pages = {}                       # pgno -> bytes (possibly several pages long for overflow runs)
    next_pg = [2]

    def alloc(n=1):
        p = next_pg[0]
        next_pg[0] += n
        return p

    nodemax = ((psize - 16) // 2) & ~1          # larger nodes move their value to overflow pages
    n_overflow = 0
    leaves, cur, cur_size, first_key = [], [], 0, None

    def flush_leaf():
        nonlocal cur, cur_size, first_key
        if cur:
            pg = alloc()
            pages[pg] = _page(psize, pg, P_LEAF, cur)
            leaves.append((first_key, pg))
        cur, cur_size, first_key = [], 0, None

    for k, v in items:
        assert 0 < len(k) <= 511
        if 8 + len(k) + len(v) > nodemax:
            npg = (15 + len(v)) // psize + 1
            opg = alloc(npg)
            run = bytearray(npg * psize)
            struct.pack_into("<QHHI", run, 0, opg, 0, P_OVERFLOW, npg)
            run[16:16 + len(v)] = v
            pages[opg] = run
            n_overflow += npg
            node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, F_BIGDATA, len(k)) + k + struct.pack("<Q", opg)
        else:
            node = struct.pack("<HHHH", len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + bytes(v)
        node += b"\0" * (_even(len(node)) - len(node))
        if cur and (16 + 2 * (len(cur) + 1) + cur_size + len(node) > psize or (max_keys and len(cur) >= max_keys)):
            flush_leaf()
        if not cur:
            first_key = k
        cur.append(node)
        cur_size += len(node)
    flush_leaf()

    level, depth, n_branch = leaves, 1 if leaves else 0, 0
    while len(level) > 1:
        up, cur, cur_size, first_key = [], [], 0, None
        for i, (k, child) in enumerate(level):
            key = b"" if not cur else k
            node = struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, (child >> 32) & 0xFFFF, len(key)) + key
            node += b"\0" * (_even(len(node)) - len(node))
            if cur and (16 + 2 * (len(cur) + 1) + cur_size + len(node) > psize or (max_keys and len(cur) >= max_keys)):
                pg = alloc()
                pages[pg] = _page(psize, pg, P_BRANCH, cur)
                up.append((first_key, pg))
                n_branch += 1
                cur, cur_size = [], 0
                node = struct.pack("<HHHH", child & 0xFFFF, (child >> 16) & 0xFFFF, (child >> 32) & 0xFFFF, 0)
            if not cur:
                first_key = k
            cur.append(node)
            cur_size += len(node)
        pg = alloc()
        pages[pg] = _page(psize, pg, P_BRANCH, cur)
        up.append((first_key, pg))
        n_branch += 1
        level = up
        depth += 1
    root = level[0][1] if level else INVALID
    last_pg = next_pg[0] - 1

    def meta(pgno, txnid, live):
        buf = bytearray(psize)
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, P_META, 0, 0)
        free_db = struct.pack("<IHHQQQQQ", psize, 0, 0, 0, 0, 0, 0, INVALID)
        if live:
            main_db = struct.pack("<IHHQQQQQ", 0, 0, depth, n_branch, len(leaves), n_overflow, len(items), root)
        else:
            main_db = struct.pack("<IHHQQQQQ", 0, 0, 0, 0, 0, 0, 0, INVALID)
        struct.pack_into("<IIQQ", buf, 16, 0xBEEFC0DE, 1, 0, 1 << 30)
        buf[16 + 24:16 + 72] = free_db
        buf[16 + 72:16 + 120] = main_db
        struct.pack_into("<QQ", buf, 16 + 120, last_pg if live else 1, txnid)
        return buf

    with open(path, "wb") as f:
        f.write(meta(0, 0, False))           # the older meta page: an empty database
        f.write(meta(1, 1, True))            # the newer one wins
        for pg in range(2, next_pg[0]):
            if pg in pages:
                f.write(pages[pg])
    return dict(depth=depth, branch_pages=n_branch, leaf_pages=len(leaves), overflow_pages=n_overflow, entries=len(items))


def pack_datapoint(dp, str_keys=False):
    """tensorpack's `dumps` for LMDBSerializer: msgpack with msgpack_numpy's ndarray / numpy-scalar encoding, use_bin_type=True.
    `str_keys`: the map keys as text, the way msgpack_numpy < 0.4.4 wrote them."""
    import msgpack
    import numpy as np

    def enc(o):
        if isinstance(o, np.ndarray):
            return {b"nd": True, b"type": o.dtype.str, b"kind": b"", b"shape": o.shape, b"data": o.tobytes()}
        if isinstance(o, (np.bool_, np.number)):
            return {b"nd": False, b"type": o.dtype.str, b"data": o.tobytes()}
        raise TypeError(type(o))

    def enc_str(o):
        return {k.decode(): v for k, v in enc(o).items()}

    return msgpack.packb(dp, default=enc_str if str_keys else enc, use_bin_type=True)
