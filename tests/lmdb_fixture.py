"""Test-side WRITER of the LMDB file format (py-lmdb / liblmdb are not installed, and the reference ships no data
base): builds a data.mdb the way liblmdb 0.9 lays one out - two meta pages, leaf pages with inline and overflow values,
branch pages bottom-up - from a {key: value} dictionary, so that the library's reader can be exercised on multi-level
trees.  Both the reader and this writer restate the published layout; no file written by liblmdb itself was available
to check them against ("parity unpinned" for the data plane)."""
import struct

PSIZE = 4096
HDR = 16
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA = 0x01
NODEMAX = (((PSIZE - HDR) // 2) & ~1) - 2          # liblmdb's me_nodemax: larger nodes go to overflow pages


def _even(n):
    return (n + 1) & ~1


def _page(pgno, flags, nodes):
    """nodes: list of bytes (header + key + data), placed from the page end downwards; returns PSIZE bytes."""
    buf = bytearray(PSIZE)
    upper = PSIZE
    ptrs = []
    for nd in nodes:
        upper -= _even(len(nd))
        buf[upper:upper + len(nd)] = nd
        ptrs.append(upper)
    lower = HDR + 2 * len(nodes)
    assert lower <= upper, "page overflow"
    struct.pack_into("<QHHHH", buf, 0, pgno, 0, flags, lower, upper)
    for i, o in enumerate(ptrs):
        struct.pack_into("<H", buf, HDR + 2 * i, o)
    return bytes(buf)


def _leaf_node(key, dsize, payload, flags):
    return struct.pack("<HHHH", dsize & 0xFFFF, dsize >> 16, flags, len(key)) + key + payload


def _branch_node(key, pgno):
    return struct.pack("<HHHH", pgno & 0xFFFF, (pgno >> 16) & 0xFFFF, (pgno >> 32) & 0xFFFF, len(key)) + key


def write_lmdb(path, items, txnid=1, fill=0.6):
    """items: {bytes key: bytes value}.  Writes `path` (a data.mdb file) and returns {'depth', 'pages', 'entries'}."""
    keys = sorted(items)
    pages = {}                       # pgno -> bytes (PSIZE, or n*PSIZE for overflow runs)
    next_pg = 2
    n_leaf = n_branch = n_over = 0
    budget = int((PSIZE - HDR) * fill)
    # ---- leaves
    level = []                       # (first key, pgno)
    cur, cur_size, first = [], 0, None
    def flush_leaf():
        nonlocal cur, cur_size, first, next_pg, n_leaf
        if not cur:
            return
        pages[next_pg] = _page(next_pg, P_LEAF, cur)
        level.append((first, next_pg))
        next_pg += 1
        n_leaf += 1
        cur, cur_size, first = [], 0, None
    for k in keys:
        v = items[k]
        if 8 + len(k) + len(v) > NODEMAX:
            npg = (HDR + len(v) + PSIZE - 1) // PSIZE
            ov = bytearray(npg * PSIZE)
            struct.pack_into("<QHHI", ov, 0, next_pg, 0, P_OVERFLOW, npg)
            ov[HDR:HDR + len(v)] = v
            # liblmdb allocates the overflow run while the leaf is still open, so leaf page numbers come later
            opg = next_pg
            pages[opg] = bytes(ov)
            next_pg += npg
            n_over += npg
            nd = _leaf_node(k, len(v), struct.pack("<Q", opg), F_BIGDATA)
        else:
            nd = _leaf_node(k, len(v), v, 0)
        need = _even(len(nd)) + 2
        if cur and cur_size + need > budget:
            flush_leaf()
        if first is None:
            first = k
        cur.append(nd)
        cur_size += need
    flush_leaf()
    depth = 1 if level else 0
    # ---- branches, bottom-up
    while len(level) > 1:
        up, cur, cur_size, first = [], [], 0, None
        for i, (k, pg) in enumerate(level):
            nd = _branch_node(b"" if not cur else k, pg)
            need = _even(len(nd)) + 2
            if cur and cur_size + need > budget:
                pages[next_pg] = _page(next_pg, P_BRANCH, cur)
                up.append((first, next_pg))
                next_pg += 1
                n_branch += 1
                cur, cur_size, first = [], 0, None
                nd = _branch_node(b"", pg)
                need = _even(len(nd)) + 2
            if first is None:
                first = k
            cur.append(nd)
            cur_size += need
        pages[next_pg] = _page(next_pg, P_BRANCH, cur)
        up.append((first, next_pg))
        next_pg += 1
        n_branch += 1
        level = up
        depth += 1
    root = level[0][1] if level else 0xFFFFFFFFFFFFFFFF
    last_pg = next_pg - 1

    def meta(pgno, tid, root_, entries, depth_):
        buf = bytearray(PSIZE)
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, P_META, 0, 0)
        o = HDR
        struct.pack_into("<IIQQ", buf, o, 0xBEEFC0DE, 1, 0, 1 << 30)
        o += 24
        struct.pack_into("<IHHQQQQQ", buf, o, PSIZE, 0, 0, 0, 0, 0, 0, 0xFFFFFFFFFFFFFFFF)          # free-list db (empty)
        o += 48
        struct.pack_into("<IHHQQQQQ", buf, o, 0, 0, depth_, n_branch, n_leaf, n_over, entries, root_)
        o += 48
        struct.pack_into("<QQ", buf, o, last_pg, tid)
        return bytes(buf)

    with open(path, "wb") as f:
        # the older meta page describes an empty data base: a reader must pick the one with the larger transaction id
        stale = meta(0, txnid - 1, 0xFFFFFFFFFFFFFFFF, 0, 0)
        live = meta(1, txnid, root, len(keys), depth)
        f.write(stale if txnid % 2 else live)
        f.write(live if txnid % 2 else stale)
        pg = 2
        while pg <= last_pg:
            blob = pages[pg]
            f.write(blob)
            pg += len(blob) // PSIZE
    return {"depth": depth, "pages": last_pg + 1, "entries": len(keys), "leaf": n_leaf, "branch": n_branch, "overflow": n_over}
