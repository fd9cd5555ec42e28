// Read-only walker of the LMDB on-disk format (the data set of /root/reference/dataset.py:173-183,213-216, written by
// /root/reference/utils/make_lmdb.py:39-75 through py-lmdb 1.x / liblmdb 0.9: data format version 1).
//
// liblmdb is not part of this image, so the reference's `lmdb.open(readonly=True, lock=False)` + `txn.get(key)` +
// `txn.stat()['entries']` are restated here over an mmap of data.mdb.  Format facts used (liblmdb 0.9 mdb.c, restated
// from its published layout; no fixture written by liblmdb itself was available to check against):
//   * pages of `psize` bytes; page header 16 bytes: pgno u64, pad u16, flags u16, lower u16, upper u16 (an overflow
//     page keeps its page count as u32 in the lower/upper field); node offsets (u16) follow the header;
//   * pages 0 and 1 are meta pages: header, then magic u32 0xBEEFC0DE, version u32, address u64, mapsize u64, two
//     48-byte database records (free list, main: pad u32 | flags u16 | depth u16 | branch, leaf, overflow pages u64 |
//     entries u64 | root u64), last page u64, transaction id u64; the live meta page is the one with the larger id;
//     the page size is the `pad` field of the free-list record;
//   * a node is lo u16, hi u16, flags u16, ksize u16, key bytes, data; leaf: data size = lo | hi << 16, flag 0x01 =
//     the data is a u64 overflow page number whose pages hold the value contiguously from byte 16 on; branch: child
//     page = lo | hi << 16 | flags << 32, the key is the smallest key of the child (empty for node 0);
//   * keys are ordered by memcmp, shorter key first on a tie.
// Sub-databases, duplicate-sorted data and LEAF2 pages are not produced by make_lmdb.py and are refused.
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>

#define MI355_ERR_ARG (-1)
#define MI355_ERR_UNSUPPORTED (-3)
int mi355_fail(int code, const char* fmt, ...);

namespace {

constexpr uint32_t MAGIC = 0xBEEFC0DEu;
constexpr int HDR = 16;
enum { P_BRANCH = 0x01, P_LEAF = 0x02, P_OVERFLOW = 0x04, P_META = 0x08, P_LEAF2 = 0x20 };
enum { F_BIGDATA = 0x01, F_SUBDATA = 0x02, F_DUPDATA = 0x04 };

struct Env {
    const uint8_t* map = nullptr;
    size_t size = 0;
    uint32_t psize = 0;
    uint64_t root = 0, entries = 0, last_pg = 0;
    uint16_t depth = 0;
    int fd = -1;
};

template <typename T>
T rd(const uint8_t* p) {
    T v;
    memcpy(&v, p, sizeof(T));
    return v;
}

struct Meta {
    bool ok = false;
    uint32_t psize = 0;
    uint16_t depth = 0, flags = 0;
    uint64_t root = 0, entries = 0, last_pg = 0, txnid = 0;
};

Meta read_meta(const uint8_t* page) {
    Meta m;
    if (!(rd<uint16_t>(page + 10) & P_META)) return m;
    const uint8_t* mm = page + HDR;
    if (rd<uint32_t>(mm) != MAGIC || rd<uint32_t>(mm + 4) != 1) return m;
    const uint8_t* free_db = mm + 24;
    const uint8_t* main_db = free_db + 48;
    m.psize = rd<uint32_t>(free_db);
    m.flags = rd<uint16_t>(main_db + 4);
    m.depth = rd<uint16_t>(main_db + 6);
    m.entries = rd<uint64_t>(main_db + 32);
    m.root = rd<uint64_t>(main_db + 40);
    m.last_pg = rd<uint64_t>(main_db + 48);
    m.txnid = rd<uint64_t>(main_db + 56);
    m.ok = true;
    return m;
}

// memcmp order, shorter key first on a common prefix
int key_cmp(const uint8_t* a, size_t an, const uint8_t* b, size_t bn) {
    const int c = memcmp(a, b, an < bn ? an : bn);
    if (c) return c;
    return an < bn ? -1 : (an > bn ? 1 : 0);
}

const uint8_t* page_at(const Env* e, uint64_t pgno) {
    if (pgno > e->last_pg || (pgno + 1) * (uint64_t)e->psize > e->size) return nullptr;
    return e->map + pgno * e->psize;
}

}  // namespace

extern "C" {

// path: the environment directory (holding data.mdb, as `lmdb.open(path)` creates it) or the data file itself.
void* mi355_lmdb_open(const char* path) {
    if (!path) { mi355_fail(MI355_ERR_ARG, "lmdb_open: null path"); return nullptr; }
    std::string file = path;
    struct stat st;
    if (stat(file.c_str(), &st) != 0) { mi355_fail(MI355_ERR_ARG, "lmdb_open: %s does not exist", path); return nullptr; }
    if (S_ISDIR(st.st_mode)) {
        file += "/data.mdb";
        if (stat(file.c_str(), &st) != 0) { mi355_fail(MI355_ERR_ARG, "lmdb_open: %s has no data.mdb", path); return nullptr; }
    }
    if (st.st_size < 2 * 512) { mi355_fail(MI355_ERR_ARG, "lmdb_open: %s is too small to hold two meta pages", file.c_str()); return nullptr; }
    const int fd = open(file.c_str(), O_RDONLY);
    if (fd < 0) { mi355_fail(MI355_ERR_ARG, "lmdb_open: cannot open %s", file.c_str()); return nullptr; }
    void* map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    if (map == MAP_FAILED) { close(fd); mi355_fail(MI355_ERR_ARG, "lmdb_open: mmap of %s failed", file.c_str()); return nullptr; }
    Env* e = new Env;
    e->map = (const uint8_t*)map; e->size = (size_t)st.st_size; e->fd = fd;
    // meta page 0 tells the page size, which places meta page 1; when page 0 itself does not validate (a torn write:
    // liblmdb then runs on page 1) page 1 is looked for at the page sizes liblmdb can have used - the common 4096
    // first, then every power of two from 512 to 64 KiB - and must name the size it was found at
    const Meta m0 = read_meta(e->map);
    Meta best = m0;
    if (m0.ok && m0.psize >= 512 && !(m0.psize & (m0.psize - 1)) && (size_t)2 * m0.psize <= e->size) {
        const Meta m1 = read_meta(e->map + m0.psize);
        if (m1.ok && m1.psize == m0.psize && m1.txnid > m0.txnid) best = m1;
    } else {
        best = Meta();
        static const uint32_t sizes[] = {4096, 512, 1024, 2048, 8192, 16384, 32768, 65536};
        for (const uint32_t ps : sizes) {
            if ((size_t)2 * ps > e->size) continue;
            const Meta m1 = read_meta(e->map + ps);
            if (m1.ok && m1.psize == ps) { best = m1; break; }
        }
    }
    if (!best.ok || best.psize < 512 || (best.psize & (best.psize - 1))) {
        mi355_fail(MI355_ERR_ARG, "lmdb_open: %s is not an LMDB data file (format version 1)", file.c_str());
        munmap(map, e->size); close(fd); delete e;
        return nullptr;
    }
    if (best.flags & 0x7E) {        // anything but MDB_REVERSEKEY-free plain keys: dupsort, integer keys, ... (make_lmdb.py sets none)
        mi355_fail(MI355_ERR_UNSUPPORTED, "lmdb_open: main database flags 0x%x are not supported (plain byte-string keys only)", best.flags);
        munmap(map, e->size); close(fd); delete e;
        return nullptr;
    }
    e->psize = best.psize; e->root = best.root; e->entries = best.entries; e->depth = best.depth; e->last_pg = best.last_pg;
    return e;
}

long long mi355_lmdb_entries(void* env) { return env ? (long long)((Env*)env)->entries : -1; }
int mi355_lmdb_depth(void* env) { return env ? (int)((Env*)env)->depth : -1; }
int mi355_lmdb_page_size(void* env) { return env ? (int)((Env*)env)->psize : -1; }

// 0: found (*data points into the map, valid until close; *size its length), 1: no such key, < 0: damaged file.
int mi355_lmdb_get(void* env, const void* key, int klen, const void** data, long long* size) {
    if (!env || !key || klen < 0 || !data || !size) return mi355_fail(MI355_ERR_ARG, "lmdb_get: bad arguments");
    const Env* e = (const Env*)env;
    *data = nullptr; *size = 0;
    if (e->entries == 0 || e->root == ~0ull) return 1;
    const uint8_t* k = (const uint8_t*)key;
    uint64_t pgno = e->root;
    for (int level = 0; level < 64; ++level) {
        const uint8_t* pg = page_at(e, pgno);
        if (!pg) return mi355_fail(MI355_ERR_ARG, "lmdb_get: page %llu lies outside the file", (unsigned long long)pgno);
        const uint16_t flags = rd<uint16_t>(pg + 10), lower = rd<uint16_t>(pg + 12);
        if (flags & P_LEAF2) return mi355_fail(MI355_ERR_UNSUPPORTED, "lmdb_get: fixed-size-key leaf pages are not supported");
        if (lower < HDR || lower > e->psize) return mi355_fail(MI355_ERR_ARG, "lmdb_get: damaged page %llu", (unsigned long long)pgno);
        const int n = (lower - HDR) / 2;
        auto node = [&](int i) -> const uint8_t* {
            const uint16_t off = rd<uint16_t>(pg + HDR + 2 * i);
            return (off >= HDR && off + 8u <= e->psize) ? pg + off : nullptr;
        };
        if (flags & P_BRANCH) {
            if (n < 1) return mi355_fail(MI355_ERR_ARG, "lmdb_get: empty branch page %llu", (unsigned long long)pgno);
            // the last node whose key is <= k; node 0 carries no key and stands for "everything smaller"
            int lo = 0, hi = n - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) / 2;
                const uint8_t* nd = node(mid);
                if (!nd) return mi355_fail(MI355_ERR_ARG, "lmdb_get: damaged branch page %llu", (unsigned long long)pgno);
                const uint16_t ks = rd<uint16_t>(nd + 6);
                if (nd + 8 + ks > pg + e->psize) return mi355_fail(MI355_ERR_ARG, "lmdb_get: damaged branch node");
                if (key_cmp(nd + 8, ks, k, (size_t)klen) <= 0) lo = mid; else hi = mid - 1;
            }
            const uint8_t* nd = node(lo);
            if (!nd) return mi355_fail(MI355_ERR_ARG, "lmdb_get: damaged branch page %llu", (unsigned long long)pgno);
            pgno = (uint64_t)rd<uint16_t>(nd) | ((uint64_t)rd<uint16_t>(nd + 2) << 16) | ((uint64_t)rd<uint16_t>(nd + 4) << 32);
            continue;
        }
        if (!(flags & P_LEAF)) return mi355_fail(MI355_ERR_ARG, "lmdb_get: page %llu is neither branch nor leaf", (unsigned long long)pgno);
        int lo = 0, hi = n - 1;
        while (lo <= hi) {
            const int mid = (lo + hi) / 2;
            const uint8_t* nd = node(mid);
            if (!nd) return mi355_fail(MI355_ERR_ARG, "lmdb_get: damaged leaf page %llu", (unsigned long long)pgno);
            const uint16_t ks = rd<uint16_t>(nd + 6), nf = rd<uint16_t>(nd + 4);
            if (nd + 8 + ks > pg + e->psize) return mi355_fail(MI355_ERR_ARG, "lmdb_get: damaged leaf node");
            const int c = key_cmp(nd + 8, ks, k, (size_t)klen);
            if (c < 0) { lo = mid + 1; continue; }
            if (c > 0) { hi = mid - 1; continue; }
            if (nf & (F_SUBDATA | F_DUPDATA)) return mi355_fail(MI355_ERR_UNSUPPORTED, "lmdb_get: sub-databases / duplicate data are not supported");
            const uint64_t dsize = (uint64_t)rd<uint16_t>(nd) | ((uint64_t)rd<uint16_t>(nd + 2) << 16);
            if (nf & F_BIGDATA) {
                if (nd + 8 + ks + 8 > pg + e->psize) return mi355_fail(MI355_ERR_ARG, "lmdb_get: damaged overflow reference");
                const uint64_t opg = rd<uint64_t>(nd + 8 + ks);
                const uint8_t* ov = page_at(e, opg);
                if (!ov || !(rd<uint16_t>(ov + 10) & P_OVERFLOW)) return mi355_fail(MI355_ERR_ARG, "lmdb_get: page %llu is not an overflow page", (unsigned long long)opg);
                const uint64_t npages = rd<uint32_t>(ov + 12);
                if (HDR + dsize > npages * e->psize || (opg + npages) * (uint64_t)e->psize > e->size)
                    return mi355_fail(MI355_ERR_ARG, "lmdb_get: value of %llu bytes does not fit its %llu overflow pages", (unsigned long long)dsize, (unsigned long long)npages);
                *data = ov + HDR;
            } else {
                if (nd + 8 + ks + dsize > pg + e->psize) return mi355_fail(MI355_ERR_ARG, "lmdb_get: value runs past its page");
                *data = nd + 8 + ks;
            }
            *size = (long long)dsize;
            return 0;
        }
        return 1;
    }
    return mi355_fail(MI355_ERR_ARG, "lmdb_get: tree deeper than 64 levels (cycle?)");
}

int mi355_lmdb_close(void* env) {
    if (!env) return 0;
    Env* e = (Env*)env;
    munmap((void*)e->map, e->size);
    close(e->fd);
    delete e;
    return 0;
}

}  // extern "C"
