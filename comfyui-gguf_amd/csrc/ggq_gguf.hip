// ggq_gguf.hip -- the C ABI of include/ggq_gguf.h: a bounds-checked GGUF container parser over a
// read-only mapping, and the file -> HBM streaming upload.  Host code only (HIP runtime calls for
// the pinned staging buffers and the async copies); nothing here knows about torch or about the
// third-party `gguf` package the reference reads the same files with (loader.py:55).
#include "../../include/ggq.h"
#include "../../include/ggq_gguf.h"

#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// ggml type id -> (elements per block, bytes per block): the public ggml table
// (== gguf.GGML_QUANT_SIZES, which the reference indexes at dequant.py:34).
struct Geometry { int id; uint32_t bs, ts; };
const Geometry GEOMETRY[] = {
    {0, 1, 4},      /* F32 */     {1, 1, 2},      /* F16 */     {2, 32, 18},    /* Q4_0 */    {3, 32, 20},    /* Q4_1 */
    {6, 32, 22},    /* Q5_0 */    {7, 32, 24},    /* Q5_1 */    {8, 32, 34},    /* Q8_0 */    {9, 32, 36},    /* Q8_1 */
    {10, 256, 84},  /* Q2_K */    {11, 256, 110}, /* Q3_K */    {12, 256, 144}, /* Q4_K */    {13, 256, 176}, /* Q5_K */
    {14, 256, 210}, /* Q6_K */    {15, 256, 292}, /* Q8_K */    {16, 256, 66},  /* IQ2_XXS */ {17, 256, 74},  /* IQ2_XS */
    {18, 256, 98},  /* IQ3_XXS */ {19, 256, 50},  /* IQ1_S */   {20, 32, 18},   /* IQ4_NL */  {21, 256, 110}, /* IQ3_S */
    {22, 256, 82},  /* IQ2_S */   {23, 256, 136}, /* IQ4_XS */  {24, 1, 1},     /* I8 */      {25, 1, 2},     /* I16 */
    {26, 1, 4},     /* I32 */     {27, 1, 8},     /* I64 */     {28, 1, 8},     /* F64 */     {29, 256, 56},  /* IQ1_M */
    {30, 1, 2},     /* BF16 */    {34, 256, 54},  /* TQ1_0 */   {35, 256, 66},  /* TQ2_0 */   {39, 32, 17},   /* MXFP4 */
};

const Geometry* geometry(int id)
{
    for (const Geometry& g : GEOMETRY)
        if (g.id == id) return &g;
    return nullptr;
}

uint32_t scalar_size(uint32_t t)
{
    switch (t) {
    case GGQ_KV_UINT8: case GGQ_KV_INT8: case GGQ_KV_BOOL: return 1;
    case GGQ_KV_UINT16: case GGQ_KV_INT16: return 2;
    case GGQ_KV_UINT32: case GGQ_KV_INT32: case GGQ_KV_FLOAT32: return 4;
    case GGQ_KV_UINT64: case GGQ_KV_INT64: case GGQ_KV_FLOAT64: return 8;
    default: return 0;
    }
}

struct KV {
    std::string key;
    uint32_t type = 0, elem_type = 0;
    uint64_t count = 1;
    uint64_t off = 0, nbytes = 0;           // payload inside the mapping
    std::vector<uint64_t> str_off;           // string arrays: offset of every element's length field
};

struct TInfo {
    std::string name;
    int32_t type = 0;
    uint32_t n_dims = 0;
    uint64_t dims[GGQ_GGUF_MAX_DIMS] = {};
    uint64_t offset = 0, nbytes = 0, n_elements = 0;
};

// A cursor over the mapping; every read is bounds-checked and a failure is sticky.
struct Cursor {
    const uint8_t* base;
    uint64_t size, pos = 0;
    bool ok = true;
    bool need(uint64_t n)
    {
        if (!ok || n > size - pos) { ok = false; return false; }   // pos <= size always
        return true;
    }
    template <class T> T get()
    {
        T v{};
        if (need(sizeof(T))) { std::memcpy(&v, base + pos, sizeof(T)); pos += sizeof(T); }
        return v;
    }
    // string = u64 length + bytes; returns (offset of the bytes, length)
    bool str(uint64_t& off, uint64_t& len)
    {
        len = get<uint64_t>();
        if (!need(len)) return false;
        off = pos;
        pos += len;
        return true;
    }
    bool skip(uint64_t n)
    {
        if (!need(n)) return false;
        pos += n;
        return true;
    }
};

}  // namespace

struct ggq_gguf {
    int fd = -1;
    const uint8_t* map = nullptr;
    uint64_t size = 0;
    uint32_t version = 0, alignment = 32;
    uint64_t data_off = 0;
    std::vector<KV> kvs;
    std::vector<TInfo> tensors;
    std::unordered_map<std::string, int64_t> kv_index;
};

namespace {

int parse(ggq_gguf* g)
{
    Cursor c{g->map, g->size};
    const uint32_t magic = c.get<uint32_t>();
    if (!c.ok || magic != 0x46554747u) return GGQ_ERR_FORMAT;          // "GGUF"
    g->version = c.get<uint32_t>();
    if (!c.ok || (g->version != 2 && g->version != 3)) return GGQ_ERR_FORMAT;   // v1 (u32 counts) and byte-swapped files
    const uint64_t n_tensors = c.get<uint64_t>(), n_kv = c.get<uint64_t>();
    // every entry takes at least 12 (kv) / 24 (tensor) bytes: rejects absurd counts before reserving
    if (!c.ok || n_kv > g->size / 12 || n_tensors > g->size / 24) return GGQ_ERR_FORMAT;
    g->kvs.reserve(n_kv);
    for (uint64_t i = 0; i < n_kv; i++) {
        KV kv;
        uint64_t ko, kl;
        if (!c.str(ko, kl) || kl > 65535) return GGQ_ERR_FORMAT;
        kv.key.assign(reinterpret_cast<const char*>(g->map + ko), kl);
        kv.type = kv.elem_type = c.get<uint32_t>();
        if (!c.ok) return GGQ_ERR_FORMAT;
        if (kv.type == GGQ_KV_STRING) {
            if (!c.str(kv.off, kv.nbytes)) return GGQ_ERR_FORMAT;
        } else if (kv.type == GGQ_KV_ARRAY) {
            kv.elem_type = c.get<uint32_t>();
            kv.count = c.get<uint64_t>();
            if (!c.ok) return GGQ_ERR_FORMAT;
            if (kv.elem_type == GGQ_KV_STRING) {
                if (kv.count > (g->size - c.pos) / 8) return GGQ_ERR_FORMAT;
                kv.off = c.pos;
                kv.str_off.reserve(kv.count);
                for (uint64_t e = 0; e < kv.count; e++) {
                    kv.str_off.push_back(c.pos);
                    uint64_t so, sl;
                    if (!c.str(so, sl)) return GGQ_ERR_FORMAT;
                }
                kv.nbytes = c.pos - kv.off;
            } else {
                const uint32_t es = scalar_size(kv.elem_type);
                if (es == 0 || kv.count > (g->size - c.pos) / es) return GGQ_ERR_FORMAT;   // nested arrays: not in any known file
                kv.off = c.pos;
                kv.nbytes = kv.count * es;
                if (!c.skip(kv.nbytes)) return GGQ_ERR_FORMAT;
            }
        } else {
            const uint32_t es = scalar_size(kv.type);
            if (es == 0) return GGQ_ERR_FORMAT;
            kv.off = c.pos;
            kv.nbytes = es;
            if (!c.skip(es)) return GGQ_ERR_FORMAT;
        }
        g->kv_index.emplace(kv.key, (int64_t)g->kvs.size());     // first occurrence wins
        g->kvs.push_back(std::move(kv));
    }
    auto it = g->kv_index.find("general.alignment");
    if (it != g->kv_index.end()) {
        const KV& kv = g->kvs[(size_t)it->second];
        // "general.alignment: uint32 ... must be a multiple of 8" (GGUF specification); gguf-py refuses any other value type too
        // ("Bad type for general.alignment field")
        if (kv.type != GGQ_KV_UINT32) return GGQ_ERR_FORMAT;
        uint32_t a;
        std::memcpy(&a, g->map + kv.off, 4);
        if (a == 0 || a % 8u != 0) return GGQ_ERR_FORMAT;
        g->alignment = a;
    }
    g->tensors.reserve(n_tensors);
    for (uint64_t i = 0; i < n_tensors; i++) {
        TInfo t;
        uint64_t no, nl;
        if (!c.str(no, nl) || nl > 65535) return GGQ_ERR_FORMAT;
        t.name.assign(reinterpret_cast<const char*>(g->map + no), nl);
        t.n_dims = c.get<uint32_t>();
        if (!c.ok || t.n_dims > GGQ_GGUF_MAX_DIMS) return GGQ_ERR_FORMAT;
        t.n_elements = 1;
        for (uint32_t d = 0; d < t.n_dims; d++) {
            t.dims[d] = c.get<uint64_t>();
            if (!c.ok || (t.dims[d] != 0 && t.n_elements > UINT64_MAX / t.dims[d])) return GGQ_ERR_FORMAT;
            t.n_elements *= t.dims[d];
        }
        t.type = (int32_t)c.get<uint32_t>();
        t.offset = c.get<uint64_t>();
        if (!c.ok) return GGQ_ERR_FORMAT;
        if (const Geometry* geo = geometry(t.type)) {
            // rows are whole blocks (ggml requires dims[0] % block_size == 0)
            if (t.n_dims > 0 && t.dims[0] % geo->bs != 0) return GGQ_ERR_FORMAT;
            const uint64_t blocks = t.n_elements / geo->bs;
            if (blocks > UINT64_MAX / geo->ts) return GGQ_ERR_FORMAT;
            t.nbytes = blocks * geo->ts;
        }
        g->tensors.push_back(std::move(t));
    }
    const uint64_t a = g->alignment;
    g->data_off = (c.pos + a - 1) / a * a;
    if (g->data_off > g->size) {
        if (n_tensors != 0) return GGQ_ERR_FORMAT;
        g->data_off = g->size;                     // metadata-only file without trailing padding
    }
    const uint64_t data_bytes = g->size - g->data_off;
    for (const TInfo& t : g->tensors) {
        if (t.offset % a != 0) return GGQ_ERR_FORMAT;
        if (t.offset > data_bytes || t.nbytes > data_bytes - t.offset) return GGQ_ERR_FORMAT;   // truncated file
    }
    return GGQ_OK;
}

// ---- pinned staging pool for the upload: kept between calls (pinning costs more than copying)
struct Staging {
    std::mutex mu;
    std::vector<void*> bufs;
    uint64_t chunk = 0;
    int device = -1;
    void release()
    {
        for (void* p : bufs) (void)hipHostFree(p);
        bufs.clear();
        chunk = 0;
    }
};
Staging g_staging;

}  // namespace

extern "C" {

int ggq_ggml_type_geometry(int qtype, uint32_t* block_size, uint32_t* type_size)
{
    const Geometry* g = geometry(qtype);
    if (block_size) *block_size = g ? g->bs : 0;
    if (type_size) *type_size = g ? g->ts : 0;
    return g ? GGQ_OK : GGQ_ERR_QTYPE;
}

int ggq_gguf_open(const char* path, ggq_gguf** out)
{
    if (!out) return GGQ_ERR_ARG;
    *out = nullptr;
    if (!path) return GGQ_ERR_ARG;
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return GGQ_ERR_IO;
    struct stat st;
    if (::fstat(fd, &st) != 0) { ::close(fd); return GGQ_ERR_IO; }
    if (st.st_size < 24) { ::close(fd); return GGQ_ERR_FORMAT; }
    void* map = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) { ::close(fd); return GGQ_ERR_IO; }
    ggq_gguf* g = new (std::nothrow) ggq_gguf();
    if (!g) { ::munmap(map, (size_t)st.st_size); ::close(fd); return GGQ_ERR_NOMEM; }
    g->fd = fd;
    g->map = static_cast<const uint8_t*>(map);
    g->size = (uint64_t)st.st_size;
    int rc;
    try {
        rc = parse(g);
    } catch (const std::bad_alloc&) {
        rc = GGQ_ERR_NOMEM;
    } catch (...) {
        rc = GGQ_ERR_FORMAT;
    }
    if (rc != GGQ_OK) { ggq_gguf_close(g); return rc; }
    *out = g;
    return GGQ_OK;
}

void ggq_gguf_close(ggq_gguf* g)
{
    if (!g) return;
    if (g->map) ::munmap(const_cast<uint8_t*>(g->map), (size_t)g->size);
    if (g->fd >= 0) ::close(g->fd);
    delete g;
}

int ggq_gguf_get_info(const ggq_gguf* g, ggq_gguf_info* out)
{
    if (!g || !out) return GGQ_ERR_ARG;
    out->version = g->version;
    out->alignment = g->alignment;
    out->n_tensors = g->tensors.size();
    out->n_kv = g->kvs.size();
    out->data_offset = g->data_off;
    out->data_bytes = g->size - g->data_off;
    out->file_bytes = g->size;
    out->base = g->map;
    return GGQ_OK;
}

int ggq_gguf_get_tensor(const ggq_gguf* g, uint64_t i, ggq_gguf_tensor* out)
{
    if (!g || !out || i >= g->tensors.size()) return GGQ_ERR_ARG;
    const TInfo& t = g->tensors[(size_t)i];
    out->name = t.name.c_str();
    out->qtype = t.type;
    out->n_dims = t.n_dims;
    std::memcpy(out->dims, t.dims, sizeof(t.dims));
    out->offset = t.offset;
    out->nbytes = t.nbytes;
    out->n_elements = t.n_elements;
    return GGQ_OK;
}

int64_t ggq_gguf_find_kv(const ggq_gguf* g, const char* key)
{
    if (!g || !key) return -1;
    try {
        auto it = g->kv_index.find(key);
        return it == g->kv_index.end() ? -1 : it->second;
    } catch (...) {
        return -1;
    }
}

int ggq_gguf_get_kv(const ggq_gguf* g, uint64_t i, ggq_gguf_kv* out)
{
    if (!g || !out || i >= g->kvs.size()) return GGQ_ERR_ARG;
    const KV& kv = g->kvs[(size_t)i];
    out->key = kv.key.c_str();
    out->type = kv.type;
    out->elem_type = kv.elem_type;
    out->count = kv.count;
    out->nbytes = kv.nbytes;
    out->data = (kv.type == GGQ_KV_ARRAY && kv.elem_type == GGQ_KV_STRING) ? nullptr : g->map + kv.off;
    return GGQ_OK;
}

int ggq_gguf_kv_string(const ggq_gguf* g, uint64_t i, uint64_t elem, const char** ptr, uint64_t* len)
{
    if (!g || !ptr || !len || i >= g->kvs.size()) return GGQ_ERR_ARG;
    const KV& kv = g->kvs[(size_t)i];
    if (kv.type == GGQ_KV_STRING) {
        if (elem != 0) return GGQ_ERR_ARG;
        *ptr = reinterpret_cast<const char*>(g->map + kv.off);
        *len = kv.nbytes;
        return GGQ_OK;
    }
    if (kv.type != GGQ_KV_ARRAY || kv.elem_type != GGQ_KV_STRING || elem >= kv.count) return GGQ_ERR_ARG;
    const uint64_t o = kv.str_off[(size_t)elem];       // validated at parse time
    uint64_t l;
    std::memcpy(&l, g->map + o, 8);
    *ptr = reinterpret_cast<const char*>(g->map + o + 8);
    *len = l;
    return GGQ_OK;
}

void ggq_gguf_upload_release(void)
{
    std::lock_guard<std::mutex> lock(g_staging.mu);
    g_staging.release();
}

int ggq_gguf_upload(const ggq_gguf* g, void* dev_dst, uint64_t offset, uint64_t nbytes, int threads, uint64_t chunk_bytes,
                    void* hip_stream)
{
    if (!g) return GGQ_ERR_ARG;
    const uint64_t data_bytes = g->size - g->data_off;
    if (offset > data_bytes || nbytes > data_bytes - offset) return GGQ_ERR_ARG;
    if (nbytes == 0) return GGQ_OK;
    if (!dev_dst) return GGQ_ERR_ARG;
    if (threads <= 0) threads = 8;
    threads = std::min(threads, 64);
    if (chunk_bytes == 0) chunk_bytes = 16ull << 20;
    chunk_bytes = (chunk_bytes + 4095) & ~4095ull;
    const uint64_t n_chunks = (nbytes + chunk_bytes - 1) / chunk_bytes;
    threads = (int)std::min<uint64_t>((uint64_t)threads, n_chunks);
    constexpr int DEPTH = 2;                         // staging buffers per thread: read k+1 while k is in flight

    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return GGQ_ERR_HIP;
    std::lock_guard<std::mutex> lock(g_staging.mu);   // one upload at a time per process: they would only share the link
    if (g_staging.chunk != chunk_bytes || g_staging.device != device || (int)g_staging.bufs.size() < threads * DEPTH) {
        g_staging.release();
        g_staging.chunk = chunk_bytes;
        g_staging.device = device;
        for (int i = 0; i < threads * DEPTH; i++) {
            void* p = nullptr;
            if (hipHostMalloc(&p, chunk_bytes, hipHostMallocDefault) != hipSuccess) {
                g_staging.release();
                return GGQ_ERR_NOMEM;
            }
            g_staging.bufs.push_back(p);
        }
    }

    // dev_dst comes from the caller's allocator on the caller's stream: a caching allocator may hand out a block that kernels
    // ALREADY QUEUED on that stream still read or write (a model swapped in one process).  The workers copy on private
    // non-blocking streams, so each of them first waits for everything the caller's stream holds at this point.
    hipEvent_t entry = nullptr;
    if (hipEventCreateWithFlags(&entry, hipEventDisableTiming) != hipSuccess) return GGQ_ERR_HIP;
    if (hipEventRecord(entry, static_cast<hipStream_t>(hip_stream)) != hipSuccess) {
        (void)hipEventDestroy(entry);
        return GGQ_ERR_HIP;
    }

    std::atomic<int> status{GGQ_OK};
    std::vector<hipEvent_t> done((size_t)threads, nullptr);
    auto worker = [&](int t) {
        if (hipSetDevice(device) != hipSuccess) { status = GGQ_ERR_HIP; return; }
        hipStream_t s = nullptr;
        hipEvent_t ev[DEPTH] = {};
        bool used[DEPTH] = {};
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { status = GGQ_ERR_HIP; return; }
        if (hipStreamWaitEvent(s, entry, 0) != hipSuccess) status = GGQ_ERR_HIP;
        for (int d = 0; d < DEPTH; d++)
            if (hipEventCreateWithFlags(&ev[d], hipEventDisableTiming) != hipSuccess) status = GGQ_ERR_HIP;
        uint64_t it = 0;
        for (uint64_t k = (uint64_t)t; k < n_chunks && status == GGQ_OK; k += (uint64_t)threads, it++) {
            const int d = (int)(it % DEPTH);
            uint8_t* buf = static_cast<uint8_t*>(g_staging.bufs[(size_t)(t * DEPTH + d)]);
            if (used[d] && hipEventSynchronize(ev[d]) != hipSuccess) { status = GGQ_ERR_HIP; break; }   // buffer free again
            const uint64_t o = k * chunk_bytes, len = std::min(chunk_bytes, nbytes - o);
            uint64_t got = 0;
            while (got < len) {
                const ssize_t r = ::pread(g->fd, buf + got, (size_t)(len - got), (off_t)(g->data_off + offset + o + got));
                if (r <= 0) { status = GGQ_ERR_IO; break; }
                got += (uint64_t)r;
            }
            if (status != GGQ_OK) break;
            if (hipMemcpyAsync(static_cast<uint8_t*>(dev_dst) + o, buf, (size_t)len, hipMemcpyHostToDevice, s) != hipSuccess ||
                hipEventRecord(ev[d], s) != hipSuccess) { status = GGQ_ERR_HIP; break; }
            used[d] = true;
        }
        // hand the caller one event that covers everything this thread enqueued
        hipEvent_t fin = nullptr;
        if (hipEventCreateWithFlags(&fin, hipEventDisableTiming) == hipSuccess && hipEventRecord(fin, s) == hipSuccess) done[(size_t)t] = fin;
        else status = GGQ_ERR_HIP;
        if (hipStreamSynchronize(s) != hipSuccess) status = GGQ_ERR_HIP;
        for (int d = 0; d < DEPTH; d++)
            if (ev[d]) (void)hipEventDestroy(ev[d]);
        (void)hipStreamDestroy(s);
    };
    std::vector<std::thread> pool;
    try {
        pool.reserve((size_t)threads);
    } catch (...) {
        (void)hipEventDestroy(entry);
        return GGQ_ERR_NOMEM;
    }
    for (int t = 0; t < threads; t++) {
        try {
            pool.emplace_back(worker, t);            // no reallocation after reserve(); the ctor itself may throw
        } catch (...) {
            status = GGQ_ERR_NOMEM;                  // the threads already running see it and stop
            break;
        }
    }
    for (std::thread& th : pool) th.join();
    (void)hipEventDestroy(entry);
    for (hipEvent_t e : done) {
        if (!e) continue;
        if (status == GGQ_OK && hipStreamWaitEvent(static_cast<hipStream_t>(hip_stream), e, 0) != hipSuccess) status = GGQ_ERR_HIP;
        (void)hipEventDestroy(e);
    }
    return status.load();
}

}  // extern "C"
