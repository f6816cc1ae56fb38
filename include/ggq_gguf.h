/*
 * ggq_gguf.h -- C ABI of libggq_hip.so, part 2: reading a GGUF file and streaming its tensor data
 * into HBM.  This is the data format on the INPUT side of the dequant path (SURVEY.md section 8f
 * item 2): the reference gets it from the third-party `gguf` package
 *
 *     reader = gguf.GGUFReader(path)                      loader.py:55
 *     reader.tensors[i].name / .tensor_type / .shape / .data (an np.memmap view)   loader.py:60-106
 *     reader.get_field(key) -> .types / .parts / .data                              loader.py:16-49
 *
 * (gguf>=0.13.0, unpinned upper bound, pyproject.toml:6 -- absent from /root/reference and from this
 * image).  The file layout restated here is the public GGUF v2/v3 container format [ggml docs/gguf.md]:
 *
 *     "GGUF" u32 version | u64 n_tensors | u64 n_kv
 *     n_kv x   { string key | u32 value_type | value }
 *     n_tensors x { string name | u32 n_dims | u64 dims[n_dims] | u32 ggml_type | u64 offset }
 *     padding to `general.alignment` (default 32)
 *     tensor data; each tensor starts at data_start + offset, offsets are multiples of the alignment
 *
 *     string = u64 length + bytes (no terminator);  array value = u32 elem_type | u64 count | elements
 *     little-endian throughout (a byte-swapped version field is reported as GGQ_ERR_FORMAT).
 *     dims[0] is the FASTEST-varying (ggml) dimension: the torch shape is dims reversed (loader.py:110).
 *
 * Same conventions as ggq.h: plain pointers and sizes, ggq_status return codes, never throws/aborts.
 * Every pointer handed out (names, keys, value payloads, the mapping base) points into memory owned
 * by the handle and stays valid until ggq_gguf_close().  The parser bounds-checks every read: a
 * truncated or corrupt file yields GGQ_ERR_FORMAT, never a fault.
 */
#ifndef GGQ_GGUF_H
#define GGQ_GGUF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* additional ggq_status values (ggq.h holds 0..5) */
#define GGQ_ERR_IO 6      /* open / fstat / mmap / pread failed (errno is left set) */
#define GGQ_ERR_FORMAT 7  /* not a GGUF file, unsupported version, truncated or inconsistent */

/* metadata value types (the container's own enum; == gguf.GGUFValueType) */
enum {
    GGQ_KV_UINT8 = 0, GGQ_KV_INT8 = 1, GGQ_KV_UINT16 = 2, GGQ_KV_INT16 = 3, GGQ_KV_UINT32 = 4, GGQ_KV_INT32 = 5,
    GGQ_KV_FLOAT32 = 6, GGQ_KV_BOOL = 7, GGQ_KV_STRING = 8, GGQ_KV_ARRAY = 9, GGQ_KV_UINT64 = 10, GGQ_KV_INT64 = 11,
    GGQ_KV_FLOAT64 = 12
};

#define GGQ_GGUF_MAX_DIMS 8   /* ggml itself allows 4; tools/fix_5d_tensors.py exists because files exceed that */

typedef struct ggq_gguf ggq_gguf;

typedef struct ggq_gguf_info {
    uint32_t version;        /* 2 or 3 */
    uint32_t alignment;      /* general.alignment, default 32 */
    uint64_t n_tensors;
    uint64_t n_kv;
    uint64_t data_offset;    /* file offset of the tensor-data section */
    uint64_t data_bytes;     /* file_bytes - data_offset */
    uint64_t file_bytes;
    const void* base;        /* read-only mapping of the whole file (CPU views: base + data_offset + tensor.offset) */
} ggq_gguf_info;

typedef struct ggq_gguf_tensor {
    const char* name;        /* NUL-terminated copy */
    int32_t qtype;           /* ggml type id, as stored (may be one this library has no unpacker for) */
    uint32_t n_dims;
    uint64_t dims[GGQ_GGUF_MAX_DIMS];   /* ggml order: dims[0] fastest */
    uint64_t offset;         /* from the start of the data section */
    uint64_t nbytes;         /* n_elements / block_size * type_size; 0 if the type's geometry is unknown */
    uint64_t n_elements;
} ggq_gguf_tensor;

typedef struct ggq_gguf_kv {
    const char* key;         /* NUL-terminated copy */
    uint32_t type;           /* GGQ_KV_* */
    uint32_t elem_type;      /* element type when type == GGQ_KV_ARRAY, else == type */
    uint64_t count;          /* array length; 1 for scalars and strings */
    const void* data;        /* scalars / arrays of scalars: the little-endian payload inside the mapping;
                                a single string: its bytes; an array of strings: NULL (use ggq_gguf_kv_string) */
    uint64_t nbytes;         /* payload bytes (string: its length) */
} ggq_gguf_kv;

/* Map and parse `path`.  Replaces: gguf.GGUFReader(path) (loader.py:55). */
int ggq_gguf_open(const char* path, ggq_gguf** out);
void ggq_gguf_close(ggq_gguf* g);

int ggq_gguf_get_info(const ggq_gguf* g, ggq_gguf_info* out);

/* i-th tensor, in file order.  Replaces: reader.tensors[i] (loader.py:60,66,98-106). */
int ggq_gguf_get_tensor(const ggq_gguf* g, uint64_t i, ggq_gguf_tensor* out);

/* Index of metadata key `key`, -1 if absent.  Replaces: reader.get_field(key) is None (loader.py:18,27,41). */
int64_t ggq_gguf_find_kv(const ggq_gguf* g, const char* key);

/* i-th metadata entry.  Replaces: reader.get_field(key).types / .parts / .data (loader.py:22-24,31-36,43-47). */
int ggq_gguf_get_kv(const ggq_gguf* g, uint64_t i, ggq_gguf_kv* out);

/* Element `elem` of a string-typed entry (elem = 0 for a single string).  Not NUL-terminated. */
int ggq_gguf_kv_string(const ggq_gguf* g, uint64_t kv, uint64_t elem, const char** ptr, uint64_t* len);

/* (block_size, type_size) of a ggml type id, 0/0 if unknown.  Replaces: gguf.GGML_QUANT_SIZES[qtype]
 * for EVERY ggml type (ggq_block_size()/ggq_type_size() in ggq.h only cover types with an unpacker). */
int ggq_ggml_type_geometry(int qtype, uint32_t* block_size, uint32_t* type_size);

/* Stream `nbytes` of the tensor-data section, starting `offset` bytes into it, to device memory
 * `dev_dst` (current HIP device).  The file is read with pread() by `threads` host threads into
 * pinned staging buffers and copied host->device in `chunk_bytes` pieces on internal streams, reads
 * of chunk k+1 overlapping the DMA of chunk k.  Load-time call: returns when every byte has landed;
 * additionally `hip_stream` is made to wait for the copies, so work enqueued on it afterwards is
 * ordered without relying on the host-side wait; and the copies themselves start only after everything
 * already enqueued on `hip_stream` at entry (dev_dst may be a recycled block of a caching allocator that
 * queued kernels still touch).  threads <= 0 / chunk_bytes == 0 pick defaults
 * (8 threads, 16 MiB).  Replaces: `s.weight.to(device)` per layer on every forward in low-VRAM mode
 * (ops.py:209) by ONE sequential pass at load time -- 288 GB of HBM holds any supported model whole. */
int ggq_gguf_upload(const ggq_gguf* g, void* dev_dst, uint64_t offset, uint64_t nbytes, int threads, uint64_t chunk_bytes,
                    void* hip_stream);

/* Free the pinned staging buffers ggq_gguf_upload keeps between calls. */
void ggq_gguf_upload_release(void);

#ifdef __cplusplus
}
#endif
#endif /* GGQ_GGUF_H */
