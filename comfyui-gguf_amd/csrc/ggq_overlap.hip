// ggq_overlap.hip -- "layer i+1 while layer i computes" (include/ggq.h ggq_overlap_*): a copy stream and an unpack stream beside the
// caller's stream, ordered among each other and against it by HIP events only -- the host never waits.
//
//   copy stream    host -> device copies of packed bytes into staging slots; a slot is rewritten only after the unpack that last read it
//   unpack stream  ggq_dequant of a staged (or already resident) packed weight into a dense scratch slot; starts after the copy it needs AND
//                  after everything the caller's stream held when the prefetch was requested (the dense slot's previous consumer)
//   caller stream  waits for a dense slot's "done" event before the GEMM that reads it
// Plain host code over the public C entry point ggq_dequant: nothing here knows about kernels or torch.
#include <hip/hip_runtime.h>
#include <new>

#include "ggq_host.hpp"
#include "../../include/ggq.h"

namespace {
constexpr int MAX_SLOTS = 16;
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// Under HIP-graph capture the overlap protocol cannot work (events recorded outside the capture, side streams that never rejoin it): the
// entry points that touch the caller's stream refuse, the host side (overlap.py eligible()) does not even try.
bool capturing(void* stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(static_cast<hipStream_t>(stream), &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}
}  // namespace

struct ggq_overlap {
    int device = 0, n_slots = 0;
    hipStream_t unpack = nullptr, copy = nullptr, copy2 = nullptr;   // copy2: the second half of every staged weight (two DMA queues keep the link fuller than one)
    hipEvent_t half = nullptr;
    hipEvent_t main_mark[MAX_SLOTS] = {}, done[MAX_SLOTS] = {};        // per dense slot
    hipEvent_t copied[MAX_SLOTS] = {}, consumed[MAX_SLOTS] = {};       // per packed staging slot
    bool consumed_valid[MAX_SLOTS] = {};
};

extern "C" {

int ggq_overlap_create(int n_slots, ggq_overlap** out)
{
    if (!out || n_slots < 1 || n_slots > MAX_SLOTS) return GGQ_ERR_ARG;
    *out = nullptr;
    ggq_overlap* ov = new (std::nothrow) ggq_overlap();
    if (!ov) return GGQ_ERR_NOMEM;
    ov->n_slots = n_slots;
    hipError_t e = hipGetDevice(&ov->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ov->unpack, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ov->copy, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ov->copy2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ov->half, hipEventDisableTiming);
    for (int i = 0; i < n_slots && e == hipSuccess; i++) {
        hipEvent_t* evs[4] = {&ov->main_mark[i], &ov->done[i], &ov->copied[i], &ov->consumed[i]};
        for (hipEvent_t* ev : evs)
            if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        ggq_overlap_destroy(ov);
        return ggq::hip_fail(e);
    }
    *out = ov;
    return GGQ_OK;
}

int ggq_overlap_copy(ggq_overlap* ov, int staging_slot, const void* host_packed, void* dev_packed, uint64_t packed_bytes)
{
    if (!ov || staging_slot < 0 || staging_slot >= ov->n_slots || (packed_bytes && (!host_packed || !dev_packed))) return GGQ_ERR_ARG;
    hipError_t e = hipSuccess;
    // weights of a few MB and more go as two halves on two copy streams; `copied` is recorded on the first after it has waited for the second
    const uint64_t split = packed_bytes >= (4ull << 20) ? ((packed_bytes / 2 + 4095) & ~4095ull) : packed_bytes;
    if (ov->consumed_valid[staging_slot]) {                                                   // the unpack that last read this slot
        e = hipStreamWaitEvent(ov->copy, ov->consumed[staging_slot], 0);
        if (e == hipSuccess && split < packed_bytes) e = hipStreamWaitEvent(ov->copy2, ov->consumed[staging_slot], 0);
    }
    // ... and the slot's previous COPY: a tenant that was staged but never unpacked (a mispredicted layer) leaves `consumed` stale, and its
    // first half -- on `copy` -- may still be landing in the bytes this tenant's second half -- on `copy2` -- is about to write.  `copied` was
    // recorded on `copy` after it had waited for that tenant's second half, so it covers both (waiting on a never-recorded event is a no-op).
    if (e == hipSuccess && split < packed_bytes) e = hipStreamWaitEvent(ov->copy2, ov->copied[staging_slot], 0);
    if (e == hipSuccess && split) e = hipMemcpyAsync(dev_packed, host_packed, (size_t)split, hipMemcpyHostToDevice, ov->copy);
    if (e == hipSuccess && split < packed_bytes) {
        e = hipMemcpyAsync(static_cast<uint8_t*>(dev_packed) + split, static_cast<const uint8_t*>(host_packed) + split, (size_t)(packed_bytes - split),
                           hipMemcpyHostToDevice, ov->copy2);
        if (e == hipSuccess) e = hipEventRecord(ov->half, ov->copy2);
        if (e == hipSuccess) e = hipStreamWaitEvent(ov->copy, ov->half, 0);
    }
    if (e == hipSuccess) e = hipEventRecord(ov->copied[staging_slot], ov->copy);
    return e == hipSuccess ? GGQ_OK : ggq::hip_fail(e);
}

int ggq_overlap_prefetch(ggq_overlap* ov, int slot, int staging_slot, int qtype, const void* dev_packed, uint64_t n_blocks, void* out,
                         int compute_dtype, int out_dtype, void* main_stream)
{
    if (!ov || slot < 0 || slot >= ov->n_slots || staging_slot >= ov->n_slots) return GGQ_ERR_ARG;
    if (!ggq_supported(qtype)) return GGQ_ERR_QTYPE;
    if (out_dtype < 0 || out_dtype > 2 || compute_dtype < 0 || compute_dtype > 2) return GGQ_ERR_ARG;
    if (n_blocks && (!dev_packed || !out)) return GGQ_ERR_ARG;
    if (n_blocks && (!aligned16(dev_packed) || !aligned16(out))) return GGQ_ERR_ALIGN;
    if (capturing(main_stream)) return GGQ_ERR_ARG;      // the side streams would be pulled into the capture and never joined back
    hipError_t e = hipEventRecord(ov->main_mark[slot], static_cast<hipStream_t>(main_stream));
    if (e == hipSuccess) e = hipStreamWaitEvent(ov->unpack, ov->main_mark[slot], 0);
    if (e == hipSuccess && staging_slot >= 0) e = hipStreamWaitEvent(ov->unpack, ov->copied[staging_slot], 0);
    if (e != hipSuccess) return ggq::hip_fail(e);
    const int rc = ggq_dequant(qtype, dev_packed, n_blocks, out, compute_dtype, out_dtype, ov->unpack);
    if (rc != GGQ_OK) return rc;
    if (staging_slot >= 0) {
        e = hipEventRecord(ov->consumed[staging_slot], ov->unpack);
        ov->consumed_valid[staging_slot] = e == hipSuccess;
    }
    if (e == hipSuccess) e = hipEventRecord(ov->done[slot], ov->unpack);
    return e == hipSuccess ? GGQ_OK : ggq::hip_fail(e);
}

int ggq_overlap_wait(ggq_overlap* ov, int slot, void* main_stream)
{
    if (!ov || slot < 0 || slot >= ov->n_slots) return GGQ_ERR_ARG;
    if (capturing(main_stream)) return GGQ_ERR_ARG;      // `done` was recorded outside the capture
    const hipError_t e = hipStreamWaitEvent(static_cast<hipStream_t>(main_stream), ov->done[slot], 0);
    return e == hipSuccess ? GGQ_OK : ggq::hip_fail(e);
}

void ggq_overlap_destroy(ggq_overlap* ov)
{
    if (!ov) return;
    for (hipStream_t s : {ov->copy, ov->copy2, ov->unpack})
        if (s) {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    for (int i = 0; i < MAX_SLOTS; i++)
        for (hipEvent_t ev : {ov->main_mark[i], ov->done[i], ov->copied[i], ov->consumed[i]})
            if (ev) (void)hipEventDestroy(ev);
    if (ov->half) (void)hipEventDestroy(ov->half);
    delete ov;
}

}  // extern "C"
