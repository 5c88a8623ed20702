// Host-side owner of the global-memory work area of the QP kernels no CU's LDS holds (clik_qp.hip: the "GLOBAL" variants,
// more than 16 rows or more than eight states).  One owner per controller handle (clik_qp): the area belongs to the
// skill it serves, lives as long as the handle and is released by clik_qp_destroy.
//
//  * sized for the blocks the batch needs, at most the device's resident blocks of the kernel (the kernels walk larger
//    batches with a block stride), grown geometrically;
//  * GROWING RETIRES the smaller area instead of freeing it (freed with the handle): a hipGraph captured while the area
//    was small has its address baked in and keeps writing into live memory; the retired areas sum to less than the
//    current one;
//  * growing needs hipMalloc, which cannot be captured: the first tick of a batch size must run outside a capture (the
//    error says so);
//  * ticks of one handle on DIFFERENT streams share the area: a launch on another stream than the handle's previous
//    one first waits (host-side) for that stream - eager launches are safe; two graphs of one handle replayed
//    concurrently on two streams are not supported (use one controller per concurrent stream);
//  * hipStreamPerThread is refused for these kernels: one handle value stands for a different stream in every thread,
//    so the ordering above cannot be kept.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>

namespace clik {

struct GwsOwner {
    std::mutex m;
    double* ptr = nullptr;
    size_t bytes = 0;
    std::vector<double*> retired;
    size_t retired_bytes = 0;
    hipStream_t last_stream = nullptr;
    bool used = false;

    // area of at least `need` bytes (never more than `cap`, the device's residency) for a launch on `stream`
    hipError_t acquire(hipStream_t stream, size_t need, size_t cap, double** out)
    {
        if (stream == hipStreamPerThread) return hipErrorInvalidResourceHandle;
        std::lock_guard<std::mutex> lock(m);
        hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(stream, &cap_status) == hipSuccess && cap_status != hipStreamCaptureStatusNone;
        if (used && last_stream != stream && !capturing) {
            // (the handle's previous launch may still be using the area on the other stream)
            hipError_t e = hipStreamSynchronize(last_stream);
            if (e != hipSuccess) return e;
        }
        if (bytes < need) {
            if (capturing) return hipErrorStreamCaptureUnsupported;      // (run one tick of this batch size before capturing)
            size_t want = bytes ? 2 * bytes : need;
            if (want < need) want = need;
            if (want > cap && cap >= need) want = cap;
            double* fresh = nullptr;
            hipError_t e = hipMalloc((void**)&fresh, want);
            if (e != hipSuccess) return e;
            if (ptr) {
                retired.push_back(ptr);
                retired_bytes += bytes;
            }
            ptr = fresh;
            bytes = want;
        }
        if (!capturing) {
            last_stream = stream;
            used = true;
        }
        *out = ptr;
        return hipSuccess;
    }
    size_t footprint() const { return bytes + retired_bytes; }
    void release()
    {
        std::lock_guard<std::mutex> lock(m);
        if (ptr) (void)hipFree(ptr);
        for (double* p : retired) (void)hipFree(p);
        retired.clear();
        ptr = nullptr;
        bytes = retired_bytes = 0;
        used = false;
    }
};

}  // namespace clik
