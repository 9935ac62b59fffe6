#pragma once
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include "gmt_common.h"

namespace gmt_bf16 {

// Row groups.  The rows of a batch are independent recurrences, so a layer call may run as NG independent chains of
// step launches, each over B / NG rows, on streams of their own: the workgroups of ONE launch move in lock-step (all in
// the contraction, then all in the HBM-heavy epilogue), two chains drift apart and one's epilogue runs under the other's
// MFMAs; a chain's launch boundary is covered by the other chain as well.  The side streams are the library's own (one
// set per device, created on first use), forked from and joined to the caller's stream with events, so the caller sees
// an ordinary in-order call.  The fork / join events are per device too: layer calls on ONE device are expected from one
// host thread at a time (as autograd issues them); MAX_GROUPS stays at 2 for the reason given in include/gmt_hip.h.
constexpr int MAX_GROUPS = 2, MAX_DEVICES = 64;
struct side_streams { hipStream_t s[MAX_GROUPS - 1]; hipEvent_t fork, join[MAX_GROUPS - 1]; int n; };

// the library's side streams of the current device, at least `want` of them (created on first use, kept until the process ends;
// default priority, non-blocking: a process has few hardware queues - GPU_MAX_HW_QUEUES - and every stream ever created keeps
// a share of one, so no more are made than a call uses)
inline side_streams* side_streams_of_device(int want)
{
    static side_streams table[MAX_DEVICES] = {};
    static std::mutex mu;
    int dev = 0;
    if (want > MAX_GROUPS - 1 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    side_streams& e = table[dev];
    if (!e.fork && hipEventCreateWithFlags(&e.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    while (e.n < want) {
        if (hipStreamCreateWithFlags(&e.s[e.n], hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&e.join[e.n], hipEventDisableTiming) != hipSuccess) return nullptr;
        ++e.n;
    }
    return &e;
}

// the wanted number of row groups: GMT_BF16_ROW_GROUPS (default 2), or what gmt_bf16_set_row_groups last set
inline std::atomic<int>& row_groups_setting()
{
    static std::atomic<int> want([] {
        const char* v = getenv("GMT_BF16_ROW_GROUPS");
        const int n = v ? atoi(v) : 2;
        return n < 1 ? 1 : (n > MAX_GROUPS ? MAX_GROUPS : n);
    }());
    return want;
}

// number of row groups for a layer call over B rows: every group keeps its `tile_rows`-row tiles in fours (XCD patches)
inline int row_groups(int B, int tile_rows)
{
    int ng = row_groups_setting().load();
    while (ng > 1 && (B % (ng * 4 * tile_rows) != 0)) --ng;
    return ng;
}

// runs launch(group, stream) for every step of every group: group 0 on the caller's stream, the others on side streams
template <class F>
int run_row_groups(int ng, int L, hipStream_t stream, F launch)
{
    side_streams* ss = ng > 1 ? side_streams_of_device(ng - 1) : nullptr;
    if (ng > 1 && !ss) return GMT_ERR_HIP;
    int rc = GMT_OK, forked = 0;
    if (ng > 1) {
        if (hipEventRecord(ss->fork, stream) != hipSuccess) return GMT_ERR_HIP;      // nothing forked yet
        for (int g = 1; g < ng && rc == GMT_OK; ++g) {
            if (hipStreamWaitEvent(ss->s[g - 1], ss->fork, 0) != hipSuccess) rc = GMT_ERR_HIP;
            else forked = g;
        }
    }
    for (int step = 0; step < L && rc == GMT_OK; ++step) {
        for (int g = 0; g < ng; ++g) launch(g, step, g == 0 ? stream : ss->s[g - 1]);
        if (hipGetLastError() != hipSuccess) rc = GMT_ERR_HIP;                       // stop issuing, still join below
    }
    // join whatever was forked, also on failure: the caller's stream must stay ordered after side-stream work on
    // buffers the caller will free or reuse as soon as this call returns
    for (int g = 1; g <= forked; ++g) {
        if (hipEventRecord(ss->join[g - 1], ss->s[g - 1]) != hipSuccess ||
            hipStreamWaitEvent(stream, ss->join[g - 1], 0) != hipSuccess) {
            rc = GMT_ERR_HIP;
            (void)hipStreamSynchronize(ss->s[g - 1]);                                    // last resort: order by waiting here
        }
    }
    return rc;
}

}  // namespace gmt_bf16
