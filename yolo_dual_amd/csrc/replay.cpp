// Launch-list replay (include/ydl.h, "launch-list replay"): a recorded training step re-issued from C.
// Host code only.  The dispatcher table (one case per stream-taking entry point) is generated from ydl.h.
#include "common.h"

#include <string.h>
#include <vector>

namespace {
union Slot { int64_t i; double d; };
enum { OP_CALL = 0, OP_RECORD = 1, OP_WAIT = 2 };
struct ReplayOp {
    int kind, fn, stream, nargs;
    ydl_conv_geom geom;
    ydl_bnred red;
    Slot a[32];
};
}  // namespace

#include "replay_table.inc"

struct ydl_replay {
    std::vector<ReplayOp> ops;
    std::vector<hipEvent_t> events;      // created lazily on the device current at the first run
    int n_events = 0;
    int device = -1;
};

extern "C" ydl_replay* ydl_replay_create(void) { return new ydl_replay(); }

extern "C" void ydl_replay_destroy(ydl_replay* r) {
    if (!r) return;
    for (hipEvent_t e : r->events)
        if (e) (void)hipEventDestroy(e);
    delete r;
}

extern "C" int ydl_replay_fn_count(void) { return YDL_REPLAY_NFN; }
extern "C" const char* ydl_replay_fn_name(int fn) { return (fn >= 0 && fn < YDL_REPLAY_NFN) ? kReplayFnName[fn] : ""; }

extern "C" int ydl_replay_add_call(ydl_replay* r, int fn, const int64_t* args, int nargs, int stream_slot) {
    YDL_CHECK(r != nullptr && args != nullptr, "null argument");
    YDL_CHECK(fn >= 0 && fn < YDL_REPLAY_NFN, "unknown entry point index");
    YDL_CHECK(nargs == kReplayFnArgs[fn], "argument count does not match the entry point's prototype");
    YDL_CHECK(nargs <= 32 && stream_slot >= 0, "bad argument count / stream slot");
    ReplayOp op;
    memset(&op, 0, sizeof(op));
    op.kind = OP_CALL; op.fn = fn; op.stream = stream_slot; op.nargs = nargs;
    for (int k = 0; k < nargs; ++k) op.a[k].i = args[k];
    const int gi = kReplayFnGeom[fn];
    if (gi >= 0) {
        const ydl_conv_geom* g = (const ydl_conv_geom*)(uintptr_t)args[gi];
        YDL_CHECK(g != nullptr, "null geometry");
        op.geom = *g;
    }
    const int ri = kReplayFnRed[fn];
    if (ri >= 0) {
        const ydl_bnred* b = (const ydl_bnred*)(uintptr_t)args[ri];
        YDL_CHECK(b != nullptr, "null reduce descriptor");
        op.red = *b;
    }
    r->ops.push_back(op);
    return 0;
}

static int add_event_op(ydl_replay* r, int kind, int event_id, int stream_slot) {
    YDL_CHECK(r != nullptr && event_id >= 0 && event_id < (1 << 20) && stream_slot >= 0, "bad event id / stream slot");
    ReplayOp op;
    memset(&op, 0, sizeof(op));
    op.kind = kind; op.fn = event_id; op.stream = stream_slot;
    r->ops.push_back(op);
    if (event_id + 1 > r->n_events) r->n_events = event_id + 1;
    return 0;
}
extern "C" int ydl_replay_add_event_record(ydl_replay* r, int event_id, int stream_slot) {
    return add_event_op(r, OP_RECORD, event_id, stream_slot);
}
extern "C" int ydl_replay_add_event_wait(ydl_replay* r, int event_id, int stream_slot) {
    return add_event_op(r, OP_WAIT, event_id, stream_slot);
}

extern "C" int ydl_replay_size(const ydl_replay* r) { return r ? (int)r->ops.size() : 0; }

extern "C" int ydl_replay_run(ydl_replay* r, int first, int last, void* const* h_streams, int nstreams) {
    YDL_CHECK(r != nullptr && h_streams != nullptr, "null argument");
    YDL_CHECK(first >= 0 && first <= last && last <= (int)r->ops.size(), "operation range out of bounds");
    if (first == last) return 0;
    int dev = 0;
    YDL_CHECK(hipGetDevice(&dev) == hipSuccess, "hipGetDevice failed");
    if (r->device < 0) r->device = dev;
    YDL_CHECK(r->device == dev, "a launch list replays on the device it was recorded on");
    if ((int)r->events.size() < r->n_events) {
        const size_t have = r->events.size();
        r->events.resize(r->n_events, nullptr);
        for (size_t e = have; e < r->events.size(); ++e)
            YDL_CHECK(hipEventCreateWithFlags(&r->events[e], hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
    }
    for (int k = first; k < last; ++k) {
        const ReplayOp& op = r->ops[k];
        YDL_CHECK(op.stream < nstreams, "operation names a stream slot beyond the table");
        hipStream_t st = (hipStream_t)h_streams[op.stream];
        if (op.kind == OP_CALL) {
            const int rc = replay_dispatch(op, (void*)st);
            if (rc != 0) {
                // keep the callee's message, say where it happened
                ydl_set_error(std::string("ydl_replay_run: operation ") + std::to_string(k) + " (" + kReplayFnName[op.fn] + ") failed: " +
                              ydl_last_error());
                return rc;
            }
        } else if (op.kind == OP_RECORD) {
            YDL_CHECK(hipEventRecord(r->events[op.fn], st) == hipSuccess, "hipEventRecord failed");
        } else {
            YDL_CHECK(hipStreamWaitEvent(st, r->events[op.fn], 0) == hipSuccess, "hipStreamWaitEvent failed");
        }
    }
    return 0;
}
