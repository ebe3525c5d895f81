// Launch tape: record the kernel launches (and stream / event edges) of a training step once, while they run, and re-issue
// them from C afterwards -- one call per segment instead of one Python + ctypes round trip per launch.
//
// What a tape holds: closures, in issue order.  A closure is one hipLaunchKernelGGL with its by-value arguments frozen
// (VSOM_LAUNCH, common.h), one event record or one stream wait (vsom_event_record / vsom_stream_wait_event: library-owned
// events, so that the edges between the step's HIP streams are part of the tape), or one RCCL all-reduce (comm.hip).
// What it cannot hold: anything whose arguments change from step to step.  Those few calls (the neighbourhood kernel with
// the temperature, the loss combination, AdamW with lr / step) stay with the host, which cuts the tape into segments around
// them (vsom_tape_cut / vsom_tape_pause) and replays segment, call, segment, ...  Inputs are staged into fixed buffers by the host.
// A replayed segment issues exactly the launches the recorded step issued: results are bit-identical to the host-driven path.
#include "common.h"

#include <memory>
#include <mutex>
#include <vector>

namespace vsom {

struct TapeRec {
    std::vector<std::vector<std::function<void()>>> segments;
    bool paused = false;
};

thread_local TapeRec* g_tape_rec = nullptr;          // what VSOM_LAUNCH appends to (null: not recording, or paused)
static thread_local TapeRec* g_tape_cur = nullptr;   // the tape being recorded on this thread (also while paused)

void tape_push(std::function<void()>&& op) { g_tape_rec->segments.back().push_back(std::move(op)); }

namespace {
std::mutex g_mu;
std::vector<std::unique_ptr<TapeRec>> g_tapes;       // id = index + 1; destroyed tapes leave a null slot
constexpr int MAX_EVENTS = 512;
hipEvent_t g_events[MAX_EVENTS] = {};

TapeRec* tape_of(int id) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (id >= 1 && id <= (int)g_tapes.size()) ? g_tapes[id - 1].get() : nullptr;
}
}  // namespace

}  // namespace vsom

using namespace vsom;

extern "C" {

int vsom_tape_begin(void) {
    VSOM_REQUIRE(g_tape_cur == nullptr, VSOM_EINVAL, "tape_begin: this thread is already recording");
    std::lock_guard<std::mutex> lk(g_mu);
    g_tapes.emplace_back(new TapeRec());
    g_tape_cur = g_tape_rec = g_tapes.back().get();
    g_tape_cur->segments.emplace_back();
    return (int)g_tapes.size();
}

int vsom_tape_cut(void) {
    VSOM_REQUIRE(g_tape_cur != nullptr, VSOM_EINVAL, "tape_cut: not recording");
    g_tape_cur->segments.emplace_back();
    return (int)g_tape_cur->segments.size() - 2;
}

int vsom_tape_pause(int paused) {
    VSOM_REQUIRE(g_tape_cur != nullptr, VSOM_EINVAL, "tape_pause: not recording");
    g_tape_cur->paused = paused != 0;
    g_tape_rec = paused ? nullptr : g_tape_cur;
    return VSOM_OK;
}

int vsom_tape_end(void) {
    VSOM_REQUIRE(g_tape_cur != nullptr, VSOM_EINVAL, "tape_end: not recording");
    const int n = (int)g_tape_cur->segments.size();
    g_tape_cur = g_tape_rec = nullptr;
    return n;
}

int vsom_tape_recording(void) { return g_tape_cur != nullptr ? (g_tape_rec ? 1 : 2) : 0; }

int vsom_tape_segment_ops(int tape, int segment) {
    TapeRec* t = tape_of(tape);
    VSOM_REQUIRE(t && segment >= 0 && segment < (int)t->segments.size(), VSOM_EINVAL, "tape_segment_ops: no such tape / segment");
    return (int)t->segments[segment].size();
}

int vsom_tape_replay(int tape, int segment) {
    TapeRec* t = tape_of(tape);
    VSOM_REQUIRE(t && t != g_tape_cur, VSOM_EINVAL, "tape_replay: no such tape (or it is still being recorded)");
    VSOM_REQUIRE(segment >= 0 && segment < (int)t->segments.size(), VSOM_EINVAL, "tape_replay: no segment %d", segment);
    for (const auto& op : t->segments[segment]) op();
    return hip_status(hipGetLastError(), "tape_replay");
}

int vsom_tape_destroy(int tape) {
    std::lock_guard<std::mutex> lk(g_mu);
    VSOM_REQUIRE(tape >= 1 && tape <= (int)g_tapes.size(), VSOM_EINVAL, "tape_destroy: no such tape");
    VSOM_REQUIRE(g_tapes[tape - 1].get() != g_tape_cur || g_tape_cur == nullptr, VSOM_EINVAL, "tape_destroy: tape is being recorded");
    g_tapes[tape - 1].reset();
    return VSOM_OK;
}

// ---- library-owned events: the edges between the step's streams, recordable on a tape
static int event_of(int ev, hipEvent_t* out) {
    VSOM_REQUIRE(ev >= 0 && ev < MAX_EVENTS, VSOM_EINVAL, "event id %d outside [0, %d)", ev, MAX_EVENTS);
    if (!g_events[ev]) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_events[ev]) {
            const int rc = hip_status(hipEventCreateWithFlags(&g_events[ev], hipEventDisableTiming), "hipEventCreateWithFlags");
            if (rc) return rc;
        }
    }
    *out = g_events[ev];
    return VSOM_OK;
}

int vsom_event_record(int ev, vsom_stream_t stream) {
    hipEvent_t e;
    int rc = event_of(ev, &e);
    if (rc) return rc;
    rc = hip_status(hipEventRecord(e, stream), "hipEventRecord");
    if (rc == VSOM_OK && g_tape_rec) tape_push([=]() { (void)hipEventRecord(e, stream); });
    return rc;
}

int vsom_stream_wait_event(vsom_stream_t stream, int ev) {
    hipEvent_t e;
    int rc = event_of(ev, &e);
    if (rc) return rc;
    rc = hip_status(hipStreamWaitEvent(stream, e, 0), "hipStreamWaitEvent");
    if (rc == VSOM_OK && g_tape_rec) tape_push([=]() { (void)hipStreamWaitEvent(stream, e, 0); });
    return rc;
}

}  // extern "C"
