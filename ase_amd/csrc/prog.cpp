// Launch programs (see prog.h) + stream fork / join + recordable memset / copy.
#include <vector>
#include "prog.h"
#include "../../include/ase_hip.h"

void ase_set_error(const char* fmt, ...);

struct AseProgram {
    struct Entry {
        int kind;                // 0 launch closure, 1 record event, 2 wait event, 3 host callback
        hipStream_t stream;
        int ev;
        std::function<void(hipStream_t)> fn;
    };
    std::vector<Entry> entries;
    std::vector<hipEvent_t> events;
};

static thread_local AseProgram* g_recording = nullptr;

AseProgram* ase_prog_recording() { return g_recording; }

void ase_prog_push(AseProgram* pg, hipStream_t stream, std::function<void(hipStream_t)>&& fn) {
    pg->entries.push_back({0, stream, -1, std::move(fn)});
}

namespace {
constexpr int kPool = 256;       // events of the eager (non-recorded) fork / join points, reused round-robin
hipEvent_t g_pool[kPool];
bool g_pool_ready = false;
int g_pool_next = 0;

int pool_init() {
    if (g_pool_ready) return ASE_OK;
    for (int i = 0; i < kPool; ++i) {
        if (hipEventCreateWithFlags(&g_pool[i], hipEventDisableTiming) != hipSuccess) {
            ase_set_error("ase_hip_mark: hipEventCreate failed");
            return ASE_ELAUNCH;
        }
    }
    g_pool_ready = true;
    return ASE_OK;
}
}  // namespace

extern "C" int ase_hip_prog_create(void** prog) {
    if (!prog) { ase_set_error("prog_create: null argument"); return ASE_EINVAL; }
    *prog = new AseProgram();
    return ASE_OK;
}

extern "C" int ase_hip_prog_destroy(void* prog) {
    AseProgram* pg = static_cast<AseProgram*>(prog);
    if (!pg) return ASE_OK;
    if (g_recording == pg) g_recording = nullptr;
    for (hipEvent_t e : pg->events) (void)hipEventDestroy(e);
    delete pg;
    return ASE_OK;
}

extern "C" int ase_hip_prog_begin(void* prog) {
    AseProgram* pg = static_cast<AseProgram*>(prog);
    if (!pg || g_recording) { ase_set_error("prog_begin: null program or a recording is already open on this thread"); return ASE_EINVAL; }
    pg->entries.clear();
    g_recording = pg;
    return ASE_OK;
}

extern "C" int ase_hip_prog_end(void* prog) {
    if (!prog || g_recording != prog) { ase_set_error("prog_end: this program is not being recorded"); return ASE_EINVAL; }
    g_recording = nullptr;
    return ASE_OK;
}

extern "C" int ase_hip_prog_size(void* prog) { return prog ? (int)static_cast<AseProgram*>(prog)->entries.size() : -1; }

extern "C" int ase_hip_prog_launch(void* prog) {
    AseProgram* pg = static_cast<AseProgram*>(prog);
    if (!pg || g_recording == pg) { ase_set_error("prog_launch: null program or still recording"); return ASE_EINVAL; }
    for (auto& e : pg->entries) {
        if (e.kind == 0 || e.kind == 3) e.fn(e.stream);
        else if (e.kind == 1) (void)hipEventRecord(pg->events[e.ev], e.stream);
        else (void)hipStreamWaitEvent(e.stream, pg->events[e.ev], 0);
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) { ase_set_error("prog_launch: %s", hipGetErrorString(err)); return ASE_ELAUNCH; }
    return ASE_OK;
}

extern "C" int ase_hip_mark(void* stream, int* id) {
    if (!id) { ase_set_error("mark: null id"); return ASE_EINVAL; }
    if (AseProgram* pg = g_recording) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ase_set_error("mark: hipEventCreate failed"); return ASE_ELAUNCH; }
        pg->events.push_back(ev);
        *id = (int)pg->events.size() - 1;
        pg->entries.push_back({1, (hipStream_t)stream, *id, nullptr});
        return ASE_OK;
    }
    if (int rc = pool_init()) return rc;
    *id = g_pool_next;
    g_pool_next = (g_pool_next + 1) % kPool;
    if (hipEventRecord(g_pool[*id], (hipStream_t)stream) != hipSuccess) { ase_set_error("mark: hipEventRecord failed"); return ASE_ELAUNCH; }
    return ASE_OK;
}

extern "C" int ase_hip_wait(void* stream, int id) {
    if (AseProgram* pg = g_recording) {
        if (id < 0 || id >= (int)pg->events.size()) { ase_set_error("wait: event %d is not part of the program being recorded", id); return ASE_EINVAL; }
        pg->entries.push_back({2, (hipStream_t)stream, id, nullptr});
        return ASE_OK;
    }
    if (id < 0 || id >= kPool || !g_pool_ready) { ase_set_error("wait: bad event id %d", id); return ASE_EINVAL; }
    if (hipStreamWaitEvent((hipStream_t)stream, g_pool[id], 0) != hipSuccess) { ase_set_error("wait: hipStreamWaitEvent failed"); return ASE_ELAUNCH; }
    return ASE_OK;
}

extern "C" int ase_hip_prog_host(void (*fn)(void*), void* arg) {
    if (!fn) { ase_set_error("prog_host: null callback"); return ASE_EINVAL; }
    if (AseProgram* pg = g_recording) {
        pg->entries.push_back({3, nullptr, -1, [=](hipStream_t) { fn(arg); }});
        return ASE_OK;
    }
    fn(arg);
    return ASE_OK;
}

extern "C" int ase_hip_memset(void* dst, int value, int64_t bytes, void* stream) {
    if (!dst || bytes <= 0) { ase_set_error("memset: null/empty operand"); return ASE_EINVAL; }
    if (AseProgram* pg = g_recording) {
        ase_prog_push(pg, (hipStream_t)stream, [=](hipStream_t s) { (void)hipMemsetAsync(dst, value, (size_t)bytes, s); });
        return ASE_OK;
    }
    if (hipMemsetAsync(dst, value, (size_t)bytes, (hipStream_t)stream) != hipSuccess) { ase_set_error("memset failed"); return ASE_ELAUNCH; }
    return ASE_OK;
}

extern "C" int ase_hip_memcpy(void* dst, const void* src, int64_t bytes, void* stream) {
    if (!dst || !src || bytes <= 0) { ase_set_error("memcpy: null/empty operand"); return ASE_EINVAL; }
    if (AseProgram* pg = g_recording) {
        ase_prog_push(pg, (hipStream_t)stream, [=](hipStream_t s) { (void)hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, s); });
        return ASE_OK;
    }
    if (hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) { ase_set_error("memcpy failed"); return ASE_ELAUNCH; }
    return ASE_OK;
}
