// fake_rccl.cpp - TEST-ONLY stand-in for librccl.so: the eight nccl* entry points rtowComm* / rtowGatherRowsDevice bind
// (raytracing-in-one-weekend_amd/csrc/rtow_api.hip: RcclApi), carried over files in /dev/shm instead of xGMI.
//
// Why it exists: the product's gather (pack rows -> ncclSend | ncclGroupStart + ncclRecv x (G - 1) + ncclGroupEnd -> scatter rows) is the one
// piece of the path that needs more than one rank, RCCL refuses two ranks on one device ("duplicate GPU"), and the boxes the tests run on
// have ONE GPU.  Pointed at this library with rtowCommSetLibraryPath the very same product code runs with 2, 3 or 8 processes sharing that
// GPU.  The product never loads it on its own; nothing here is a model of RCCL's performance.
//
// Semantics kept from NCCL, because the product's correctness depends on them:
//   * ncclCommInitRank is collective: it returns once all `nranks` ranks of the id have joined (or fails after a timeout);
//   * ncclSend / ncclRecv are ordered per (source, destination) pair; a receive matches the oldest unmatched send of that pair and must
//     ask for exactly the count that was sent (else ncclInvalidArgument);
//   * operations posted between ncclGroupStart and ncclGroupEnd do NOTHING until the group is closed - a caller that forgets ncclGroupEnd
//     (e.g. by returning early on an error) never receives its data, and fakeRcclGroupDepth() shows the open group;
//   * stream order: an operation sees everything enqueued on its stream before it, and what is enqueued after it sees its result (here by
//     synchronising the stream and copying synchronously - slow, correct).
// Failure injection for the product's error paths: environment FAKE_RCCL_FAIL_RECV=<k> makes the k-th ncclRecv of the process (1-based)
// return ncclInternalError without posting anything.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define FAKE_API extern "C" __attribute__((visibility("default")))

namespace {

enum { kSuccess = 0, kUnhandledCudaError = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5, kRemoteError = 6 };
constexpr int kFloat32 = 7;          // ncclFloat32
constexpr double kTimeoutSeconds = 120.0;

struct UniqueId { char internal[128]; };

struct Op { bool send; void* buffer; size_t bytes; int peer; hipStream_t stream; };

struct Comm {
    std::string base;                 // /dev/shm/fake_rccl_<id>
    int rank = 0, nranks = 1;
    std::vector<unsigned> sendSeq, recvSeq;   // next message number per peer
    std::vector<Op> pending;          // posted inside an open group
};

thread_local int tGroupDepth = 0;
thread_local std::vector<std::pair<Comm*, Op>> tGroupOps;
std::atomic<int> gRecvCalls{0};
std::atomic<int> gOpenGroups{0};

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

bool exists(const std::string& path) { struct stat st; return stat(path.c_str(), &st) == 0; }

bool waitFor(const std::string& path)
{
    const double t0 = now();
    while (!exists(path)) {
        if (now() - t0 > kTimeoutSeconds) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    return true;
}

std::string messagePath(const Comm* c, int src, int dst, unsigned seq)
{
    char buf[64];
    snprintf(buf, sizeof(buf), ".m.%d.%d.%u", src, dst, seq);
    return c->base + buf;
}

int doSend(Comm* c, const Op& op)
{
    if (hipStreamSynchronize(op.stream) != hipSuccess) return kUnhandledCudaError;        // everything enqueued before the send is done
    std::vector<char> host(op.bytes);
    if (op.bytes && hipMemcpy(host.data(), op.buffer, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return kUnhandledCudaError;
    const std::string path = messagePath(c, c->rank, op.peer, c->sendSeq[(size_t)op.peer]++);
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return kSystemError;
    const uint64_t n = op.bytes;
    bool ok = fwrite(&n, sizeof(n), 1, f) == 1 && (op.bytes == 0 || fwrite(host.data(), 1, op.bytes, f) == op.bytes);
    ok = fclose(f) == 0 && ok;
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) return kSystemError;                 // rename: the receiver never sees half a message
    return kSuccess;
}

int doRecv(Comm* c, const Op& op)
{
    if (hipStreamSynchronize(op.stream) != hipSuccess) return kUnhandledCudaError;        // earlier users of the destination are done
    const std::string path = messagePath(c, op.peer, c->rank, c->recvSeq[(size_t)op.peer]++);
    if (!waitFor(path)) return kRemoteError;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return kSystemError;
    uint64_t n = 0;
    bool ok = fread(&n, sizeof(n), 1, f) == 1;
    if (ok && n != op.bytes) { fclose(f); unlink(path.c_str()); return kInvalidArgument; }   // count mismatch between the two ends
    std::vector<char> host(op.bytes);
    ok = ok && (op.bytes == 0 || fread(host.data(), 1, op.bytes, f) == op.bytes);
    fclose(f);
    unlink(path.c_str());
    if (!ok) return kSystemError;
    if (op.bytes && hipMemcpy(op.buffer, host.data(), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return kUnhandledCudaError;
    return kSuccess;
}

int post(Comm* c, const Op& op)
{
    if (tGroupDepth > 0) { tGroupOps.emplace_back(c, op); return kSuccess; }
    return op.send ? doSend(c, op) : doRecv(c, op);
}

} // namespace

FAKE_API int ncclGetUniqueId(UniqueId* out)
{
    if (!out) return kInvalidArgument;
    memset(out->internal, 0, sizeof(out->internal));
    unsigned long long r = 0;
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f || fread(&r, sizeof(r), 1, f) != 1) r = (unsigned long long)getpid() * 2654435761ull ^ (unsigned long long)(now() * 1e6);
    if (f) fclose(f);
    snprintf(out->internal, sizeof(out->internal), "fake_rccl_%d_%016llx", (int)getpid(), r);
    return kSuccess;
}

FAKE_API int ncclCommInitRank(void** comm, int nranks, UniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return kInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    if (strncmp(id.internal, "fake_rccl_", 10) != 0) return kInvalidArgument;
    Comm* c = new Comm();
    const char* dir = getenv("FAKE_RCCL_DIR");
    c->base = std::string(dir ? dir : "/dev/shm") + "/" + id.internal;
    c->rank = rank;
    c->nranks = nranks;
    c->sendSeq.assign((size_t)nranks, 0u);
    c->recvSeq.assign((size_t)nranks, 0u);
    // collective: every rank announces itself, then waits for all the others
    const std::string mine = c->base + ".join." + std::to_string(rank);
    FILE* f = fopen(mine.c_str(), "wb");
    if (!f) { delete c; return kSystemError; }
    fclose(f);
    for (int r = 0; r < nranks; r++)
        if (!waitFor(c->base + ".join." + std::to_string(r))) { unlink(mine.c_str()); delete c; return kRemoteError; }
    *comm = c;
    return kSuccess;
}

FAKE_API int ncclCommDestroy(void* comm)
{
    Comm* c = (Comm*)comm;
    if (!c) return kInvalidArgument;
    // leave: once every rank has left, the last one removes the join files (a rank that left early must stay visible to late joiners' waits)
    const std::string left = c->base + ".left." + std::to_string(c->rank);
    if (FILE* f = fopen(left.c_str(), "wb")) fclose(f);
    bool all = true;
    for (int r = 0; r < c->nranks; r++) all = all && exists(c->base + ".left." + std::to_string(r));
    if (all)
        for (int r = 0; r < c->nranks; r++) { unlink((c->base + ".join." + std::to_string(r)).c_str()); unlink((c->base + ".left." + std::to_string(r)).c_str()); }
    delete c;
    return kSuccess;
}

FAKE_API int ncclGroupStart()
{
    if (tGroupDepth++ == 0) gOpenGroups++;
    return kSuccess;
}

FAKE_API int ncclGroupEnd()
{
    if (tGroupDepth <= 0) return kInvalidUsage;
    if (--tGroupDepth > 0) return kSuccess;
    gOpenGroups--;
    // the group's operations run now: sends first (they never wait for a peer), then the receives
    int rc = kSuccess;
    std::vector<std::pair<Comm*, Op>> ops;
    ops.swap(tGroupOps);
    for (auto& o : ops) if (o.second.send && rc == kSuccess) rc = doSend(o.first, o.second);
    for (auto& o : ops) if (!o.second.send && rc == kSuccess) rc = doRecv(o.first, o.second);
    return rc;
}

FAKE_API int ncclSend(const void* buffer, size_t count, int datatype, int peer, void* comm, hipStream_t stream)
{
    Comm* c = (Comm*)comm;
    if (!c || datatype != kFloat32 || peer < 0 || peer >= c->nranks || peer == c->rank || (count && !buffer)) return kInvalidArgument;
    return post(c, Op{true, const_cast<void*>(buffer), count * 4u, peer, stream});
}

FAKE_API int ncclRecv(void* buffer, size_t count, int datatype, int peer, void* comm, hipStream_t stream)
{
    Comm* c = (Comm*)comm;
    if (!c || datatype != kFloat32 || peer < 0 || peer >= c->nranks || peer == c->rank || (count && !buffer)) return kInvalidArgument;
    const int call = ++gRecvCalls;
    if (const char* fail = getenv("FAKE_RCCL_FAIL_RECV")) if (atoi(fail) == call) return kInternalError;
    return post(c, Op{false, buffer, count * 4u, peer, stream});
}

FAKE_API const char* ncclGetErrorString(int result)
{
    switch (result) {
        case kSuccess: return "no error";
        case kUnhandledCudaError: return "unhandled hip error (fake transport)";
        case kSystemError: return "system error (fake transport: /dev/shm file)";
        case kInternalError: return "internal error (fake transport: injected)";
        case kInvalidArgument: return "invalid argument";
        case kInvalidUsage: return "invalid usage";
        case kRemoteError: return "remote error (fake transport: a peer did not show up in time)";
    }
    return "unknown result";
}

// test hooks (not part of NCCL)
FAKE_API int fakeRcclGroupDepth() { return tGroupDepth; }          // of the calling thread
FAKE_API int fakeRcclOpenGroups() { return gOpenGroups.load(); }   // of the process
FAKE_API int fakeRcclRecvCalls() { return gRecvCalls.load(); }
