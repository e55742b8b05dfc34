// fake_rccl.cpp -- TEST-ONLY stand-in for librccl, for ranks that SHARE one GPU (real RCCL refuses two ranks on one device).
// Exports exactly the nccl* symbols libpdlp_hip.so resolves with dlsym (csrc/pdlp_hip.hip, rccl_load), so that the
// library's own exchange path -- pdlp_comm_init, pdlp_iterate on a sharded handle -- can run with 2 and 3 ranks on the
// one-GPU test box.  Transport: a POSIX shared-memory segment named by the "unique id"; a collective drains the stream it
// is given, stages this rank's block through the segment and meets the other ranks at a barrier (with a timeout: a missing
// peer is an error, never a hang).  The all-reduce adds the ranks' contributions in rank order on every rank, so all ranks
// get the same bits.  Not a product file: nothing under torchpdlp_amd/ refers to it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

namespace {

struct Header {
    std::atomic<int> count;
    std::atomic<int> gen;
    std::atomic<int> joined;
};

struct FakeComm {
    int rank, nranks;
    size_t cap;          // bytes of the data area
    size_t map_bytes;
    Header* hdr;
    char* data;
};

constexpr size_t HDR_BYTES = 4096;
double timeout_s()
{
    static const double t = std::getenv("PDLP_FAKE_RCCL_TIMEOUT") ? std::atof(std::getenv("PDLP_FAKE_RCCL_TIMEOUT")) : 60.0;
    return t;
}
#define TIMEOUT_S (timeout_s())

size_t data_capacity()
{
    const char* e = std::getenv("PDLP_FAKE_RCCL_BYTES");
    return e ? (size_t)std::strtoull(e, nullptr, 10) : ((size_t)96 << 20);
}

bool barrier(FakeComm* c)
{
    Header* h = c->hdr;
    const int g = h->gen.load(std::memory_order_acquire);
    if (h->count.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {
        h->count.store(0, std::memory_order_relaxed);
        h->gen.fetch_add(1, std::memory_order_release);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->gen.load(std::memory_order_acquire) == g) {
        std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > TIMEOUT_S) return false;
    }
    return true;
}

size_t dtype_bytes(ncclDataType_t t)
{
    switch (t) {
        case ncclFloat32: case ncclInt32: case ncclUint32: return 4;
        case ncclFloat64: case ncclInt64: case ncclUint64: return 8;
        case ncclInt8: case ncclUint8: return 1;
        default: return 0;
    }
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    static std::atomic<int> serial{0};
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "/pdlp_fake_rccl_%d_%d_%lld", (int)getpid(), serial.fetch_add(1),
                  (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    char name[128];
    std::memcpy(name, id.internal, sizeof(name));
    name[sizeof(name) - 1] = 0;
    const size_t cap = data_capacity(), total = HDR_BYTES + cap;
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, (off_t)total) != 0) { close(fd); return ncclSystemError; }      // (a fresh segment is zero filled)
    void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    FakeComm* c = new FakeComm{rank, nranks, cap, total, (Header*)p, (char*)p + HDR_BYTES};
    c->hdr->joined.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->hdr->joined.load() < nranks) {
        std::this_thread::sleep_for(std::chrono::microseconds(100));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > TIMEOUT_S) {
            munmap(p, total);
            delete c;
            return ncclSystemError;
        }
    }
    if (!barrier(c)) { munmap(p, total); delete c; return ncclSystemError; }
    if (rank == 0) shm_unlink(name);              // every rank has it mapped: the name can go
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    FakeComm* c = (FakeComm*)comm;
    if (!c) return ncclSuccess;
    munmap((void*)c->hdr, c->map_bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream)
{
    FakeComm* c = (FakeComm*)comm;
    const size_t bytes = sendcount * dtype_bytes(datatype);
    if (!c || dtype_bytes(datatype) == 0 || bytes * c->nranks > c->cap) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes && hipMemcpy(c->data + (size_t)c->rank * bytes, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    if (bytes && hipMemcpy(recvbuff, c->data, bytes * c->nranks, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;      // nobody stages the next collective before everybody has read this one
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
    FakeComm* c = (FakeComm*)comm;
    if (!c || datatype != ncclFloat64 || op != ncclSum || count * 8 * c->nranks > c->cap || count > 4096) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    double* slots = (double*)c->data;
    if (hipMemcpy(slots + (size_t)c->rank * count, sendbuff, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    double out[4096];
    for (size_t i = 0; i < count; ++i) {
        double s = slots[i];
        for (int r = 1; r < c->nranks; ++r) s += slots[(size_t)r * count + i];      // rank order, on every rank
        out[i] = s;
    }
    if (hipMemcpy(recvbuff, out, count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    return ncclSuccess;
}

// grouped calls execute one by one, in call order (the same on every rank): nothing to collect
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream)
{
    FakeComm* c = (FakeComm*)comm;
    const size_t bytes = count * dtype_bytes(datatype);
    if (!c || dtype_bytes(datatype) == 0 || bytes > c->cap || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (c->rank == root && bytes && hipMemcpy(c->data, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    if (c->rank != root && bytes && hipMemcpy(recvbuff, c->data, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (c->rank == root && recvbuff != sendbuff && bytes &&
        hipMemcpy(recvbuff, sendbuff, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "fake rccl: success";
        case ncclInvalidArgument: return "fake rccl: invalid argument (or the staging segment is too small: PDLP_FAKE_RCCL_BYTES)";
        case ncclSystemError: return "fake rccl: a peer did not arrive (timeout) or shared memory failed";
        default: return "fake rccl: HIP error";
    }
}

}  // extern "C"
