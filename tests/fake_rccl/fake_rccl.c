/* TEST INFRASTRUCTURE ONLY -- a stand-in for librccl.so.1 that lets the multi-rank code of the library's own communicator
 * (fcn8s_comm_* in fcn8s_tensorflow_amd/csrc/model.hip: enqueue under the mutex, the watchdog thread, ncclCommAbort on a timeout or
 * an asynchronous error, the polling fcn8s_comm_destroy) run with world > 1 on a box that has ONE GPU: every rank is a process of
 * its own on the same device and the "wire" is a POSIX shared-memory slab.  Nothing under fcn8s_tensorflow_amd/ links or loads this
 * file; tests/test_fake_rccl_gpu.py points the library at it through the FCN8S_RCCL_LIBRARY environment variable.
 *
 * What it keeps of RCCL's contract, because the code under test depends on it:
 *   - collectives are ASYNCHRONOUS and stream-ordered: ncclAllReduce / ncclBroadcast return at once; the exchange runs when the
 *     stream reaches it (device -> pinned host copy, a host function that meets the peers in the slab and adds the ranks' buffers
 *     in rank order -- every rank gets the same bits --, pinned host -> device copy) and BLOCKS THE STREAM until every peer arrived;
 *   - ncclCommAbort releases a blocked stream (the host function gives up) and frees the communicator without waiting for peers;
 *   - ncclCommGetAsyncError reports a failure that happened after the enqueue.
 * Fault injection (environment, read at ncclCommInitRank):
 *   FAKE_RCCL_STALL_RANK=r        rank r never arrives at any all-reduce: its peers (and itself) block until they are aborted
 *   FAKE_RCCL_ASYNC_ERROR_RANK=r  on rank r ncclCommGetAsyncError reports ncclRemoteError once an all-reduce has been enqueued,
 *                                 and that all-reduce never completes (as if the peer had died under it)
 *   FAKE_RCCL_VERSION=code        what ncclGetVersion reports (default 22707 = 2.27.7, the version of the image's RCCL)        */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <fcntl.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclRedOp_t;      /* ncclSum = 0 */
typedef int ncclDataType_t;   /* ncclFloat = 7, ncclDouble = 8 */

#define MAX_RANKS 8
#define SLOT_BYTES (8u << 20)                 /* per-rank window of the slab; larger buffers go through it in chunks */

typedef struct {
    _Atomic uint64_t joined;                  /* ranks that have mapped the slab */
    _Atomic uint64_t arrive[MAX_RANKS];       /* barrier tickets: rank r has passed arrive[r] barriers */
    unsigned char pad[4096 - 8 * (1 + MAX_RANKS)];
    unsigned char slot[MAX_RANKS][SLOT_BYTES];
} Slab;

typedef struct Pending { void* host; hipEvent_t done; struct Pending* next; } Pending;

struct ncclComm {
    Slab* slab; char name[128];
    int rank, world;
    _Atomic int aborted;
    int stall, async_error;
    _Atomic int enqueued_allreduces;
    _Atomic int live_ops;                     /* host functions not yet finished (they hold pointers into this struct) */
    Pending* pending;
};
typedef struct ncclComm* ncclComm_t;

typedef struct { ncclComm_t c; void* host; size_t count, esize; int is_double, bcast_root, never; } Op;

static void nap(void) { struct timespec t = {0, 200000}; nanosleep(&t, NULL); }

/* every rank passes barrier k together; 0 = passed, -1 = this communicator was aborted while waiting */
static int barrier(ncclComm_t c)
{
    const uint64_t mine = atomic_fetch_add(&c->slab->arrive[c->rank], 1) + 1;
    for (;;) {
        int all = 1;
        for (int r = 0; r < c->world; ++r) if (atomic_load(&c->slab->arrive[r]) < mine) { all = 0; break; }
        if (all) return 0;
        if (atomic_load(&c->aborted)) return -1;
        nap();
    }
}

static void exchange(void* user)
{
    Op* op = (Op*)user; ncclComm_t c = op->c;
    const size_t total = op->count * op->esize;
    if (op->never) {                          /* the injected faults: this rank never meets its peers */
        while (!atomic_load(&c->aborted)) nap();
        goto out;
    }
    for (size_t off = 0; off < total; off += SLOT_BYTES) {
        const size_t nb = total - off < SLOT_BYTES ? total - off : SLOT_BYTES;
        unsigned char* mine = (unsigned char*)op->host + off;
        if (op->bcast_root < 0 || op->bcast_root == c->rank) memcpy(c->slab->slot[c->rank], mine, nb);
        if (barrier(c)) goto out;
        if (op->bcast_root >= 0) { if (op->bcast_root != c->rank) memcpy(mine, c->slab->slot[op->bcast_root], nb); }
        else if (op->is_double) {
            double* o = (double*)mine; const size_t n = nb / 8;
            for (size_t i = 0; i < n; ++i) { double s = 0.0; for (int r = 0; r < c->world; ++r) s += ((const double*)c->slab->slot[r])[i]; o[i] = s; }
        } else {
            float* o = (float*)mine; const size_t n = nb / 4;
            for (size_t i = 0; i < n; ++i) { float s = 0.f; for (int r = 0; r < c->world; ++r) s += ((const float*)c->slab->slot[r])[i]; o[i] = s; }
        }
        if (barrier(c)) goto out;             /* nobody overwrites a window a peer is still reading */
    }
out:
    atomic_fetch_sub(&c->live_ops, 1);
    free(op);
}

static void reap(ncclComm_t c, int all)
{
    Pending** pp = &c->pending;
    while (*pp) {
        Pending* p = *pp;
        if (all || hipEventQuery(p->done) == hipSuccess) {
            if (all) hipEventSynchronize(p->done);
            hipEventDestroy(p->done); hipHostFree(p->host); *pp = p->next; free(p);
        } else pp = &p->next;
    }
    (void)hipGetLastError();
}

static ncclResult_t collective(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t c, hipStream_t s)
{
    if (!c || !recv || (dt != 7 && dt != 8)) return ncclInvalidArgument;
    if (atomic_load(&c->aborted)) return ncclInvalidUsage;
    if (count == 0) return ncclSuccess;
    reap(c, 0);
    const size_t esize = dt == 8 ? 8 : 4, total = count * esize;
    Op* op = (Op*)calloc(1, sizeof(Op)); Pending* p = (Pending*)calloc(1, sizeof(Pending));
    if (!op || !p) return ncclSystemError;
    if (hipHostMalloc(&op->host, total, hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
    op->c = c; op->count = count; op->esize = esize; op->is_double = dt == 8; op->bcast_root = root;
    if (root < 0) {
        const int k = atomic_fetch_add(&c->enqueued_allreduces, 1);
        op->never = c->stall || (c->async_error && k >= 0);
    }
    if (root < 0 || root == c->rank) if (hipMemcpyAsync(op->host, send, total, hipMemcpyDeviceToHost, s) != hipSuccess) return ncclUnhandledCudaError;
    atomic_fetch_add(&c->live_ops, 1);
    void* host = op->host;
    if (hipLaunchHostFunc(s, exchange, op) != hipSuccess) { atomic_fetch_sub(&c->live_ops, 1); return ncclUnhandledCudaError; }
    if (hipMemcpyAsync(recv, host, total, hipMemcpyHostToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
    p->host = host;
    if (hipEventCreateWithFlags(&p->done, hipEventDisableTiming) != hipSuccess || hipEventRecord(p->done, s) != hipSuccess) return ncclUnhandledCudaError;
    p->next = c->pending; c->pending = p;
    return ncclSuccess;
}

ncclResult_t ncclGetVersion(int* v) { const char* e = getenv("FAKE_RCCL_VERSION"); if (!v) return ncclInvalidArgument; *v = e ? atoi(e) : 22707; return ncclSuccess; }

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error (fake rccl)";
    case ncclUnhandledCudaError: return "unhandled cuda error (fake rccl)";
    case ncclSystemError: return "unhandled system error (fake rccl)";
    case ncclInternalError: return "internal error (fake rccl)";
    case ncclInvalidArgument: return "invalid argument (fake rccl)";
    case ncclInvalidUsage: return "invalid usage (fake rccl)";
    case ncclRemoteError: return "remote process exited or there was a network error (fake rccl)";
    default: return "unknown result code (fake rccl)";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (!id) return ncclInvalidArgument;
    struct timespec t; clock_gettime(CLOCK_REALTIME, &t);
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/fcn8s_fake_rccl_%d_%ld%09ld", (int)getpid(), (long)t.tv_sec, (long)t.tv_nsec);
    return ncclSuccess;
}

static int env_is_rank(const char* name, int rank) { const char* e = getenv(name); return e && *e && atoi(e) == rank; }

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank)
{
    if (!out || world < 1 || world > MAX_RANKS || rank < 0 || rank >= world || id.internal[0] != '/') return ncclInvalidArgument;
    ncclComm_t c = (ncclComm_t)calloc(1, sizeof *c);
    if (!c) return ncclSystemError;
    id.internal[sizeof id.internal - 1] = 0;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)sizeof(Slab)) != 0) { if (fd >= 0) close(fd); free(c); return ncclSystemError; }
    c->slab = (Slab*)mmap(NULL, sizeof(Slab), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);          /* (a fresh shm object reads as zeros) */
    close(fd);
    if (c->slab == MAP_FAILED) { free(c); return ncclSystemError; }
    c->rank = rank; c->world = world;
    c->stall = env_is_rank("FAKE_RCCL_STALL_RANK", rank);
    c->async_error = env_is_rank("FAKE_RCCL_ASYNC_ERROR_RANK", rank);
    atomic_fetch_add(&c->slab->joined, 1);
    for (int i = 0; atomic_load(&c->slab->joined) < (uint64_t)world; ++i) {            /* the bootstrap rendezvous: 60 s */
        if (i > 300000) { munmap(c->slab, sizeof(Slab)); free(c); return ncclSystemError; }
        nap();
    }
    if (rank == 0) shm_unlink(c->name);       /* everybody has it mapped: the name can go */
    *out = c;
    return ncclSuccess;
}

static void release(ncclComm_t c)
{
    while (atomic_load(&c->live_ops) > 0) nap();      /* host functions still hold the communicator */
    reap(c, 1);
    munmap(c->slab, sizeof(Slab));
    free(c);
}

ncclResult_t ncclCommDestroy(ncclComm_t c) { if (!c) return ncclInvalidArgument; release(c); return ncclSuccess; }

ncclResult_t ncclCommAbort(ncclComm_t c)
{
    if (!c) return ncclInvalidArgument;
    atomic_store(&c->aborted, 1);             /* blocked host functions give up: the streams run on (with whatever the buffers hold) */
    release(c);
    return ncclSuccess;
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t c, ncclResult_t* e)
{
    if (!c || !e) return ncclInvalidArgument;
    *e = (c->async_error && atomic_load(&c->enqueued_allreduces) > 0) ? ncclRemoteError : ncclSuccess;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c, hipStream_t s)
{
    if (op != 0) return ncclInvalidArgument;
    return collective(send, recv, count, dt, -1, c, s);
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t c, hipStream_t s)
{
    if (!c || root < 0 || root >= c->world) return ncclInvalidArgument;
    return collective(send, recv, count, dt, root, c, s);
}
