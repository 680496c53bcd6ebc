// bsmr_sharded_*: one SDDMM over several GPUs of one node, driven from ONE process (include/bsmr_hip.h).
//
// Row panels are independent: the caller cuts the rows of S into contiguous ranges, builds the RPHM of
// every range (host pipeline), and hands the ranges' RPHM arrays over.  Shard i lives on devices[i]: its plan, its rows
// of A, a replica of B, its part of P.  A step launches bsmr_sddmm on every device (asynchronous launches from the
// calling thread, one stream per shard), then ONE gather-v of the compact fp32 outputs to devices[0] - RCCL send / recv
// inside one group, created once per sharded object (ncclCommInitAll over the DISTINCT devices of the list); ranges are
// contiguous in S's row order, so the root's P is the concatenation of the shards' outputs and no permutation pass is
// needed.  A device may appear more than once in the list (several shards on one GPU: a rehearsal of the N-shard
// arithmetic on a one-GPU box, or more shards than GPUs): a shard on the root's device hands its part over with a
// device-to-device copy, RCCL only moves what crosses devices.
// Included at the end of bsmr_capi.hip (one translation unit).
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

// RCCL is bound at run time, when a sharded object over more than one device is created: libbsmr_hip.so carries no
// load-time dependency on librccl (a process that also loads PyTorch must end up with ONE copy of it - whichever
// librccl.so.1 is already in the process is the one dlopen returns).
namespace {
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) commInitAll = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr;
    decltype(&ncclGroupEnd) groupEnd = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGetErrorString) errorString = nullptr;
    bool ok() const { return commInitAll && commDestroy && groupStart && groupEnd && send && recv && errorString; }
};
RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.lib) break;
        }
        if (a.lib) {
            a.commInitAll = reinterpret_cast<decltype(a.commInitAll)>(dlsym(a.lib, "ncclCommInitAll"));
            a.commDestroy = reinterpret_cast<decltype(a.commDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
            a.groupStart = reinterpret_cast<decltype(a.groupStart)>(dlsym(a.lib, "ncclGroupStart"));
            a.groupEnd = reinterpret_cast<decltype(a.groupEnd)>(dlsym(a.lib, "ncclGroupEnd"));
            a.send = reinterpret_cast<decltype(a.send)>(dlsym(a.lib, "ncclSend"));
            a.recv = reinterpret_cast<decltype(a.recv)>(dlsym(a.lib, "ncclRecv"));
            a.errorString = reinterpret_cast<decltype(a.errorString)>(dlsym(a.lib, "ncclGetErrorString"));
        }
        return a;
    }();
    return api;
}
}  // namespace

struct bsmr_sharded {
    std::vector<int> devices;
    std::vector<bsmr_plan*> plans;
    std::vector<uint32_t> rowBegin;      // [n+1] global row of every shard's first row
    std::vector<uint64_t> entryBegin;    // [n+1] offset of every shard's entries in P
    uint32_t M = 0, N = 0;
    std::vector<int> uniqueDevices;      // the distinct devices, root first; rankOf[i] = index of devices[i] in it
    std::vector<int> rankOf;
    std::vector<ncclComm_t> comms;       // one per distinct device (single-process communicator clique), empty for one device
    std::vector<hipStream_t> streams;    // compute stream of every shard
    std::vector<hipStream_t> gatherStreams;   // the shard's part travels to the root on a stream of its own, behind an event
    std::vector<hipEvent_t> start, stop;
    std::vector<hipEvent_t> computed[2], gathered[2], gatherStart;   // per shard and P buffer: SDDMM done / part delivered
    // device buffers of the last K served
    uint32_t K = 0;
    std::vector<float*> A, B;
    // Two output buffers per shard, used by alternate steps: the gather of step i runs behind the SDDMM of step i + 1
    // (the Python driver's PipelinedSteps, python/shard.py).  P[b][0] holds the whole result of a step (root), P[b][i > 0] the
    // shard's part.
    std::vector<float*> P[2];
};

namespace {

void freeShardedBuffers(bsmr_sharded* s) {
    for (size_t i = 0; i < s->devices.size(); ++i) {
        if (hipSetDevice(s->devices[i]) != hipSuccess) (void)hipGetLastError();
        if (i < s->A.size() && s->A[i]) (void)hipFree(s->A[i]);
        if (i < s->B.size() && s->B[i]) (void)hipFree(s->B[i]);
        for (auto& buf : s->P)
            if (i < buf.size() && buf[i]) (void)hipFree(buf[i]);
    }
    s->A.clear();
    s->B.clear();
    s->P[0].clear();
    s->P[1].clear();
    s->K = 0;
}

#define BSMR_NCCL(call)                                                        \
    do {                                                                       \
        const ncclResult_t r_ = (call);                                        \
        if (r_ != ncclSuccess) {                                               \
            g_lastHipError = std::string(#call) + ": " + rccl().errorString(r_); \
            return BSMR_ERR_HIP;                                               \
        }                                                                      \
    } while (0)

// the calling thread's current device is the caller's business (a process that also hosts PyTorch relies on it)
struct DeviceRestore {
    int saved = -1;
    DeviceRestore() {
        if (hipGetDevice(&saved) != hipSuccess) {
            (void)hipGetLastError();
            saved = -1;
        }
    }
    ~DeviceRestore() {
        if (saved >= 0 && hipSetDevice(saved) != hipSuccess) (void)hipGetLastError();
    }
};

// the gather-v of one step (output buffers `b`): peers hand their part to the root, which receives each into its slot.  All of
// it runs on the gather streams, behind the event that says the shard's SDDMM of this step is done, so the compute streams
// are free for the next step at once.
int shardedGather(bsmr_sharded* s, int b) {
    const size_t n = s->devices.size();
    if (n == 1) return BSMR_OK;
    for (size_t i = 0; i < n; ++i) {
        BSMR_HIP(hipSetDevice(s->devices[i]));
        BSMR_HIP(hipStreamWaitEvent(s->gatherStreams[i], s->computed[b][i], 0));
        if (i == 0) {   // the root's receives also wait for nothing else: its own part was written in place
            continue;
        }
    }
    // shards on the root's device: a device-to-device copy
    for (size_t i = 1; i < n; ++i) {
        const uint64_t count = s->entryBegin[i + 1] - s->entryBegin[i];
        if (!count || s->rankOf[i] != 0) continue;
        BSMR_HIP(hipSetDevice(s->devices[i]));
        BSMR_HIP(hipMemcpyAsync(s->P[b][0] + s->entryBegin[i], s->P[b][i], count * 4, hipMemcpyDeviceToDevice, s->gatherStreams[i]));
    }
    int status = BSMR_OK;
    if (!s->comms.empty()) {
        // the rest crosses devices: one RCCL group; an error inside it still closes the group
        BSMR_NCCL(rccl().groupStart());
        for (size_t i = 1; i < n && status == BSMR_OK; ++i) {
            const uint64_t count = s->entryBegin[i + 1] - s->entryBegin[i];
            if (!count || s->rankOf[i] == 0) continue;
            ncclResult_t r = rccl().recv(s->P[b][0] + s->entryBegin[i], count, ncclFloat, s->rankOf[i], s->comms[0], s->gatherStreams[0]);
            if (r == ncclSuccess) r = rccl().send(s->P[b][i], count, ncclFloat, 0, s->comms[(size_t)s->rankOf[i]], s->gatherStreams[i]);
            if (r != ncclSuccess) {
                g_lastHipError = std::string("ncclSend / ncclRecv: ") + rccl().errorString(r);
                status = BSMR_ERR_HIP;
            }
        }
        const ncclResult_t e = rccl().groupEnd();
        if (e != ncclSuccess && status == BSMR_OK) {
            g_lastHipError = std::string("ncclGroupEnd: ") + rccl().errorString(e);
            status = BSMR_ERR_HIP;
        }
    }
    for (size_t i = 0; i < n && status == BSMR_OK; ++i) {   // the buffer is free again once its part has been delivered
        BSMR_HIP(hipSetDevice(s->devices[i]));
        BSMR_HIP(hipEventRecord(s->gathered[b][i], s->gatherStreams[i]));
    }
    return status;
}

}  // namespace

extern "C" {

int bsmr_sharded_destroy(bsmr_sharded* s) {
    if (!s) return BSMR_OK;
    DeviceRestore restore;
    freeShardedBuffers(s);
    for (size_t u = 0; u < s->comms.size(); ++u) {
        if (hipSetDevice(s->uniqueDevices[u]) != hipSuccess) (void)hipGetLastError();
        if (s->comms[u]) (void)rccl().commDestroy(s->comms[u]);
    }
    for (size_t i = 0; i < s->devices.size(); ++i) {
        if (hipSetDevice(s->devices[i]) != hipSuccess) (void)hipGetLastError();
        if (i < s->streams.size() && s->streams[i]) (void)hipStreamDestroy(s->streams[i]);
        if (i < s->gatherStreams.size() && s->gatherStreams[i]) (void)hipStreamDestroy(s->gatherStreams[i]);
        if (i < s->start.size() && s->start[i]) (void)hipEventDestroy(s->start[i]);
        if (i < s->stop.size() && s->stop[i]) (void)hipEventDestroy(s->stop[i]);
        if (i < s->gatherStart.size() && s->gatherStart[i]) (void)hipEventDestroy(s->gatherStart[i]);
        for (int b = 0; b < 2; ++b) {
            if (i < s->computed[b].size() && s->computed[b][i]) (void)hipEventDestroy(s->computed[b][i]);
            if (i < s->gathered[b].size() && s->gathered[b][i]) (void)hipEventDestroy(s->gathered[b][i]);
        }
        if (i < s->plans.size() && s->plans[i]) bsmr_plan_destroy(s->plans[i]);
    }
    delete s;
    return BSMR_OK;
}

int bsmr_sharded_create(bsmr_sharded** out, const int* devices, uint32_t num_devices,
                        const bsmr_rphm_desc* const* shard_descs, const uint32_t* row_begin,
                        const bsmr_plan_options* options) {
    if (!out || !devices || !shard_descs || !row_begin || num_devices == 0 || num_devices > 64) return BSMR_ERR_INVALID_ARG;
    *out = nullptr;
    for (uint32_t i = 0; i < num_devices; ++i) {
        if (!shard_descs[i] || row_begin[i + 1] < row_begin[i] || shard_descs[i]->M != row_begin[i + 1] - row_begin[i] ||
            shard_descs[i]->N != shard_descs[0]->N)
            return BSMR_ERR_INVALID_ARG;
    }
    DeviceRestore restore;
    bsmr_sharded* s = new (std::nothrow) bsmr_sharded;
    if (!s) return BSMR_ERR_OOM;
    s->devices.assign(devices, devices + num_devices);
    for (uint32_t i = 0; i < num_devices; ++i) {   // (a device may serve several shards: the distinct ones, root first)
        size_t u = 0;
        while (u < s->uniqueDevices.size() && s->uniqueDevices[u] != devices[i]) ++u;
        if (u == s->uniqueDevices.size()) s->uniqueDevices.push_back(devices[i]);
        s->rankOf.push_back((int)u);
    }
    s->rowBegin.assign(row_begin, row_begin + num_devices + 1);
    s->M = row_begin[num_devices] - row_begin[0];
    s->N = shard_descs[0]->N;
    s->entryBegin.assign(num_devices + 1, 0);
    s->plans.assign(num_devices, nullptr);
    s->streams.assign(num_devices, nullptr);
    s->gatherStreams.assign(num_devices, nullptr);
    s->start.assign(num_devices, nullptr);
    s->stop.assign(num_devices, nullptr);
    s->gatherStart.assign(num_devices, nullptr);
    for (int b = 0; b < 2; ++b) {
        s->computed[b].assign(num_devices, nullptr);
        s->gathered[b].assign(num_devices, nullptr);
    }
    int st = BSMR_OK;
    for (uint32_t i = 0; i < num_devices; ++i) s->entryBegin[i + 1] = s->entryBegin[i] + shard_descs[i]->nnz;
    {
        // the shards' plans: one host thread per distinct device (BSMR_SHARD_BUILD_THREADS=n: n threads whatever the devices
        // are), every thread its shards in turn.  (A row range without rows - more shards than non-empty row panels - has no
        // plan and takes no part in a step.)
        std::vector<std::vector<uint32_t>> lanes;
        const int forced = std::max(0, envInt("BSMR_SHARD_BUILD_THREADS", 0));
        if (forced > 0) {
            lanes.assign(std::min<size_t>((size_t)forced, num_devices), {});
            for (uint32_t i = 0; i < num_devices; ++i) lanes[i % lanes.size()].push_back(i);
        } else {
            lanes.assign(s->uniqueDevices.size(), {});
            for (uint32_t i = 0; i < num_devices; ++i) lanes[(size_t)s->rankOf[i]].push_back(i);
        }
        std::vector<int> laneStatus(lanes.size(), BSMR_OK);
        std::vector<std::string> laneError(lanes.size());
        auto buildLane = [&](size_t t) {
            for (const uint32_t i : lanes[t]) {
                if (shard_descs[i]->M == 0 || shard_descs[i]->nnz == 0) continue;
                int one = BSMR_ERR_INVALID_ARG;
                try {
                    one = bsmr_plan_create_ex(&s->plans[i], devices[i], shard_descs[i], options);
                } catch (...) {
                    one = BSMR_ERR_OOM;
                }
                if (one != BSMR_OK) {
                    laneStatus[t] = one;
                    laneError[t] = g_lastHipError;   // (thread-local: carried to the calling thread below)
                    return;
                }
            }
        };
        if (lanes.size() <= 1) {
            if (!lanes.empty()) buildLane(0);
        } else {
            std::vector<std::thread> pool;
            for (size_t t = 0; t < lanes.size(); ++t) pool.emplace_back(buildLane, t);
            for (std::thread& th : pool) th.join();
        }
        for (size_t t = 0; t < lanes.size() && st == BSMR_OK; ++t)
            if (laneStatus[t] != BSMR_OK) {
                st = laneStatus[t];
                g_lastHipError = laneError[t];
            }
    }
    for (uint32_t i = 0; i < num_devices && st == BSMR_OK; ++i) {
        if (hipSetDevice(devices[i]) != hipSuccess) {
            (void)hipGetLastError();
            st = BSMR_ERR_NO_DEVICE;
            break;
        }
        if (!hipOk(hipStreamCreateWithFlags(&s->streams[i], hipStreamNonBlocking), "hipStreamCreate") ||
            !hipOk(hipStreamCreateWithFlags(&s->gatherStreams[i], hipStreamNonBlocking), "hipStreamCreate") ||
            !hipOk(hipEventCreate(&s->start[i]), "hipEventCreate") || !hipOk(hipEventCreate(&s->stop[i]), "hipEventCreate") ||
            !hipOk(hipEventCreate(&s->gatherStart[i]), "hipEventCreate") ||
            !hipOk(hipEventCreateWithFlags(&s->computed[0][i], hipEventDisableTiming), "hipEventCreate") ||
            !hipOk(hipEventCreateWithFlags(&s->computed[1][i], hipEventDisableTiming), "hipEventCreate") ||
            !hipOk(hipEventCreateWithFlags(&s->gathered[0][i], hipEventDisableTiming), "hipEventCreate") ||
            !hipOk(hipEventCreateWithFlags(&s->gathered[1][i], hipEventDisableTiming), "hipEventCreate"))
            st = BSMR_ERR_HIP;
    }
    if (st == BSMR_OK && s->uniqueDevices.size() > 1) {
        if (!rccl().ok()) {
            g_lastHipError = "librccl.so.1 could not be loaded (needed for more than one device)";
            st = BSMR_ERR_HIP;
        } else {
            s->comms.assign(s->uniqueDevices.size(), nullptr);
            const ncclResult_t r = rccl().commInitAll(s->comms.data(), (int)s->uniqueDevices.size(), s->uniqueDevices.data());
            if (r != ncclSuccess) {
                g_lastHipError = std::string("ncclCommInitAll: ") + rccl().errorString(r);
                st = BSMR_ERR_HIP;
            }
        }
    }
    if (st != BSMR_OK) {
        bsmr_sharded_destroy(s);
        return st;
    }
    *out = s;
    return BSMR_OK;
}

int bsmr_sharded_num_entries(const bsmr_sharded* s, uint64_t* total, uint64_t* per_shard) {
    if (!s || !total) return BSMR_ERR_INVALID_ARG;
    *total = s->entryBegin.back();
    if (per_shard)
        for (size_t i = 0; i < s->devices.size(); ++i) per_shard[i] = s->entryBegin[i + 1] - s->entryBegin[i];
    return BSMR_OK;
}

int bsmr_sharded_sddmm_host(bsmr_sharded* s, uint32_t K, const float* A_host, const float* B_host, float* P_host,
                            int mode, int iters, bsmr_sharded_timing* timing) {
    if (!s || !A_host || !B_host || !P_host) return BSMR_ERR_INVALID_ARG;
    if (K == 0 || (K & 31u)) return BSMR_ERR_UNSUPPORTED_K;
    if (iters <= 0) iters = 1;
    DeviceRestore restore;
    const size_t n = s->devices.size();
    const uint64_t nnz = s->entryBegin.back();
    // operands: shard i's rows of A, all of B, its part of P twice (the root holds the whole P, twice)
    if (s->K != K) {
        freeShardedBuffers(s);
        s->A.assign(n, nullptr);
        s->B.assign(n, nullptr);
        s->P[0].assign(n, nullptr);
        s->P[1].assign(n, nullptr);
        for (size_t i = 0; i < n; ++i) {
            BSMR_HIP(hipSetDevice(s->devices[i]));
            const size_t rows = s->rowBegin[i + 1] - s->rowBegin[i];
            const uint64_t part = i == 0 ? nnz : s->entryBegin[i + 1] - s->entryBegin[i];
            if (!hipOk(hipMalloc(reinterpret_cast<void**>(&s->A[i]), std::max<size_t>(rows * K * 4, 16)), "hipMalloc(A shard)") ||
                !hipOk(hipMalloc(reinterpret_cast<void**>(&s->B[i]), std::max<size_t>((size_t)s->N * K * 4, 16)), "hipMalloc(B replica)") ||
                !hipOk(hipMalloc(reinterpret_cast<void**>(&s->P[0][i]), std::max<size_t>(part * 4, 16)), "hipMalloc(P shard)") ||
                !hipOk(hipMalloc(reinterpret_cast<void**>(&s->P[1][i]), std::max<size_t>(part * 4, 16)), "hipMalloc(P shard)"))
                return BSMR_ERR_OOM;
            // (every entry of P is written by exactly one shard's SDDMM or arrives with the gather: a byte pattern that reads
            // as NaN makes an entry that nobody wrote visible to the caller)
            BSMR_HIP(hipMemsetAsync(s->P[0][i], 0xFF, std::max<size_t>(part * 4, 16), s->streams[i]));
            BSMR_HIP(hipMemsetAsync(s->P[1][i], 0xFF, std::max<size_t>(part * 4, 16), s->streams[i]));
            int st = s->plans[i] ? bsmr_plan_reserve(s->plans[i], K) : BSMR_OK;
            if (st != BSMR_OK) return st;
        }
        s->K = K;
    }
    for (size_t i = 0; i < n; ++i) {
        BSMR_HIP(hipSetDevice(s->devices[i]));
        const size_t rows = s->rowBegin[i + 1] - s->rowBegin[i];
        BSMR_HIP(hipMemcpyAsync(s->A[i], A_host + (size_t)(s->rowBegin[i] - s->rowBegin[0]) * K, rows * K * 4, hipMemcpyHostToDevice, s->streams[i]));
        BSMR_HIP(hipMemcpyAsync(s->B[i], B_host, (size_t)s->N * K * 4, hipMemcpyHostToDevice, s->streams[i]));
    }
    auto syncAll = [&]() -> int {
        for (size_t i = 0; i < n; ++i) {
            BSMR_HIP(hipSetDevice(s->devices[i]));
            BSMR_HIP(hipStreamSynchronize(s->streams[i]));
            BSMR_HIP(hipStreamSynchronize(s->gatherStreams[i]));
        }
        return BSMR_OK;
    };
    if (int st = syncAll()) return st;
    // one step into output buffers b: every device's SDDMM on its compute stream (which first waits until the gather that
    // last read these buffers, two steps ago, has delivered), then the gather on the gather streams
    auto step = [&](int b, bool waitForBuffer) -> int {
        for (size_t i = 0; i < n; ++i) {
            BSMR_HIP(hipSetDevice(s->devices[i]));
            if (waitForBuffer) BSMR_HIP(hipStreamWaitEvent(s->streams[i], s->gathered[b][i], 0));
            if (s->plans[i]) {
                float* dst = i == 0 ? s->P[b][0] + s->entryBegin[0] : s->P[b][i];
                const int st = bsmr_sddmm(s->plans[i], K, s->A[i], s->B[i], dst, mode, s->streams[i]);
                if (st != BSMR_OK) return st;
            }
            BSMR_HIP(hipEventRecord(s->computed[b][i], s->streams[i]));
        }
        return shardedGather(s, b);
    };
    int st = step(0, false);   // warm-up (workspaces, communicator channels)
    if (st == BSMR_OK) st = syncAll();
    if (st != BSMR_OK) return st;
    // what the two halves of a step cost by themselves (one step each, nothing overlapped): SDDMM on every device, then the gather
    float computeMs = 0.0f, gatherMs = 0.0f;
    {
        for (size_t i = 0; i < n; ++i) {
            BSMR_HIP(hipSetDevice(s->devices[i]));
            BSMR_HIP(hipEventRecord(s->start[i], s->streams[i]));
            if (s->plans[i]) {
                float* dst = i == 0 ? s->P[1][0] + s->entryBegin[0] : s->P[1][i];
                if ((st = bsmr_sddmm(s->plans[i], K, s->A[i], s->B[i], dst, mode, s->streams[i])) != BSMR_OK) return st;
            }
            BSMR_HIP(hipEventRecord(s->stop[i], s->streams[i]));
            BSMR_HIP(hipEventRecord(s->computed[1][i], s->streams[i]));
        }
        for (size_t i = 0; i < n; ++i) {
            BSMR_HIP(hipSetDevice(s->devices[i]));
            BSMR_HIP(hipEventSynchronize(s->stop[i]));
            float ms = 0.0f;
            BSMR_HIP(hipEventElapsedTime(&ms, s->start[i], s->stop[i]));
            computeMs = std::max(computeMs, ms);
        }
        for (size_t i = 0; i < n; ++i) {
            BSMR_HIP(hipSetDevice(s->devices[i]));
            BSMR_HIP(hipEventRecord(s->gatherStart[i], s->gatherStreams[i]));
        }
        if ((st = shardedGather(s, 1)) != BSMR_OK) return st;
        for (size_t i = 0; i < n; ++i) {
            BSMR_HIP(hipSetDevice(s->devices[i]));
            BSMR_HIP(hipEventRecord(s->stop[i], s->gatherStreams[i]));
        }
        for (size_t i = 0; i < n; ++i) {
            BSMR_HIP(hipSetDevice(s->devices[i]));
            BSMR_HIP(hipEventSynchronize(s->stop[i]));
            float ms = 0.0f;
            BSMR_HIP(hipEventElapsedTime(&ms, s->gatherStart[i], s->stop[i]));
            gatherMs = std::max(gatherMs, ms);
        }
        if ((st = syncAll()) != BSMR_OK) return st;
    }
    // the timed steps, pipelined: step k computes into buffers k & 1 while the gather of step k - 1 delivers the others
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n; ++i) {
        BSMR_HIP(hipSetDevice(s->devices[i]));
        BSMR_HIP(hipEventRecord(s->start[i], s->streams[i]));
    }
    for (int it = 0; it < iters && st == BSMR_OK; ++it) st = step(it & 1, true);
    if (st != BSMR_OK) return st;
    float maxMs = 0.0f;
    for (size_t i = 0; i < n; ++i) {   // a step has ended when its part has reached the root: the stop events sit behind the last gather
        BSMR_HIP(hipSetDevice(s->devices[i]));
        BSMR_HIP(hipEventRecord(s->stop[i], s->gatherStreams[i]));
    }
    for (size_t i = 0; i < n; ++i) {
        BSMR_HIP(hipSetDevice(s->devices[i]));
        BSMR_HIP(hipEventSynchronize(s->stop[i]));
        float ms = 0.0f;
        BSMR_HIP(hipEventElapsedTime(&ms, s->start[i], s->stop[i]));
        maxMs = std::max(maxMs, ms);
    }
    if ((st = syncAll()) != BSMR_OK) return st;
    const double wallMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    BSMR_HIP(hipSetDevice(s->devices[0]));
    if (nnz) BSMR_HIP(hipMemcpy(P_host, s->P[(iters - 1) & 1][0], nnz * 4, hipMemcpyDeviceToHost));
    if (timing) {
        bsmr_sharded_timing t{};
        t.step_ms = maxMs / iters;
        t.wall_ms = (float)(wallMs / iters);
        t.num_devices = (uint32_t)n;
        t.compute_ms = computeMs;
        t.gather_ms = gatherMs;
        *timing = t;
    }
    return BSMR_OK;
}

}  // extern "C"
