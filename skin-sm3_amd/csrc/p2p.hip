// Peer-to-peer exchange of SyncBatchNorm statistic sums between the ranks of ONE node (the reference's
// exchange is torch.nn.SyncBatchNorm's all_gather of (mean, invstd, count), tools/backbone_train.py:510 -- here the fp64
// [views][2C] sums the BatchNorm kernels already produce).
//
// Why not RCCL for this: a step issues 110 such exchanges per execution lane, each a few KB and each on the critical path
// (the next kernel of the lane needs the result).  Through torch.distributed every one of them is two cross-stream event
// dependencies plus a collective launch (7 us with ONE rank, where the collective is the identity; an all-reduce of a few
// KB between devices is a multiple of that).  Here the exchange is ONE kernel on the lane's own stream (3 us on one rank;
// profiles/r03b_bench_force_dp_world1.txt):
//
//   every rank owns a mailbox (device memory, opened by all peers through hipIpc); slot = seq & 1, per source rank a data
//   area of kMaxN doubles, per (slot, source, block) a 64-bit flag.
//   block b of rank r:  (1) stores its chunk of r's sums into mailbox[p][slot][r] of every rank p (self included),
//                       (2) waits for those stores to be acknowledged, then flag[p][slot][r][b] = seq,
//                       (3) waits until flag[r][slot][q][b] == seq for every source q  (wall-clock timeout -> *err = 1, the
//                           caller's sums become NaN and the kernel returns: no wave spins forever; once *err is set
//                           every later exchange returns NaN at once),
//                       (4) adds the world's chunks in RANK ORDER and writes the result in place: bit-identical on every
//                           rank.  (Every mailbox access is a system-scope relaxed atomic; no fences -- see the kernel.)
//   Two slots suffice: a rank can enter exchange i + 2 (same slot as i) only after every peer has raised its flag for
//   i + 1, which a peer does only after it has finished reading exchange i (stream order).
//
// Opt-in (SM3_SYNCBN_P2P=1, sm3hip/p2p.py), RCCL stays the default: two processes sharing one GPU exercise every line of
// this file (tests/test_p2p_gpu.py), but the xGMI path between devices has never run -- this pool has one-GPU boxes only.
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace {

constexpr int kMaxWorld = 8;
constexpr int kMaxN = 16384 + 64;  // doubles per message: [bn3 | downsample][2 views][2 x 2048 channels]
constexpr int kChunk = 1024;       // doubles per block
constexpr int kMaxBlocks = (kMaxN + kChunk - 1) / kChunk;

struct Mailbox {
    double data[2][kMaxWorld][kMaxN];
    unsigned long long flag[2][kMaxWorld][kMaxBlocks];
};

struct Peers {
    Mailbox* box[kMaxWorld];
};

__global__ __launch_bounds__(256) void p2p_allreduce_kernel(double* __restrict__ buf, int n, Peers peers, int rank, int world,
                                                            unsigned long long seq, int* __restrict__ err,
                                                            unsigned long long timeout_ticks) {
    const int b = blockIdx.x, slot = (int)(seq & 1);
    const int i0 = b * kChunk, i1 = min(n, i0 + kChunk);
    // An exchange that has already failed on this rank stays failed: no further waiting (a dead peer would cost its full
    // timeout 220 times per step), and the caller's sums are replaced by NaN so that whatever is computed from them -- the
    // BatchNorm statistics, the loss the trainer returns -- shows it, with or without a host read of *err.
    // ONE thread reads the flag and the block branches on the shared copy: the other lane's exchange kernel shares the word
    // and may set it while this block's waves are starting -- per-thread reads could send some waves home and others into
    // the barriers below.
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_bad != 0) {
        for (int i = i0 + threadIdx.x; i < i1; i += 256) buf[i] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    // Every access to a mailbox is a system-scope atomic (relaxed): such stores write through and such loads read past the
    // caches, so no release / acquire FENCE is needed (a system-scope fence writes back and invalidates the whole L2, paid by
    // the NEXT kernel of the lane).  Ordering comes from the in-order issue of a wave plus
    // s_waitcnt vmcnt(0): the data stores are acknowledged before the flag store is issued, and the data loads are issued
    // after the flag load has returned.
    // (1) my chunk into everybody's mailbox
    for (int p = 0; p < world; ++p) {
        double* dst = peers.box[p]->data[slot][rank];
        for (int i = i0 + threadIdx.x; i < i1; i += 256)
            __hip_atomic_store(dst + i, buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (2) raise my flag everywhere
    if ((int)threadIdx.x < world)
        __hip_atomic_store(&peers.box[threadIdx.x]->flag[slot][rank][b], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // (3) wait for everybody's flag in MY mailbox  (s_bad is 0 here, and every wave has read it: the barrier above)
    if ((int)threadIdx.x < world) {
        const unsigned long long* f = &peers.box[rank]->flag[slot][threadIdx.x][b];
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz wall clock
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
                s_bad = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s_bad) {
        if (threadIdx.x == 0) atomicExch(err, 1);
        for (int i = i0 + threadIdx.x; i < i1; i += 256) buf[i] = __longlong_as_double(0x7ff8000000000000LL);
        return;
    }
    // (4) the world's chunks, added in rank order
    const Mailbox* mine = peers.box[rank];
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        double a = 0.0;
        for (int q = 0; q < world; ++q)
            a += __hip_atomic_load(&mine->data[slot][q][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        buf[i] = a;
    }
}

}  // namespace

extern "C" int sm3_p2p_mailbox_bytes(void) { return (int)sizeof(Mailbox); }
extern "C" int sm3_p2p_max_elems(void) { return kMaxN; }
extern "C" int sm3_p2p_max_world(void) { return kMaxWorld; }
// Byte offsets inside a mailbox of the data area [first, first + bytes) that source rank `src` writes for exchange slot
// `slot`, and of the 8-byte arrival flag of its block `block` -- computed from the SAME struct the kernel addresses, so that a
// host test can check without a GPU that the areas of 8 ranks x 2 slots x every block are disjoint and inside the allocation.
extern "C" int sm3_p2p_layout(int slot, int src, int block, int64_t* data_first, int64_t* data_bytes, int64_t* flag_off,
                              int* elems_per_block) {
    if (slot < 0 || slot > 1 || src < 0 || src >= kMaxWorld || block < 0 || block >= kMaxBlocks || !data_first || !data_bytes ||
        !flag_off)
        return SM3_EINVAL;
    const uintptr_t base = (uintptr_t)1 << 40;  // any address: only differences are taken
    const Mailbox* m = reinterpret_cast<const Mailbox*>(base);
    *data_first = (int64_t)((uintptr_t)&m->data[slot][src][0] - base);
    *data_bytes = (int64_t)sizeof(m->data[slot][src]);
    *flag_off = (int64_t)((uintptr_t)&m->flag[slot][src][block] - base);
    if (elems_per_block) *elems_per_block = kChunk;
    return 0;
}

// Mailboxes are FINE-GRAINED device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained), what RCCL uses for its own
// IPC flag / data buffers): stores of another agent become visible to a kernel that is already running and polling, which
// coarse-grained hipMalloc memory promises only at kernel boundaries.  Plain hipMalloc is the fallback when the
// fine-grained allocation or its IPC export fails; *kind_out says which one this mailbox is (1 fine-grained, 0 plain) so a
// run can record what it measured.
static hipError_t alloc_export(void** p, hipIpcMemHandle_t* h, bool fine) {
    hipError_t e = fine ? hipExtMallocWithFlags(p, sizeof(Mailbox), hipDeviceMallocFinegrained) : hipMalloc(p, sizeof(Mailbox));
    if (e != hipSuccess) {
        *p = nullptr;
        return e;
    }
    e = hipMemset(*p, 0, sizeof(Mailbox));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(h, *p);
    if (e != hipSuccess) {
        (void)hipFree(*p);
        *p = nullptr;
    }
    return e;
}

extern "C" int sm3_p2p_alloc(void** ptr, void* ipc_handle_64, int* kind_out) {
    if (!ptr || !ipc_handle_64) return SM3_EINVAL;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
    const char* v = getenv("SM3_P2P_FINEGRAINED");  // 0: plain hipMalloc (A/B of the two memory types)
    const bool want_fine = !(v && atoi(v) == 0);
    void* p = nullptr;
    int kind = 0;
    hipError_t e = hipErrorUnknown;
    if (want_fine) {
        e = alloc_export(&p, (hipIpcMemHandle_t*)ipc_handle_64, true);
        if (e == hipSuccess) kind = 1;
        else (void)hipGetLastError();  // the fallback below starts from a clean error state
    }
    if (e != hipSuccess) e = alloc_export(&p, (hipIpcMemHandle_t*)ipc_handle_64, false);
    if (e != hipSuccess) return (int)e;
    *ptr = p;
    if (kind_out) *kind_out = kind;
    return 0;
}

extern "C" int sm3_p2p_open(const void* ipc_handle_64, void** ptr) {
    if (!ptr || !ipc_handle_64) return SM3_EINVAL;
    hipIpcMemHandle_t h;
    std::memcpy(&h, ipc_handle_64, sizeof(h));
    hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int sm3_p2p_close(void* ptr) { return ptr ? (int)hipIpcCloseMemHandle(ptr) : SM3_EINVAL; }
extern "C" int sm3_p2p_free(void* ptr) { return ptr ? (int)hipFree(ptr) : SM3_EINVAL; }

extern "C" int sm3_p2p_allreduce_f64(double* buf, int n, void* const* mailboxes, int rank, int world, uint64_t seq, int* err_flag,
                                     double timeout_s, void* stream) {
    if (!buf || !mailboxes || !err_flag || n <= 0 || n > kMaxN || world < 1 || world > kMaxWorld || rank < 0 || rank >= world ||
        seq == 0)
        return SM3_EINVAL;
    Peers peers;
    for (int p = 0; p < kMaxWorld; ++p) peers.box[p] = p < world ? (Mailbox*)mailboxes[p] : nullptr;
    for (int p = 0; p < world; ++p)
        if (!peers.box[p]) return SM3_EINVAL;
    const unsigned long long ticks = (unsigned long long)((timeout_s > 0 ? timeout_s : 10.0) * 1e8);
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3((n + kChunk - 1) / kChunk), dim3(256), 0, (hipStream_t)stream, buf, n, peers,
                       rank, world, (unsigned long long)seq, err_flag, ticks);
    SM3_CHECK_LAUNCH();
    return 0;
}
