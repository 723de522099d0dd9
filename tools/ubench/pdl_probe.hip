// pdl_probe.hip -- is a "software programmatic dependent launch" worth building for the batch-1 decode chain?
//
// A chain of S weight-streaming stages, each depending on the previous one through an 8 KB activation vector (the
// decode layer's shape: every workgroup reads the whole vector, streams its own weight slice, writes 16 outputs):
//   mode 0  plain: one stream, kernel boundaries are the dependency (what the model does today)
//   mode 1  overlapped: stages alternate between TWO streams with no edge between neighbours; a stage issues its first
//           weight loads (the register ring) at once, waits for them to LAND (so its own queue is empty), then polls the
//           previous stage's completion word, reads the vector with sc1 loads and goes on.  Completion = write-through
//           (sc1) output stores, every wave drains vmcnt, one relaxed agent-scope add per workgroup on a counter sharded
//           by blockIdx % 8, the shard's last arriver bumps the top word that consumers poll (relaxed, s_sleep).
// Every word of the vector is CHECKED by every workgroup of every stage (stale data = error count), every spin is
// bounded (an expiry is reported, never a hang).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/pdl_probe tools/ubench/pdl_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kT = 512, kRing = 7, kVec = 4096;   // vector elements (u16) = 8 KB
struct Ctl { unsigned shard[8][32]; unsigned top[32]; };   // one 128-B line per word
struct Err { unsigned stale, expired, maxspin, pad; unsigned exp_stage[64]; unsigned long long t_start[64], t_end[64]; };

typedef unsigned u4 __attribute__((ext_vector_type(4)));

// vin: this stage's input vector (all of it is read by every workgroup); vout: its output (workgroup b READS its own 16
// elements first, like the residual add, then writes them) -- so L1 / L2 lines of vout with stale neighbours exist on
// every CU / XCD when the next stage reads vout as its input.  mode 1: sc1 loads of the vector; mode 2: agent-scope
// acquire fence (one lane) + plain loads (the guide's R1 form).
__global__ __launch_bounds__(kT, 2) void k_stage(const u4* __restrict__ w, int items_per_wave, const unsigned short* vin,
                                                 unsigned short* vout, int stage, const Ctl* wait_ctl, Ctl* my_ctl,
                                                 int overlapped, Err* err, unsigned* sink, int stamp = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (stamp && threadIdx.x == 0) atomicMin(&err->t_start[stage & 63], wall_clock64());
    const u4* src = w + ((size_t)(blockIdx.x * 8 + wave) * items_per_wave) * 64 + lane;   // 1 KiB items, contiguous per wave
    u4 ring[kRing];
    int issued = 0;
#pragma unroll
    for (int s = 0; s < kRing; ++s) {
        ring[s] = issued < items_per_wave ? __builtin_nontemporal_load(src + (size_t)issued * 64) : (u4){0, 0, 0, 0};
        ++issued;
    }
    if (overlapped) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ring landed: this CU's queue is empty for the poll
        if (wait_ctl && threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(&wait_ctl->top[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000u) { atomicAdd(&err->expired, 1u); atomicAdd(&err->exp_stage[stage & 63], 1u); break; }
            }
            atomicMax(&err->maxspin, spins);
            if (overlapped == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    unsigned short resid = 0;
    if (threadIdx.x < 16) resid = vout[blockIdx.x * 16 + threadIdx.x];   // plain load: the line stays in L1 / L2
    // the activation vector: 16 B per thread; sc1 (agent-scope) loads when overlapped, plain otherwise
    unsigned long long xa, xb;
    const unsigned long long* xv = reinterpret_cast<const unsigned long long*>(vin) + threadIdx.x * 2;
    if (overlapped == 1) {
        xa = __hip_atomic_load(xv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        xb = __hip_atomic_load(xv + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        xa = xv[0];
        xb = xv[1];
    }
    const unsigned long long want = 0x0001000100010001ull * (unsigned)(stage & 0xffff);
    if (xa != want || xb != want) atomicAdd(&err->stale, 1u);
    // stream the rest
    unsigned acc = (unsigned)xa;
    for (int i = 0; i < items_per_wave; i += kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            const u4 v = ring[s];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
            ring[s] = issued < items_per_wave ? __builtin_nontemporal_load(src + (size_t)issued * 64) : (u4){0, 0, 0, 0};
            ++issued;
        }
    }
    if (acc == 0x12345678u) sink[threadIdx.x] = acc;
    // outputs: workgroup b owns elements 16 b .. 16 b + 15 (in place, like the residual stream)
    if (threadIdx.x < 16) {
        unsigned short* dst = vout + blockIdx.x * 16 + threadIdx.x;
        const unsigned short nv = (unsigned short)(((stage + 1) & 0xffff) + (resid == 0xffff ? 1 : 0));
        if (overlapped) __hip_atomic_store(dst, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *dst = nv;
    }
    if (overlapped) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int sh = blockIdx.x & 7;
            const unsigned members = (gridDim.x - sh + 7) / 8;
            const unsigned old = __hip_atomic_fetch_add(&my_ctl->shard[sh][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == members) __hip_atomic_fetch_add(&my_ctl->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (stamp && threadIdx.x == 0) atomicMax(&err->t_end[stage & 63], wall_clock64());
}

int main(int argc, char** argv) {
    const int stages = argc > 1 ? atoi(argv[1]) : 40;
    const int grid = 256;
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    const size_t wbytes = (size_t)grid * 8 * 32 * 1024;   // up to 32 items per wave = 256 KiB per workgroup
    const int nbuf = 6;                                     // rotate: 6 x 64 MB > Infinity Cache
    std::vector<u4*> w(nbuf);
    for (auto& p : w) { CK(hipMalloc(&p, wbytes)); CK(hipMemset(p, 0x5a, wbytes)); }
    unsigned short* vec; CK(hipMalloc(&vec, kVec * 2 * 2));   // ping-pong pair
    Ctl* ctl; CK(hipMalloc(&ctl, sizeof(Ctl) * stages));
    Err* err; CK(hipMalloc(&err, sizeof(Err)));
    unsigned* sink; CK(hipMalloc(&sink, 4096));
    if (argc == 3) {   // eager two-stream run of 8 stages with a timeline (100 MHz ticks -> us)
        const int ns = 8, ipw = 14;
        std::vector<Err> hz(1);
        CK(hipMemset(err, 0, sizeof(Err)));
        Err init{}; for (int i = 0; i < 64; ++i) init.t_start[i] = ~0ull;
        CK(hipMemcpy(err, &init, sizeof(Err), hipMemcpyHostToDevice));
        CK(hipMemset(ctl, 0, sizeof(Ctl) * stages)); CK(hipMemset(vec, 0, kVec * 4));
        CK(hipDeviceSynchronize());
        for (int st = 0; st < ns; ++st)
            hipLaunchKernelGGL(k_stage, dim3(grid), dim3(kT), 0, (st & 1) ? s1 : s0, w[st % nbuf], ipw, vec + (st & 1) * kVec,
                               vec + ((st + 1) & 1) * kVec, st, st > 0 ? ctl + st - 1 : nullptr, ctl + st, 1, err, sink, 1);
        CK(hipDeviceSynchronize());
        Err h; CK(hipMemcpy(&h, err, sizeof(h), hipMemcpyDeviceToHost));
        printf("eager two-stream: stale=%u expired=%u\n", h.stale, h.expired);
        for (int st = 0; st < ns; ++st)
            printf("  stage %d: start %8.2f us  end %8.2f us  expired WGs %u\n", st, (h.t_start[st] - h.t_start[0]) / 100.0,
                   (h.t_end[st] - h.t_start[0]) / 100.0, h.exp_stage[st]);
        return 0;
    }
    if (argc > 3) {   // eager launches from this host thread: one stream with kernel boundaries vs two streams + completion words
        const int ns = 400;
        Ctl* ctl2; CK(hipMalloc(&ctl2, sizeof(Ctl) * ns));
        for (int ipw : {4, 8, 14, 28}) for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f;
            Err h{};
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(ctl2, 0, sizeof(Ctl) * ns)); CK(hipMemset(vec, 0, kVec * 4)); CK(hipMemset(err, 0, sizeof(Err)));
                CK(hipDeviceSynchronize());
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0, s0));
                if (mode) { hipEvent_t f; CK(hipEventCreate(&f)); CK(hipEventRecord(f, s0)); CK(hipStreamWaitEvent(s1, f, 0)); }
                for (int st = 0; st < ns; ++st)
                    hipLaunchKernelGGL(k_stage, dim3(grid), dim3(kT), 0, (mode && (st & 1)) ? s1 : s0, w[st % nbuf], ipw, vec + (st & 1) * kVec,
                                       vec + ((st + 1) & 1) * kVec, st, (mode && st > 0) ? ctl2 + st - 1 : nullptr, ctl2 + st, mode, err, sink, 0);
                if (mode) { hipEvent_t j; CK(hipEventCreate(&j)); CK(hipEventRecord(j, s1)); CK(hipStreamWaitEvent(s0, j, 0)); }
                CK(hipEventRecord(e1, s0)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
                CK(hipMemcpy(&h, err, sizeof(h), hipMemcpyDeviceToHost));
            }
            printf("eager %3d KiB/WG %s: %6.2f us/stage  stale=%u expired=%u maxspin=%u\n", ipw * 8, mode ? "two streams + flags" : "one stream          ",
                   best * 1e3 / ns, h.stale, h.expired, h.maxspin);
        }
        return 0;
    }
    for (int ipw : {4, 8, 14, 28}) {                        // items (KiB) per wave: 32 / 64 / 112 / 224 KiB per workgroup
        for (int mode = 0; mode < 3; ++mode) {
            hipGraph_t g; hipGraphExec_t ge;
            hipEvent_t fork, join; CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
            CK(hipMemsetAsync(ctl, 0, sizeof(Ctl) * stages, s0));
            CK(hipMemsetAsync(vec, 0, kVec * 2 * 2, s0));
            if (mode) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
            for (int st = 0; st < stages; ++st) {
                hipStream_t s = (mode && (st & 1)) ? s1 : s0;
                // stage st reads buffer st & 1 (every element == st after the first lap ... see below) and writes the other
                hipLaunchKernelGGL(k_stage, dim3(grid), dim3(kT), 0, s, w[st % nbuf], ipw, vec + (st & 1) * kVec,
                                   vec + ((st + 1) & 1) * kVec, st, (mode && st > 0) ? ctl + st - 1 : nullptr, ctl + st, mode, err, sink,
                                   (mode == 1 && ipw == 14) ? 1 : 0);
            }
            if (mode) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
            CK(hipStreamEndCapture(s0, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            { Err init{}; for (int i = 0; i < 64; ++i) init.t_start[i] = ~0ull; CK(hipMemcpy(err, &init, sizeof(Err), hipMemcpyHostToDevice)); }
            CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
            if (mode == 1 && ipw == 14) {
                Err h1; CK(hipMemcpy(&h1, err, sizeof(h1), hipMemcpyDeviceToHost));
                for (int st = 0; st < 8; ++st)
                    printf("  graph stage %d: start %8.2f us  end %8.2f us  expired WGs %u\n", st, (h1.t_start[st] - h1.t_start[0]) / 100.0,
                           (h1.t_end[st] - h1.t_start[0]) / 100.0, h1.exp_stage[st]);
            }
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int reps = 5;
            CK(hipEventRecord(e0, s0));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s0));
            CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            Err h; CK(hipMemcpy(&h, err, sizeof(h), hipMemcpyDeviceToHost));
            const double us = ms * 1e3 / (reps * stages), mb = (double)grid * 8 * ipw * 1024 / 1e6;
            printf("%3d KiB/WG (%5.1f MB/stage) %s: %6.2f us/stage  (%5.2f TB/s)  stale=%u expired=%u maxspin=%u\n", ipw * 8, mb,
                   mode == 0 ? "plain        " : mode == 1 ? "overlap sc1  " : "overlap fence", us, mb / us, h.stale, h.expired, h.maxspin);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
