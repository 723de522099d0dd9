// cluster_gather_probe.hip -- VERDICT r05 item 2, "probe first": what does a 32-CU all-gather of 1.5 KB cost INSIDE a launch,
// under a live weight stream?  The un-priced fusion of the batch-1 decode layer is qkv -> attention inside kv-head clusters: the
// 32 workgroups that produce the q / k / v columns of ONE kv head (4 q heads + k + v = 768 columns = 24 per workgroup) would hand
// their 48 bytes to each other and go on as that head's attention workgroups, with the cached K / V already prefetched.  What
// that hand-off replaces is the qkv -> attention kernel boundary (1.2 .. 1.9 us, guide row "boundary") plus the attention launch's
// own q fetch.
//
// Chain of dependent launches in a hipGraph (like a decode step), 256 workgroups x 8 waves (one per CU):
//   stage body = stream `kib` KiB of HBM-cold weights per workgroup through a register ring (8 x 1 KiB per wave in flight),
//   then PUBLISH 12 tagged 8-byte granules {2 halfs, epoch} (= 24 columns) with one sc1 store each from 12 lanes, then GATHER the
//   cluster's 32 x 12 = 384 granules (3 KB): wave 0 sweeps them with six 8-byte sc1 loads per lane-row until every tag carries
//   the launch's epoch (relaxed polling, s_sleep between sweeps), stores the payload to LDS, barrier.
//   Cluster = the 32 workgroups of one XCD (blockIdx % 8: observed placement, used for speed only) or 32 consecutive ones
//   (cross-XCD) -- both measured.
//   LATE = 1: half of the chip (every other cluster) streams 4 x as much and finishes late -- the gathering CUs then poll next
//   to a live weight stream ("streaming" in the guide's terms); LATE = 0: everybody streams the same amount ("parked").
// Stamps per workgroup (100 MHz wall clock): entry, stream done, own publish issued, gather complete.  Reported: per stage,
//   hand-off = gather complete - the cluster's LAST publish (p50 / p90 / max over clusters x workgroups),
//   own = gather complete - own publish, and the whole stage time against the same chain WITHOUT publish / gather.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/cluster_gather_probe tools/ubench/cluster_gather_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int kT = 512, kRing = 8;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct P {
    const u4* w;
    int kib;                       // KiB per wave of the short streamers
    unsigned long long* gran;      // [256 workgroups][12] tagged granules
    unsigned epoch;
    int xcd_cluster, late, gather;
    unsigned* sink;
    unsigned long long* stamps;    // [stage][wg][4] or null
    int stage;
};

__global__ __launch_bounds__(kT, 2) void k_stage(const P p) {
    __shared__ unsigned payload[384];
    __shared__ int done;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (p.stamps && threadIdx.x == 0) t0 = wall_clock64();
    // cluster id and rank inside it
    const int b = blockIdx.x;
    const int cluster = p.xcd_cluster ? (b & 7) : (b >> 5), crank = p.xcd_cluster ? (b >> 3) : (b & 31);
    int n = p.kib;
    if (p.late && (cluster & 1)) n *= 4;
    u4 ring[kRing];
    auto issue = [&](int slot, int j) {
        if (j < n) ring[slot] = __builtin_nontemporal_load(p.w + ((size_t)(b * 8 + wave) * 64 + j) * 64 + lane);
        else ring[slot] = (u4){0, 0, 0, 0};
    };
#pragma unroll
    for (int s = 0; s < kRing; ++s) issue(s, s);
    unsigned acc = 0;
    for (int j0 = 0; j0 < n; j0 += kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            const u4 v = ring[s];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
            issue(s, j0 + s + kRing);
        }
    }
    __syncthreads();
    if (p.stamps && threadIdx.x == 0) t1 = wall_clock64();
    if (p.gather) {
        // ---- publish: 12 granules of this workgroup, one write-through 8-byte store each
        if (threadIdx.x < 12) {
            const unsigned long long g = ((unsigned long long)p.epoch << 32) | (acc & 0xffffu) | ((unsigned)threadIdx.x << 16);
            __hip_atomic_store(p.gran + (size_t)b * 12 + threadIdx.x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (p.stamps && threadIdx.x == 0) t2 = wall_clock64();
        // ---- gather: wave 0 sweeps the cluster's 384 granules (6 per lane) until all carry this launch's epoch
        if (wave == 0) {
            bool ok = false;
            int spins = 0;
            while (!ok && spins < 100000) {
                unsigned long long g[6];
                bool all = true;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const int idx = i * 64 + lane, peer = idx / 12, k = idx % 12;
                    const int pb = p.xcd_cluster ? (peer * 8 + cluster) : (cluster * 32 + peer);
                    g[i] = __hip_atomic_load(p.gran + (size_t)pb * 12 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) all = all && (unsigned)(g[i] >> 32) == p.epoch;
                ok = __builtin_amdgcn_ballot_w64(!all) == 0;
                if (ok) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) payload[i * 64 + lane] = (unsigned)g[i];
                } else {
                    __builtin_amdgcn_s_sleep(2);
                    ++spins;
                }
            }
            if (lane == 0) done = ok ? 1 : -1;
        }
        __syncthreads();
        if (p.stamps && threadIdx.x == 0) t3 = wall_clock64();
        acc ^= payload[(threadIdx.x * 5 + crank) % 384] ^ (unsigned)done;
    }
    if (acc == 0x12345678u) p.sink[threadIdx.x] = acc;
    if (p.stamps && threadIdx.x == 0) {
        unsigned long long* s = p.stamps + ((size_t)p.stage * 256 + b) * 4;
        s[0] = t0; s[1] = t1; s[2] = t2; s[3] = t3;
    }
}

int main() {
    const int grid = 256, stages = 40, nbuf = 6;
    hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    const size_t wbytes = (size_t)grid * 8 * 64 * 1024;            // up to 64 KiB per wave
    std::vector<u4*> w(nbuf);
    for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&w[i], wbytes)); CK(hipMemset(w[i], 0x5a, wbytes)); }
    unsigned long long* gran; CK(hipMalloc(&gran, 256 * 12 * 8)); CK(hipMemset(gran, 0, 256 * 12 * 8));
    unsigned* sink; CK(hipMalloc(&sink, 4096));
    unsigned long long* stamps; CK(hipMalloc(&stamps, (size_t)stages * 256 * 4 * 8));
    unsigned epoch = 1;
    auto run = [&](const char* name, int kib, int xcd_cluster, int late, int gather) {
        double stage_us = 0;
        for (int pass = 0; pass < 2; ++pass) {       // pass 0: timing, pass 1: stamps
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
            // the epoch is baked into the captured launches: every replay must see fresh tags, so a replay uses epochs nobody used
            const unsigned base = epoch;
            for (int st = 0; st < stages; ++st) {
                P p{w[st % nbuf], kib, gran, base + (unsigned)st, xcd_cluster, late, gather, sink, pass ? stamps : nullptr, st};
                hipLaunchKernelGGL(k_stage, dim3(grid), dim3(kT), 0, s0, p);
            }
            epoch += stages;
            CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            if (pass == 0) {
                // one graph = one set of epochs: time single replays with the tags reset in between
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipMemsetAsync(gran, 0, 256 * 12 * 8, s0));
                    CK(hipEventRecord(e0, s0));
                    CK(hipGraphLaunch(ge, s0));
                    CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep) best = ms < best ? ms : best;
                }
                stage_us = best * 1e3 / stages;
                printf("%-46s %2d KiB/wave: %6.2f us/stage", name, kib, stage_us);
            } else {
                CK(hipMemsetAsync(gran, 0, 256 * 12 * 8, s0));
                CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
                std::vector<unsigned long long> h((size_t)stages * 256 * 4);
                CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
                std::vector<double> hand, own, streamd;
                for (int st = 4; st < stages; ++st) {
                    for (int c = 0; c < 8; ++c) {
                        unsigned long long last_pub = 0;
                        for (int r = 0; r < 32; ++r) {
                            const int b = xcd_cluster ? r * 8 + c : c * 32 + r;
                            last_pub = std::max(last_pub, h[((size_t)st * 256 + b) * 4 + 2]);
                        }
                        for (int r = 0; r < 32; ++r) {
                            const int b = xcd_cluster ? r * 8 + c : c * 32 + r;
                            const unsigned long long* s = &h[((size_t)st * 256 + b) * 4];
                            if (gather) {
                                hand.push_back(((double)s[3] - (double)last_pub) * 0.01);
                                own.push_back(((double)s[3] - (double)s[2]) * 0.01);
                            }
                            streamd.push_back(((double)s[1] - (double)s[0]) * 0.01);
                        }
                    }
                }
                auto pct = [](std::vector<double>& v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
                printf("   stream p50 %.2f", pct(streamd, 0.5));
                if (gather) printf("   hand-off (ready - cluster's last publish) p50 %.2f p90 %.2f max %.2f   own publish -> ready p50 %.2f p90 %.2f us",
                                   pct(hand, 0.5), pct(hand, 0.9), pct(hand, 1.0), pct(own, 0.5), pct(own, 0.9));
                printf("\n");
            }
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        return stage_us;
    };
    for (int kib : {6, 16}) {                     // 6 KiB per wave = the qkv projection's 13 MB over 256 x 8 waves; 16: a longer producer
        const double base = run("no hand-off (stream only)", kib, 0, 0, 0);
        const double a = run("gather in XCD clusters, everybody alike", kib, 1, 0, 1);
        const double c = run("gather in consecutive-32 clusters (cross-XCD)", kib, 0, 0, 1);
        const double l0 = run("no hand-off, every other cluster streams 4x", kib, 1, 1, 0);
        const double l1 = run("gather in XCD clusters, every other one 4x", kib, 1, 1, 1);
        printf("   => in-launch hand-off adds %.2f (XCD clusters) / %.2f (cross-XCD) us per stage; next to a live stream %.2f us\n", a - base, c - base,
               l1 - l0);
    }
    return 0;
}
