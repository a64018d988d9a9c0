// ffb6d_amd/csrc/knn.hip -- batched exact K-nearest-neighbour search for gfx950 (MI355X).
//
// Replaces the reference's CPU kd-tree search
//   ffb6d/models/RandLA/utils/nearest_neighbors/knn_.cxx:104-135 (cpp_knn_batch_omp)
// whose semantics (nanoflann.hpp:79-145, 323-348) are: exact K-NN, ascending squared f32
// distance ((dx*dx+dy*dy)+dz*dz, no FMA), first-visited wins ties.  We scan the support
// set in ascending index order, so ties resolve to the lowest index.
//
// Design (wave64, LDS, no kd-tree -- a tree walk diverges per lane; a tiled scan does not):
//   * one lane owns QPT queries; support points are staged through LDS as float4 tiles
//     and read back as wave-uniform ds_read_b128 broadcasts (conflict free);
//   * the running top-K of a query lives in registers (K dists + K ids, fully unrolled);
//     the expensive sorted insert is taken off the hot loop: a candidate that beats the
//     lane's current K-th distance is only *appended* to a small per-lane LDS queue
//     (exec-masked ds_write_b64), and the wave drains all queues together when any lane's
//     queue is nearly full -- so the ~100-op insert runs with most lanes active instead
//     of once per candidate with one lane active;
//   * when B*ceil(Q/256) blocks cannot fill 256 CUs the support range is split across
//     blockIdx.y and the per-split sorted lists are merged by a second tiny kernel
//     (splits are merged in index order, preserving the lowest-index tie rule).
//
// Since knn_pruned.hip exists this scan only serves SMALL support sets (< 512 points since round 5, < 2048 before; larger ones
// are Morton-prepared and searched with tile pruning) and the host-pointer cpp_knn* entry points
// route through ffb6d_knn_batch_device, i.e. through whichever kernel fits the shape.
//
// Roofline: VALU-bound (8 f32 ops + compare per pair), algorithmic HBM traffic is ~10.7 MB/frame
// (SURVEY.md section 8d).
#include "common.h"

#include <cfloat>
#include <cmath>

namespace ffb6d {
namespace {

constexpr int KNN_BLOCK = 256;  // 4 waves
constexpr int KNN_TILE = 1024;  // support points per LDS tile (16 KiB as float4)
constexpr int KNN_GROUP = 8;    // candidates per unrolled group / between queue-occupancy checks
constexpr int KNN_QCAP = 16;    // queue slots per query (>= 2*GROUP)

template <int K>
struct TopK {
    float d[K];
    uint32_t i[K];

    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int k = 0; k < K; ++k) { d[k] = FLT_MAX; i[k] = 0u; }
    }
    // sorted insert; an equal distance goes BEHIND the entries already present
    // (nanoflann.hpp:118-135: shift only while dists[i-1] > dist)
    __device__ __forceinline__ void insert(float nd, uint32_t ni)
    {
#pragma unroll
        for (int k = K - 1; k >= 1; --k) {
            const bool gt_prev = d[k - 1] > nd;
            const bool gt_cur = d[k] > nd;
            d[k] = gt_prev ? d[k - 1] : (gt_cur ? nd : d[k]);
            i[k] = gt_prev ? i[k - 1] : (gt_cur ? ni : i[k]);
        }
        if (d[0] > nd) { d[0] = nd; i[0] = ni; }
    }
};

template <>
struct TopK<1> {
    float d[1];
    uint32_t i[1];
    __device__ __forceinline__ void init() { d[0] = FLT_MAX; i[0] = 0u; }
    __device__ __forceinline__ void insert(float nd, uint32_t ni)
    {
        if (nd < d[0]) { d[0] = nd; i[0] = ni; }
    }
};

// squared distance with the reference's operation order, each op rounded to f32
__device__ __forceinline__ float sqdist(float qx, float qy, float qz, const float4& p)
{
    const float dx = __fsub_rn(qx, p.x);
    const float dy = __fsub_rn(qy, p.y);
    const float dz = __fsub_rn(qz, p.z);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

template <int K>
__device__ __forceinline__ void store_result(const TopK<K>& top, size_t row, int Kout,
                                             int64_t* __restrict__ idx64,
                                             int32_t* __restrict__ idx32,
                                             float* __restrict__ dist)
{
    const size_t o = row * (size_t)Kout;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k < Kout) {
            if (idx64) idx64[o + k] = (int64_t)top.i[k];
            if (idx32) idx32[o + k] = (int32_t)top.i[k];
            if (dist) dist[o + k] = top.d[k];
        }
    }
}

// grid = (ceil(Q / (256*QPT)), nsplit, B)
template <int K, int QPT>
__device__ __forceinline__ void
knn_scan_body(const float* __restrict__ support, const float* __restrict__ query,
              int S, int Q, int nsplit, int chunk,
              float* __restrict__ part_d, uint32_t* __restrict__ part_i,
              int64_t* __restrict__ idx64, int32_t* __restrict__ idx32,
              float* __restrict__ dist, int Kout, const int bx, const int split, const int b)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* tile = reinterpret_cast<float4*>(smem);
    uint2* queue = reinterpret_cast<uint2*>(smem + KNN_TILE * sizeof(float4));

    const int tid = threadIdx.x;
    const int q0 = bx * (KNN_BLOCK * QPT);
    const float* sup = support + (size_t)b * S * 3;
    const float* qry = query + (size_t)b * Q * 3;
    const int s_begin = split * chunk;
    const int s_end = min(S, s_begin + chunk);

    float qx[QPT], qy[QPT], qz[QPT], worst[QPT];
    int cnt[QPT];
    TopK<K> top[QPT];
#pragma unroll
    for (int t = 0; t < QPT; ++t) {
        const int qi = min(q0 + t * KNN_BLOCK + tid, Q - 1);
        qx[t] = qry[(size_t)qi * 3 + 0];
        qy[t] = qry[(size_t)qi * 3 + 1];
        qz[t] = qry[(size_t)qi * 3 + 2];
        top[t].init();
        worst[t] = FLT_MAX;
        cnt[t] = 0;
    }

    auto drain = [&]() {
#pragma unroll
        for (int t = 0; t < QPT; ++t) {
            if constexpr (K > 1) {
                for (int it = 0; __any(it < cnt[t]); ++it) {
                    if (it < cnt[t]) {
                        const uint2 e = queue[(t * KNN_QCAP + it) * KNN_BLOCK + tid];
                        const float d = __uint_as_float(e.x);
                        if (d < top[t].d[K - 1]) top[t].insert(d, e.y);
                    }
                }
                cnt[t] = 0;
                worst[t] = top[t].d[K - 1];
            }
        }
    };

    for (int base = s_begin; base < s_end; base += KNN_TILE) {
        const int n = min(KNN_TILE, s_end - base);
        __syncthreads();  // every wave is done with the previous tile
#pragma unroll
        for (int j = tid; j < KNN_TILE; j += KNN_BLOCK) {
            // pad: d = inf, never kept; .w carries the global support index
            float4 p = make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float((uint32_t)(base + j)));
            if (j < n) {
                const float* s = sup + (size_t)(base + j) * 3;
                p.x = s[0]; p.y = s[1]; p.z = s[2];
            }
            tile[j] = p;
        }
        __syncthreads();

        const int n_pad = (n + KNN_GROUP - 1) / KNN_GROUP * KNN_GROUP;
        for (int j0 = 0; j0 < n_pad; j0 += KNN_GROUP) {
            // 1) all LDS reads of the group first (wave-uniform addresses: ds_read_b128
            //    broadcasts), 2) all distances, 3) the rare predicated queue appends.  Keeping
            //    the three phases apart lets the loads and the arithmetic of a group overlap
            //    instead of waiting for LDS once per candidate.
            float4 p[KNN_GROUP];
#pragma unroll
            for (int u = 0; u < KNN_GROUP; ++u) p[u] = tile[j0 + u];
            float d[QPT][KNN_GROUP];
#pragma unroll
            for (int u = 0; u < KNN_GROUP; ++u)
#pragma unroll
                for (int t = 0; t < QPT; ++t) d[t][u] = sqdist(qx[t], qy[t], qz[t], p[u]);
#pragma unroll
            for (int u = 0; u < KNN_GROUP; ++u) {
                const uint32_t sidx = __float_as_uint(p[u].w);   // global support index rides in .w
#pragma unroll
                for (int t = 0; t < QPT; ++t) {
                    if constexpr (K == 1) {
                        top[t].insert(d[t][u], sidx);
                    } else {
                        if (d[t][u] < worst[t]) {
                            queue[(t * KNN_QCAP + cnt[t]) * KNN_BLOCK + tid] =
                                make_uint2(__float_as_uint(d[t][u]), sidx);
                            cnt[t]++;
                        }
                    }
                }
            }
            if constexpr (K > 1) {
                bool need = false;
#pragma unroll
                for (int t = 0; t < QPT; ++t) need |= (cnt[t] > KNN_QCAP - KNN_GROUP);
                if (__any(need)) drain();
            }
        }
    }
    drain();

#pragma unroll
    for (int t = 0; t < QPT; ++t) {
        const int qi = q0 + t * KNN_BLOCK + tid;
        if (qi >= Q) continue;
        if (nsplit == 1) {
            store_result<K>(top[t], (size_t)b * Q + qi, Kout, idx64, idx32, dist);
        } else {
            const size_t o = (((size_t)b * nsplit + split) * Q + qi) * K;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                part_d[o + k] = top[t].d[k];
                part_i[o + k] = top[t].i[k];
            }
        }
    }
}

template <int K, int QPT>
__global__ void __launch_bounds__(KNN_BLOCK)
knn_scan_kernel(const float* __restrict__ support, const float* __restrict__ query,
                int S, int Q, int nsplit, int chunk,
                float* __restrict__ part_d, uint32_t* __restrict__ part_i,
                int64_t* __restrict__ idx64, int32_t* __restrict__ idx32,
                float* __restrict__ dist, int Kout)
{
    knn_scan_body<K, QPT>(support, query, S, Q, nsplit, chunk, part_d, part_i, idx64, idx32, dist, Kout, blockIdx.x, blockIdx.y, blockIdx.z);
}

// several scans in ONE launch (ffb6d_knn_search_multi): blockIdx.x walks the (search, frame, query block) triples of a table
// in the kernel arguments; every block scans its whole support (supports routed here are short: no split, no merge pass)
struct ScanArgs {
    const float *support, *query;
    int64_t* idx64;
    int32_t* idx32;
    float* dist;
    int S, Q, Kout, gx, blk0;
};
constexpr int MAX_SCANS = 16;
struct MultiScan {
    ScanArgs a[MAX_SCANS];
    int n;
};

template <int K, int QPT>
__global__ void __launch_bounds__(KNN_BLOCK)
knn_scan_multi_kernel(const MultiScan m)
{
    int sidx = 0;
#pragma unroll
    for (int i = 1; i < MAX_SCANS; ++i) sidx += (i < m.n && m.a[i].blk0 <= (int)blockIdx.x) ? 1 : 0;
    const ScanArgs& a = m.a[sidx];
    const int local = blockIdx.x - a.blk0;
    knn_scan_body<K, QPT>(a.support, a.query, a.S, a.Q, 1, a.S, nullptr, nullptr, a.idx64, a.idx32, a.dist, a.Kout, local % a.gx, 0,
                          local / a.gx);
}

// merges the nsplit sorted partial lists of a query, in split (= index) order
template <int K>
__global__ void __launch_bounds__(KNN_BLOCK)
knn_merge_kernel(const float* __restrict__ part_d, const uint32_t* __restrict__ part_i,
                 int Q, int nsplit, int64_t* __restrict__ idx64, int32_t* __restrict__ idx32,
                 float* __restrict__ dist, int Kout)
{
    const int qi = blockIdx.x * KNN_BLOCK + threadIdx.x;
    const int b = blockIdx.y;
    if (qi >= Q) return;
    TopK<K> top;
    top.init();
    for (int s = 0; s < nsplit; ++s) {
        const size_t o = (((size_t)b * nsplit + s) * Q + qi) * K;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float d = part_d[o + k];
            if (d < top.d[K - 1]) top.insert(d, part_i[o + k]);
        }
    }
    store_result<K>(top, (size_t)b * Q + qi, Kout, idx64, idx32, dist);
}

int pad_k(int K)
{
    int p = 1;
    while (p < K) p <<= 1;
    return p;
}

constexpr int qpt_for(int Kp) { return Kp == 1 ? 4 : (Kp <= 4 ? 2 : 1); }

struct Plan {
    int Kp, qpt, qblocks, nsplit, chunk;
};

Plan make_plan(int64_t B, int64_t S, int64_t Q, int K)
{
    Plan p;
    p.Kp = pad_k(K);
    p.qpt = qpt_for(p.Kp);
    p.qblocks = (int)ceil_div(Q, (int64_t)KNN_BLOCK * p.qpt);
    // aim at >= 4 blocks per CU (256 CUs); never split finer than 2 tiles per chunk
    const int64_t target = 1024;
    int64_t want = ceil_div(target, B * p.qblocks);
    int64_t max_split = ceil_div(S, 2 * (int64_t)KNN_TILE);
    int64_t ns = want < 1 ? 1 : want;
    if (ns > max_split) ns = max_split;
    if (ns > 64) ns = 64;
    if (ns < 1) ns = 1;
    int64_t chunk = ceil_div(S, ns);
    chunk = ceil_div(chunk, (int64_t)KNN_TILE) * KNN_TILE;  // whole tiles
    ns = ceil_div(S, chunk);
    p.nsplit = (int)ns;
    p.chunk = (int)chunk;
    return p;
}

template <int K, int QPT>
int launch_knn(const Plan& p, const float* support, const float* query, int64_t B, int64_t S,
               int64_t Q, int Kout, int64_t* idx64, int32_t* idx32, float* dist, void* ws,
               hipStream_t st)
{
    float* part_d = nullptr;
    uint32_t* part_i = nullptr;
    if (p.nsplit > 1) {
        const size_t n = (size_t)B * p.nsplit * Q * K;
        part_d = reinterpret_cast<float*>(ws);
        part_i = reinterpret_cast<uint32_t*>(part_d + n);
    }
    const size_t lds = KNN_TILE * sizeof(float4) +
                       (K > 1 ? (size_t)QPT * KNN_QCAP * KNN_BLOCK * sizeof(uint2) : 0);
    dim3 grid(p.qblocks, p.nsplit, (unsigned)B);
    hipLaunchKernelGGL((knn_scan_kernel<K, QPT>), grid, dim3(KNN_BLOCK), lds, st, support, query,
                       (int)S, (int)Q, p.nsplit, p.chunk, part_d, part_i, idx64, idx32, dist, Kout);
    FFB6D_LAUNCH_CHECK();
    if (p.nsplit > 1) {
        dim3 mgrid((unsigned)ceil_div(Q, KNN_BLOCK), (unsigned)B);
        hipLaunchKernelGGL((knn_merge_kernel<K>), mgrid, dim3(KNN_BLOCK), 0, st, part_d, part_i,
                           (int)Q, p.nsplit, idx64, idx32, dist, Kout);
        FFB6D_LAUNCH_CHECK();
    }
    return FFB6D_OK;
}

int check_shape(int64_t B, int64_t S, int64_t Q, int64_t dim, int64_t K)
{
    FFB6D_REQUIRE(dim == 3, "knn: dim must be 3 (got %lld)", (long long)dim);
    FFB6D_REQUIRE(K >= 1 && K <= 32, "knn: K must be in [1,32] (got %lld)", (long long)K);
    FFB6D_REQUIRE(B >= 0 && Q >= 0 && S >= 0, "knn: negative size");
    FFB6D_REQUIRE(S >= K || B == 0 || Q == 0,
                  "knn: npts (%lld) < K (%lld): the reference leaves such rows undefined",
                  (long long)S, (long long)K);
    FFB6D_REQUIRE(S < (1LL << 31) && Q < (1LL << 31) && B < 65536, "knn: size too large");
    return FFB6D_OK;
}

// host-pointer path shared by the four cpp_knn* entry points
void knn_host(const char* who, const float* pts, size_t B, size_t npts, size_t dim,
              const float* queries, size_t nq, size_t K, long* out)
{
    if (B == 0 || nq == 0 || K == 0) return;
    int rc = check_shape((int64_t)B, (int64_t)npts, (int64_t)nq, (int64_t)dim, (int64_t)K);
    float *d_s = nullptr, *d_q = nullptr;
    int64_t* d_i = nullptr;
    void* d_ws = nullptr;
    auto fail = [&](const char* what, hipError_t e) {
        set_error(FFB6D_ERR_HIP, "%s: %s failed: %s", who, what, hipGetErrorString(e));
        rc = FFB6D_ERR_HIP;
    };
    if (rc == FFB6D_OK) {
        const size_t sb = B * npts * 3 * sizeof(float), qb = B * nq * 3 * sizeof(float);
        const size_t ib = B * nq * K * sizeof(int64_t);
        const size_t wb = ffb6d_knn_workspace_bytes((int64_t)B, (int64_t)npts, (int64_t)nq, (int)K);
        hipError_t e;
        if ((e = hipMalloc(&d_s, sb)) != hipSuccess) fail("hipMalloc", e);
        else if ((e = hipMalloc(&d_q, qb)) != hipSuccess) fail("hipMalloc", e);
        else if ((e = hipMalloc(&d_i, ib)) != hipSuccess) fail("hipMalloc", e);
        else if (wb && (e = hipMalloc(&d_ws, wb)) != hipSuccess) fail("hipMalloc", e);
        else if ((e = hipMemcpy(d_s, pts, sb, hipMemcpyHostToDevice)) != hipSuccess) fail("H2D", e);
        else if ((e = hipMemcpy(d_q, queries, qb, hipMemcpyHostToDevice)) != hipSuccess) fail("H2D", e);
        else {
            rc = ffb6d_knn_batch_device(d_s, d_q, (int64_t)B, (int64_t)npts, (int64_t)nq, (int)K,
                                        d_i, nullptr, nullptr, d_ws, wb, nullptr);
            if (rc == FFB6D_OK) {
                static_assert(sizeof(long) == sizeof(int64_t), "LP64 expected");
                if ((e = hipMemcpy(out, d_i, ib, hipMemcpyDeviceToHost)) != hipSuccess) fail("D2H", e);
            }
        }
    }
    if (d_s) (void)hipFree(d_s);
    if (d_q) (void)hipFree(d_q);
    if (d_i) (void)hipFree(d_i);
    if (d_ws) (void)hipFree(d_ws);
    if (rc != FFB6D_OK)  // the reference signature is void: be loud, leave `out` untouched
        fprintf(stderr, "[ffb6d_amd] %s failed: %s\n", who, ffb6d_last_error());
}

template <int K>
int launch_scan_multi(const MultiScan& m, int blocks, hipStream_t st)
{
    constexpr int QPT = qpt_for(K);
    const size_t lds = KNN_TILE * sizeof(float4) + (K > 1 ? (size_t)QPT * KNN_QCAP * KNN_BLOCK * sizeof(uint2) : 0);
    hipLaunchKernelGGL((knn_scan_multi_kernel<K, QPT>), dim3((unsigned)blocks), dim3(KNN_BLOCK), lds, st, m);
    FFB6D_LAUNCH_CHECK();
    return FFB6D_OK;
}

}  // namespace

int knn_search_multi_prepared(const ffb6d_knn_search_t* s, const int* which, int n, int64_t B, hipStream_t st);   // knn_pruned.hip
}  // namespace ffb6d

using namespace ffb6d;

extern "C" int ffb6d_knn_search_multi(int n, const ffb6d_knn_search_t* s, int64_t B, ffb6d_stream_t stream)
{
    FFB6D_REQUIRE(n >= 0 && (n == 0 || s) && B >= 1, "knn_search_multi: bad arguments");
    if (n == 0) return FFB6D_OK;
    FFB6D_REQUIRE(n <= 64, "knn_search_multi: at most 64 searches per call (got %d)", n);
    hipStream_t st = as_stream(stream);
    int pruned[64], np = 0;
    for (int i = 0; i < n; ++i) {
        const int rc = check_shape(B, s[i].S, s[i].Q, 3, s[i].K);
        if (rc != FFB6D_OK) return rc;
        FFB6D_REQUIRE(s[i].Q >= 1 && (s[i].idx64 || s[i].idx32 || s[i].dist), "knn_search_multi: search %d has no queries or no output", i);
        if (ffb6d_knn_uses_pruning(B, s[i].S, s[i].Q, s[i].K)) {
            FFB6D_REQUIRE(s[i].K <= 16, "knn_search_multi: K > 16 on a long support is not batched (search %d): call ffb6d_knn_search_prepared", i);
            pruned[np++] = i;
        } else {
            FFB6D_REQUIRE(s[i].support && s[i].query, "knn_search_multi: search %d runs the scan kernel and needs the raw arrays", i);
        }
    }
    // (Round 5 measured the four kernels of a batch on library-owned side streams, forked from and joined into the caller's stream: the
    // K = 16 scans of a pyramid are 32 workgroups, 91 us with 224 CUs idle.  Alone the pyramid went from 923 to 901 us -- the two big
    // kernels share the chip either way -- and the bench step from 20.4 to 28.1 ms: six streams on the runtime's four hardware queues
    // put the colour branch and the point branch behind one another.  No hidden streams.)
    int rc = FFB6D_OK;
    const int kps[6] = {1, 2, 4, 8, 16, 32};
    for (int kp : kps) {
        MultiScan m;
        m.n = 0;
        int blocks = 0;
        auto flush = [&]() -> int {
            if (m.n == 0) return FFB6D_OK;
            int r = FFB6D_OK;
            hipStream_t ls = st;
            switch (kp) {
                case 1: r = launch_scan_multi<1>(m, blocks, ls); break;
                case 2: r = launch_scan_multi<2>(m, blocks, ls); break;
                case 4: r = launch_scan_multi<4>(m, blocks, ls); break;
                case 8: r = launch_scan_multi<8>(m, blocks, ls); break;
                case 16: r = launch_scan_multi<16>(m, blocks, ls); break;
                default: r = launch_scan_multi<32>(m, blocks, ls); break;
            }
            m.n = 0;
            blocks = 0;
            return r;
        };
        for (int i = 0; i < n && rc == FFB6D_OK; ++i) {
            if (ffb6d_knn_uses_pruning(B, s[i].S, s[i].Q, s[i].K) || pad_k(s[i].K) != kp) continue;
            ScanArgs& a = m.a[m.n];
            a.support = s[i].support; a.query = s[i].query; a.idx64 = s[i].idx64; a.idx32 = s[i].idx32; a.dist = s[i].dist;
            a.S = (int)s[i].S; a.Q = (int)s[i].Q; a.Kout = s[i].K;
            a.gx = (int)ceil_div(s[i].Q, (int64_t)KNN_BLOCK * qpt_for(kp));
            a.blk0 = blocks;
            blocks += a.gx * (int)B;
            if (++m.n == MAX_SCANS) rc = flush();
        }
        if (rc == FFB6D_OK) rc = flush();
        if (rc != FFB6D_OK) break;
    }
    if (rc == FFB6D_OK) rc = knn_search_multi_prepared(s, pruned, np, B, st);
    return rc;
}

extern "C" {

int ffb6d_knn_uses_pruning(int64_t B, int64_t S, int64_t Q, int K)
{
    // sorting + tile boxes pay off once the support set is a few tiles long; tiny sets are
    // cheaper to scan outright.  Round 5 measured the threshold on the 22 searches of a pyramid (bs = 8): 2048 -> 512 moves the
    // 768-point supports (five searches, among them 76 800 grid queries against 768 points: 148 -> 127 us, and the K = 16 self search
    // of level 2: 108 -> 35 us) to the Morton-ordered kernels: pyramid alone 908 -> 808 us, same indices; 256 gains nothing more.
    return (B >= 1 && Q >= 1 && K >= 1 && K <= 32 && S >= 512) ? 1 : 0;
}

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

size_t ffb6d_knn_workspace_bytes(int64_t B, int64_t S, int64_t Q, int K)
{
    if (B <= 0 || S <= 0 || Q <= 0 || K < 1 || K > 32) return 0;
    if (ffb6d_knn_uses_pruning(B, S, Q, K)) {
        return align256(ffb6d_knn_prepared_bytes(B, S)) + align256(ffb6d_knn_prepared_bytes(B, Q)) +
               align256(ffb6d_knn_prepare_workspace_bytes(B, S > Q ? S : Q));
    }
    const Plan p = make_plan(B, S, Q, K);
    if (p.nsplit <= 1) return 0;
    return (size_t)B * p.nsplit * Q * p.Kp * (sizeof(float) + sizeof(uint32_t));
}

int ffb6d_knn_batch_device(const float* support, const float* query, int64_t B, int64_t S,
                           int64_t Q, int K, int64_t* idx64, int32_t* idx32, float* dist,
                           void* workspace, size_t workspace_bytes, ffb6d_stream_t stream)
{
    int rc = check_shape(B, S, Q, 3, K);
    if (rc != FFB6D_OK) return rc;
    if (B == 0 || Q == 0) return FFB6D_OK;
    FFB6D_REQUIRE(support && query, "knn: null input pointer");
    FFB6D_REQUIRE(idx64 || idx32 || dist, "knn: no output requested");
    const size_t need = ffb6d_knn_workspace_bytes(B, S, Q, K);
    if (need > 0 && (workspace == nullptr || workspace_bytes < need))
        return set_error(FFB6D_ERR_WORKSPACE, "knn: workspace of %zu bytes required, got %zu",
                         need, workspace ? workspace_bytes : (size_t)0);
    if (ffb6d_knn_uses_pruning(B, S, Q, K)) {
        // Morton-sort both sets, then the tile-pruned search (knn_pruned.hip)
        char* ws = static_cast<char*>(workspace);
        const size_t ps = align256(ffb6d_knn_prepared_bytes(B, S)), pq = align256(ffb6d_knn_prepared_bytes(B, Q));
        void* prep_s = ws;
        void* prep_q = ws + ps;
        void* scratch = ws + ps + pq;
        const size_t scratch_bytes = workspace_bytes - ps - pq;
        rc = ffb6d_knn_prepare(support, B, S, prep_s, ps, scratch, scratch_bytes, stream);
        if (rc != FFB6D_OK) return rc;
        if (query == support && Q == S) {
            prep_q = prep_s;
        } else {
            rc = ffb6d_knn_prepare(query, B, Q, prep_q, pq, scratch, scratch_bytes, stream);
            if (rc != FFB6D_OK) return rc;
        }
        return ffb6d_knn_search_prepared(prep_s, prep_q, nullptr, B, S, Q, K, idx64, idx32, dist, stream);
    }
    const Plan p = make_plan(B, S, Q, K);
    hipStream_t st = as_stream(stream);
    switch (p.Kp) {
        case 1:  return launch_knn<1, qpt_for(1)>(p, support, query, B, S, Q, K, idx64, idx32, dist, workspace, st);
        case 2:  return launch_knn<2, qpt_for(2)>(p, support, query, B, S, Q, K, idx64, idx32, dist, workspace, st);
        case 4:  return launch_knn<4, qpt_for(4)>(p, support, query, B, S, Q, K, idx64, idx32, dist, workspace, st);
        case 8:  return launch_knn<8, qpt_for(8)>(p, support, query, B, S, Q, K, idx64, idx32, dist, workspace, st);
        case 16: return launch_knn<16, qpt_for(16)>(p, support, query, B, S, Q, K, idx64, idx32, dist, workspace, st);
        case 32: return launch_knn<32, qpt_for(32)>(p, support, query, B, S, Q, K, idx64, idx32, dist, workspace, st);
    }
    return set_error(FFB6D_ERR_ARG, "knn: unsupported K=%d", K);
}

void cpp_knn(const float* points, const size_t npts, const size_t dim, const float* queries,
             const size_t nqueries, const size_t K, long* indices)
{
    knn_host("cpp_knn", points, 1, npts, dim, queries, nqueries, K, indices);
}

void cpp_knn_omp(const float* points, const size_t npts, const size_t dim, const float* queries,
                 const size_t nqueries, const size_t K, long* indices)
{
    knn_host("cpp_knn_omp", points, 1, npts, dim, queries, nqueries, K, indices);
}

void cpp_knn_batch(const float* batch_data, const size_t batch_size, const size_t npts,
                   const size_t dim, const float* queries, const size_t nqueries, const size_t K,
                   long* batch_indices)
{
    knn_host("cpp_knn_batch", batch_data, batch_size, npts, dim, queries, nqueries, K, batch_indices);
}

void cpp_knn_batch_omp(const float* batch_data, const size_t batch_size, const size_t npts,
                       const size_t dim, const float* queries, const size_t nqueries,
                       const size_t K, long* batch_indices)
{
    knn_host("cpp_knn_batch_omp", batch_data, batch_size, npts, dim, queries, nqueries, K,
             batch_indices);
}

}  // extern "C"
