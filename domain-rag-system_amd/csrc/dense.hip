// Exact dense kNN over a resident [N x D] fp32 matrix: fp32 MFMA tile GEMM fused with a
// per-query running top-K, then a per-query merge.  gfx950 (MI355X) only.
//
// Replaces the embedded Qdrant store and its brute-force query_points call behind
// VectorIndexer.search (reference src/indexing/vector_indexer.py:165-230).
//
// Layout in HBM: rows[N][D] fp32 row-major (already L2-normalised for cosine),
// row_norm2[N] fp32 only for the Euclidean metric.
//
// Kernel 1 (dense_scan_kernel): the grid is (query blocks) x (row strips).  A workgroup
// keeps BQ queries against one strip of rows: per 128-row tile it accumulates the
// [rows x queries] score tile with v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains) from
// LDS-staged, double-buffered k-slices, and in the tile epilogue appends every score that
// reaches the query's running threshold to that query's candidate list.  When a list is
// about to overflow, one wave radix-selects its K best, which also raises the threshold.
// Kernel 2 (merge_topk_kernel, in merge.h): per query, selects and sorts the K best of
// the per-strip lists.
#include "common.h"
#include "merge.h"

#include <math.h>
#include <type_traits>
#include <utility>
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 32;          // k-slice per stage
constexpr int LDK = BK + 4;     // padded LDS row (floats): conflict-free ds_read_b128
constexpr int BN_MAX = 256;     // rows per tile: 128 or 256, by tile configuration

struct DenseScanParams {
    const float* rows;          // [n_rows][dim] fp32, or the split image of the same size (see split_rows_kernel)
    const float* queries;       // [n_queries][dim], likewise
    const float* row_norm2;     // [n_rows] or null (Euclidean only)
    const uint8_t* row_mask;    // [n_rows] or null
    int64_t n_rows;
    int dim;
    int n_queries;
    int k;                      // wanted per query
    int kpub;                   // entries a strip may publish per query (the stride of cand): 2 k, so that the
                                // last tile rarely has to cut a list; in the threshold pre-pass the groups of a strip
    int cap;                    // candidate list capacity per (block, query), >= k + rows per tile
    int n_strips;               // multiple of 8
    int n_qblocks;
    int64_t strip_rows;         // multiple of the rows per tile
    uint2* scratch;             // [n_blocks][BQ][cap]  (key, local row)
    uint2* cand;                // [n_queries][n_strips][k]
    int32_t* cand_count;        // [n_queries][n_strips]
    uint32_t* tau_shared;       // [n_queries] key of a lower bound of each query's final K-th score
    uint32_t* fault;            // RAG_FAULT_* bits of this device (may be null)
    int euclid;
    int split;                  // rows and queries are split images
    int debug_row_mod;          // tuning aid: when > 0, row loads wrap at this many rows (results meaningless)
    int maxima;                 // threshold pre-pass: publish the largest score of every 16-row group, keep no lists
};

template <int WQ, int TQ, int WR, int TR>
struct DenseCfg {
    static constexpr int NW = WQ * WR;
    static constexpr int NT = NW * 64;
    static constexpr int BQ = WQ * TQ * 32;
    static constexpr int BN = WR * TR * 32;
    static_assert(BN == 128 || BN == BN_MAX, "row tile of 128 or 256 rows");
    static constexpr int E_LOADS = BN * (BK / 4) / NT;
    static constexpr int Q_LOADS = (BQ * (BK / 4) + NT - 1) / NT;
    static constexpr size_t LDS_BYTES =
        (size_t)2 * (BN + BQ) * LDK * sizeof(float) + BQ * 8 + NW * RAG_SEL_LDS_WORDS * 4;
};

// One wave: keep the K best of the n candidates of one query (in place), return the
// smallest kept key.
__device__ inline uint32_t wave_compact(uint2* list, int n, int K, uint32_t* sel_lds, int lane) {
    auto get = [&](int i, uint32_t& key, uint32_t& row) {
        uint2 v = list[i];
        key = v.x;
        row = v.y;
    };
    SelThr<uint32_t> thr = radix_select<uint32_t, 64>(get, n, K, sel_lds, lane);
    int w = 0;
    uint32_t kmin = 0xffffffffu;
    for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        uint2 v = make_uint2(0, 0);
        bool keep = false;
        if (i < n) {
            v = list[i];
            keep = sel_pass(thr, v.x, v.y);
        }
        unsigned long long m = __ballot(keep);
        if (keep) {
            int pos = w + __popcll(m & ((1ull << lane) - 1ull));
            list[pos] = v;
            kmin = min(kmin, v.x);
        }
        w += __popcll(m);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    return kmin;
}

// End of a row tile: every score of the [rows x queries] accumulator tile that reaches its query's running
// threshold joins that query's candidate list; lists that could overflow in the next tile are cut back to their
// K best (which raises the threshold); the accumulators are cleared.  Called by all waves of the workgroup
// (holds one barrier; the caller adds one behind it).
template <typename Cfg, bool EUCLID, int TQ, int TR>
__device__ __forceinline__ void tile_epilogue(const DenseScanParams& p, f32x16 (&acc)[TQ][TR], int tile, int n_tiles,
                                              int64_t row_begin, int64_t row_end, int q0, int wq, int wr, int wave,
                                              int lane, int strip, int* cnt_s, float* tau_s, uint32_t* sel_s,
                                              uint2* my_scratch) {
    constexpr int BN = Cfg::BN, BQ = Cfg::BQ, NW = Cfg::NW;
    const int lrow = lane & 31;
    // ---- tile epilogue: D[i][j], i = row (register + lane half), j = query (lane & 31)
    const int64_t trow0 = row_begin + (int64_t)tile * BN;
    const int rows_in_tile = (int)min((int64_t)BN, row_end - trow0);
    const int rbase = wr * TR * 32 + 4 * (lane >> 5);
    if (p.maxima) {
        // Threshold pre-pass.  A lane holds 16 rows of a sub-tile for its query: the largest of them is the score
        // of one real row, and the groups are disjoint, so the K-th largest of a query's group maxima is reached
        // by K rows — a lower bound of its final K-th score, found without keeping any list.
#pragma unroll
        for (int tq = 0; tq < TQ; ++tq) {
            const int ql = (wq * TQ + tq) * 32 + lrow;
#pragma unroll
            for (int tr = 0; tr < TR; ++tr) {
                float m = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = rbase + tr * 32 + (r & 3) + 8 * (r >> 2);
                    float v = acc[tq][tr][r];
                    if (EUCLID && rl < rows_in_tile) v = 2.0f * v - p.row_norm2[trow0 + rl];
                    if (rl < rows_in_tile) m = fmaxf(m, v);
                    acc[tq][tr][r] = 0.f;
                }
                if (q0 + ql < p.n_queries) {
                    const int g = tile * (BN / 16) + (wr * TR + tr) * 2 + (lane >> 5);
                    p.cand[((size_t)(q0 + ql) * p.n_strips + strip) * p.kpub + g] =
                        make_uint2(f32_to_key(m), (uint32_t)(trow0 + rbase + tr * 32));
                }
            }
        }
        __syncthreads();
        return;
    }
#pragma unroll
    for (int tq = 0; tq < TQ; ++tq) {
        const int ql = (wq * TQ + tq) * 32 + lrow;
        const bool qvalid = q0 + ql < p.n_queries && p.debug_row_mod >= 0;
        const float tauq = tau_s[ql];
        uint2* list = my_scratch + (size_t)ql * p.cap;
        if (EUCLID) {
#pragma unroll
            for (int tr = 0; tr < TR; ++tr)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = rbase + tr * 32 + (r & 3) + 8 * (r >> 2);
                    if (rl < rows_in_tile) acc[tq][tr][r] = 2.0f * acc[tq][tr][r] - p.row_norm2[trow0 + rl];
                }
        }
        if (p.row_mask == nullptr) {
            // a lane counts what it will append from its column of all TR sub-tiles, reserves that many
            // places with one LDS atomic and then only stores: no round trip per candidate
            int n = 0;
#pragma unroll
            for (int tr = 0; tr < TR; ++tr)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = rbase + tr * 32 + (r & 3) + 8 * (r >> 2);
                    n += (rl < rows_in_tile && acc[tq][tr][r] >= tauq) ? 1 : 0;
                }
            if (!qvalid) n = 0;
            if (__ballot(n > 0)) {
                int slot = n > 0 ? atomicAdd(&cnt_s[ql], n) : 0;
                if (n > 0 && slot + n > p.cap) rag_flag(p.fault, RAG_FAULT_DENSE_LIST);
#pragma unroll
                for (int tr = 0; tr < TR; ++tr)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = rbase + tr * 32 + (r & 3) + 8 * (r >> 2);
                        const float sc = acc[tq][tr][r];
                        if (n > 0 && rl < rows_in_tile && sc >= tauq) {
                            if (slot < p.cap) list[slot] = make_uint2(f32_to_key(sc), (uint32_t)(trow0 + rl));
                            ++slot;
                        }
                    }
            }
        } else {
            // with a row mask a candidate is known only after its mask byte is read: one place at a time
#pragma unroll
            for (int tr = 0; tr < TR; ++tr) {
                const f32x16& a = acc[tq][tr];
                float m = fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(a[4], a[5]), fmaxf(a[6], a[7])));
                m = fmaxf(m, fmaxf(fmaxf(fmaxf(a[8], a[9]), fmaxf(a[10], a[11])), fmaxf(fmaxf(a[12], a[13]), fmaxf(a[14], a[15]))));
                const bool any = qvalid && m >= tauq;
                if (__ballot(any)) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = rbase + tr * 32 + (r & 3) + 8 * (r >> 2);
                        const float sc = a[r];
                        if (any && rl < rows_in_tile && sc >= tauq) {
                            const uint32_t grow = (uint32_t)(trow0 + rl);
                            if (p.row_mask[grow]) {
                                int slot = atomicAdd(&cnt_s[ql], 1);
                                if (slot < p.cap)
                                    list[slot] = make_uint2(f32_to_key(sc), grow);
                                else
                                    rag_flag(p.fault, RAG_FAULT_DENSE_LIST);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int tr = 0; tr < TR; ++tr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tq][tr][r] = 0.f;
    }
    __syncthreads();
    // ---- lists that could overflow in the next tile are cut back to their K best
    const bool last_tile = tile == n_tiles - 1;
    for (int base = wave * (BQ / NW); base < (wave + 1) * (BQ / NW); base += 64) {
        int ql = base + lane;
        bool need = false;
        if (ql < (wave + 1) * (BQ / NW)) {
            int c = cnt_s[ql];
            need = last_tile ? c > p.kpub : c > p.cap - BN;
        }
        unsigned long long m = __ballot(need);
        while (m) {
            int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            int q = base + b;
            int n = min(cnt_s[q], p.cap);
            uint32_t kmin = wave_compact(my_scratch + (size_t)q * p.cap, n, p.k,
                                         sel_s + wave * RAG_SEL_LDS_WORDS, lane);
            if (lane == 0) {
                cnt_s[q] = p.k;
                tau_s[q] = key_to_f32(kmin);
                // K rows of this strip reach kmin, so the final K-th score does too
                if (q0 + q < p.n_queries) atomicMax(&p.tau_shared[q0 + q], kmin);
            }
        }
        // adopt the best lower bound any strip has published for this query
        if (!last_tile && ql < (wave + 1) * (BQ / NW) && q0 + ql < p.n_queries) {
            uint32_t g = __hip_atomic_load(&p.tau_shared[q0 + ql], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float tg = key_to_f32(g);
            if (tg > tau_s[ql]) tau_s[ql] = tg;
        }
    }
}

// End of a strip: the workgroup's lists (cut to K entries by the last tile's epilogue) go to their places in
// cand; in the threshold pre-pass the group maxima are there already and only their number is recorded.
template <typename Cfg>
__device__ __forceinline__ void publish_strip(const DenseScanParams& p, int n_tiles, int q0, int wave, int lane, int strip,
                                              const int* cnt_s, const uint2* my_scratch) {
    constexpr int BQ = Cfg::BQ, NW = Cfg::NW;
    for (int ql = wave * (BQ / NW); ql < (wave + 1) * (BQ / NW); ++ql) {
        const int q = q0 + ql;
        if (q >= p.n_queries) break;
        if (p.maxima) {
            if (lane == 0) p.cand_count[(size_t)q * p.n_strips + strip] = n_tiles * (Cfg::BN / 16);
            continue;
        }
        const int n = min(cnt_s[ql], p.kpub);
        uint2* dst = p.cand + ((size_t)q * p.n_strips + strip) * p.kpub;
        const uint2* src = my_scratch + (size_t)ql * p.cap;
        for (int i = lane; i < n; i += 64) dst[i] = src[i];
        if (lane == 0) p.cand_count[(size_t)q * p.n_strips + strip] = n;
    }
}

// SPLIT: rows and queries are split images: each 16 values of a row (64 bytes either way) are held as their 16
// bf16 high parts, then their 16 bf16 low parts (x = hi + lo up to 2^-17 relative).  The
// score tile is then hi.hi + hi.lo + lo.hi on the bf16 matrix cores with fp32 accumulation: three
// v_mfma_f32_32x32x16_bf16 per 16 values of k instead of eight v_mfma_f32_32x32x2_f32, at a fifth of the
// matrix time, with the same loads, LDS image and epilogue.  What is dropped (lo.lo) is below 2^-16 of
// |x||q| in the worst case and about 1e-7 for unit vectors.
template <int WQ, int TQ, int WR, int TR, bool EUCLID, bool SPLIT>
__global__ __launch_bounds__(WQ * WR * 64, 2) void dense_scan_kernel(DenseScanParams p) {
    using Cfg = DenseCfg<WQ, TQ, WR, TR>;
    constexpr int NT = Cfg::NT, BQ = Cfg::BQ, BN = Cfg::BN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* Es = reinterpret_cast<float*>(smem_raw);                 // [2][BN][LDK]
    float* Qs = Es + 2 * BN * LDK;                                  // [2][BQ][LDK]
    int* cnt_s = reinterpret_cast<int*>(Qs + 2 * BQ * LDK);         // [BQ]
    float* tau_s = reinterpret_cast<float*>(cnt_s + BQ);            // [BQ]
    uint32_t* sel_s = reinterpret_cast<uint32_t*>(tau_s + BQ);      // [NW][RAG_SEL_LDS_WORDS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wq = wave / WR, wr = wave % WR;

    // blocks that share a strip get the same blockIdx % 8 (one XCD's L2 under round-robin
    // placement: a speed choice only)
    const int id = blockIdx.x;
    const int xcd = id & 7;
    const int j = id >> 3;
    const int qb = j % p.n_qblocks;
    const int strip = (j / p.n_qblocks) * 8 + xcd;
    const int q0 = qb * BQ;

    const int64_t row_begin = (int64_t)strip * p.strip_rows;
    const int64_t row_end = min(p.n_rows, row_begin + p.strip_rows);
    uint2* my_scratch = p.scratch + (size_t)blockIdx.x * BQ * p.cap;

    // the running threshold starts from the best lower bound known for the query (-inf, or what a
    // pre-pass over a sample of rows / other strips have published)
    for (int i = tid; i < BQ; i += NT) {
        cnt_s[i] = 0;
        tau_s[i] = q0 + i < p.n_queries
                       ? key_to_f32(__hip_atomic_load(&p.tau_shared[q0 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                       : -INFINITY;
    }
    __syncthreads();

    const int n_tiles = row_begin < row_end ? (int)((row_end - row_begin + BN - 1) / BN) : 0;
    const int k_stages = (p.dim + BK - 1) / BK;
    const int n_stages = n_tiles * k_stages;

    float4 e_reg[Cfg::E_LOADS];
    float4 q_reg[Cfg::Q_LOADS];

    auto stage_load = [&](int st) {
        const int tile = st / k_stages;
        const int kb = (st - tile * k_stages) * BK;
        const int64_t trow0 = row_begin + (int64_t)tile * BN;
#pragma unroll
        for (int i = 0; i < Cfg::E_LOADS; ++i) {
            int idx = tid + i * NT;
            int r = idx >> 3, k4 = (idx & 7) * 4;
            int64_t grow = trow0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (grow < row_end && kb + k4 < p.dim)
                v = *reinterpret_cast<const float4*>(p.rows + (p.debug_row_mod > 0 ? grow % p.debug_row_mod : grow) * p.dim + kb + k4);
            e_reg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < Cfg::Q_LOADS; ++i) {
            int idx = tid + i * NT;
            int r = idx >> 3, k4 = (idx & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < BQ && q0 + r < p.n_queries && kb + k4 < p.dim)
                v = *reinterpret_cast<const float4*>(p.queries + (int64_t)(q0 + r) * p.dim + kb + k4);
            q_reg[i] = v;
        }
    };
    auto stage_store = [&](int buf) {
        float* eb = Es + buf * BN * LDK;
        float* qbuf = Qs + buf * BQ * LDK;
#pragma unroll
        for (int i = 0; i < Cfg::E_LOADS; ++i) {
            int idx = tid + i * NT;
            int r = idx >> 3, k4 = (idx & 7) * 4;
            *reinterpret_cast<float4*>(eb + r * LDK + k4) = e_reg[i];
        }
#pragma unroll
        for (int i = 0; i < Cfg::Q_LOADS; ++i) {
            int idx = tid + i * NT;
            int r = idx >> 3, k4 = (idx & 7) * 4;
            if (r < BQ) *reinterpret_cast<float4*>(qbuf + r * LDK + k4) = q_reg[i];
        }
    };

    f32x16 acc[TQ][TR];
#pragma unroll
    for (int a = 0; a < TQ; ++a)
#pragma unroll
        for (int b = 0; b < TR; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (n_stages > 0) {
        stage_load(0);
        stage_store(0);
    }
    __syncthreads();

    const int lrow = lane & 31;           // row of the A tile / column (query) of the B tile
    const int khalf = (lane >> 5) * 4;    // which half of each 8-wide k group this lane feeds

    for (int st = 0; st < n_stages; ++st) {
        const int buf = st & 1;
        if (st + 1 < n_stages) stage_load(st + 1);

        const float* eb = Es + buf * BN * LDK;
        const float* qbuf = Qs + buf * BQ * LDK;
        if constexpr (SPLIT) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // lane (row, half) feeds k = 16s + 8*half + 0..7 of the slice: 16 bytes of the high parts,
                // 16 bytes of the low parts 32 bytes further on
                bf16x8 ah[TR], al[TR], bh[TQ], bl[TQ];
#pragma unroll
                for (int tr = 0; tr < TR; ++tr) {
                    const float* src = eb + ((wr * TR + tr) * 32 + lrow) * LDK + 16 * s + khalf;
                    ah[tr] = *reinterpret_cast<const bf16x8*>(src);
                    al[tr] = *reinterpret_cast<const bf16x8*>(src + 8);
                }
#pragma unroll
                for (int tq = 0; tq < TQ; ++tq) {
                    const float* src = qbuf + ((wq * TQ + tq) * 32 + lrow) * LDK + 16 * s + khalf;
                    bh[tq] = *reinterpret_cast<const bf16x8*>(src);
                    bl[tq] = *reinterpret_cast<const bf16x8*>(src + 8);
                }
#pragma unroll
                for (int tq = 0; tq < TQ; ++tq)
#pragma unroll
                    for (int tr = 0; tr < TR; ++tr)
                        acc[tq][tr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tr], bh[tq], acc[tq][tr], 0, 0, 0);
#pragma unroll
                for (int tq = 0; tq < TQ; ++tq)
#pragma unroll
                    for (int tr = 0; tr < TR; ++tr)
                        acc[tq][tr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tr], bl[tq], acc[tq][tr], 0, 0, 0);
#pragma unroll
                for (int tq = 0; tq < TQ; ++tq)
#pragma unroll
                    for (int tr = 0; tr < TR; ++tr)
                        acc[tq][tr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tr], bh[tq], acc[tq][tr], 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            float4 a[TR], b[TQ];
#pragma unroll
            for (int tr = 0; tr < TR; ++tr)
                a[tr] = *reinterpret_cast<const float4*>(eb + ((wr * TR + tr) * 32 + lrow) * LDK + kk * 8 + khalf);
#pragma unroll
            for (int tq = 0; tq < TQ; ++tq)
                b[tq] = *reinterpret_cast<const float4*>(qbuf + ((wq * TQ + tq) * 32 + lrow) * LDK + kk * 8 + khalf);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int tq = 0; tq < TQ; ++tq)
#pragma unroll
                    for (int tr = 0; tr < TR; ++tr) {
                        float av = c == 0 ? a[tr].x : c == 1 ? a[tr].y : c == 2 ? a[tr].z : a[tr].w;
                        float bv = c == 0 ? b[tq].x : c == 1 ? b[tq].y : c == 2 ? b[tq].z : b[tq].w;
                        acc[tq][tr] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tq][tr], 0, 0, 0);
                    }
            }
        }
        }

        const int tile = st / k_stages;
        if (st - tile * k_stages == k_stages - 1) {
            tile_epilogue<Cfg, EUCLID, TQ, TR>(p, acc, tile, n_tiles, row_begin, row_end, q0, wq, wr, wave, lane, strip, cnt_s, tau_s,
                                               sel_s, my_scratch);
            __syncthreads();
        }

        if (st + 1 < n_stages) stage_store(buf ^ 1);
        __syncthreads();
    }

    publish_strip<Cfg>(p, n_tiles, q0, wave, lane, strip, cnt_s, my_scratch);
}

// One wave-instruction of a direct-to-LDS load: lane l's 16 bytes at gbase + voff land at LDS byte
// lds_dst + 16 l.  M0 carries the destination and belongs to the compiler, so it is saved and restored inside
// the statement.  The compiler does not count this load: wait with s_waitcnt vmcnt before reading.
__device__ __forceinline__ void lds_dma16(uint32_t lds_dst, const void* gbase, uint32_t voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
}

// LDS reads whose completion the SOURCE waits for, so that they can run a group of MFMAs ahead of their use (the
// compiler issues a fragment read right before the MFMA that needs it and waits for it there).  lds_read16 only
// issues; the value is defined after an lds_wait8<N> that names the register, with N = the number of LDS reads
// issued after it (LDS reads complete in order; anything else the compiler has in flight on the same counter
// only makes the wait stricter).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void lds_read16(u32x4& dst, uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "16-bit offset");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait8(u32x4 (&a)[4], u32x4 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 : "n"(N));
}
template <int... Is, typename F>
__device__ __forceinline__ void unrolled(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}

// The split scan for batches of 256 queries and more: a [256 rows x 256 queries] tile per workgroup of eight
// waves (two per SIMD, 128 accumulator registers each), so 16 values of k bring 32 KiB into LDS for 24 MFMAs
// per wave where the [128 x 256] tile of dense_scan_kernel brings 24 KiB for 12 — the scan is bound by how
// fast a CU can fill its LDS, not by the matrix cores.  The slices (16 NS values of k) go straight from L2 to
// LDS (direct-to-LDS loads, no staging registers) into a ring of RING buffers, RING - 1 slices ahead of the one
// being multiplied; one barrier per slice.  A row of a buffer is the slice itself (NS x [hi x 16 | lo x 16]);
// its 16-byte pieces are XOR-swizzled with low row bits instead of padded (conflict-free ds_read_b128).
// Measured at 1M x 768, 1024 queries: NS = 2, RING = 2 (128 KiB of buffers) 4.93 ms; NS = 1, RING = 4 (the same
// bytes, 1.5 x as many in flight, twice the barriers) 5.06 ms.
// The loads do not check bounds: the caller pads the row image with 256 rows and the query image to whole
// blocks of 256 queries (scores of rows or queries past the end are computed and never used).
using DmaCfg = DenseCfg<4, 2, 2, 4>;
constexpr int DMA_BUF_BYTES = 128 * 1024;                                                // all buffers of the ring
constexpr size_t DMA_LDS_BYTES = DMA_BUF_BYTES + DmaCfg::BQ * 8 + DmaCfg::NW * RAG_SEL_LDS_WORDS * 4;

template <bool EUCLID, int NS, int DMA_RING>
__global__ __launch_bounds__(DmaCfg::NT, 2) void dense_scan_dma_kernel(DenseScanParams p) {
    using Cfg = DmaCfg;
    constexpr int TQ = 2, TR = 4, WR = 2, NT = Cfg::NT, BQ = Cfg::BQ, BN = Cfg::BN;
    constexpr int DMA_BK = 16 * NS;                                              // values of k per slice
    constexpr int ROW_B = DMA_BK * 4;                                            // bytes of a buffer row: 64 or 128
    constexpr int DMA_STAGE_BYTES = (BN + BQ) * ROW_B;                           // 32 or 64 KiB
    constexpr int DMA_PER_WAVE = DMA_STAGE_BYTES / 1024 / Cfg::NW;               // loads per wave and slice
    constexpr int ROWS_PER_DMA = 1024 / ROW_B;                                   // 16 or 8
    constexpr int GROUP_ROWS = ROWS_PER_DMA * Cfg::NW;                           // rows the eight waves fill at once
    static_assert((NS == 1 || NS == 2) && DMA_RING * DMA_STAGE_BYTES == DMA_BUF_BYTES && (DMA_RING & (DMA_RING - 1)) == 0,
                  "ring of 2 x 64 KiB or 4 x 32 KiB");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // [RING buffers][rows 256 x ROW_B | queries 256 x ROW_B], then the per-query counters and thresholds
    int* cnt_s = reinterpret_cast<int*>(smem_raw + DMA_RING * DMA_STAGE_BYTES);
    float* tau_s = reinterpret_cast<float*>(cnt_s + BQ);
    uint32_t* sel_s = reinterpret_cast<uint32_t*>(tau_s + BQ);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave / WR, wr = wave % WR;
    const int id = blockIdx.x;                        // same mapping as dense_scan_kernel
    const int xcd = id & 7;
    const int j = id >> 3;
    const int qb = j % p.n_qblocks;
    const int strip = (j / p.n_qblocks) * 8 + xcd;
    const int q0 = qb * BQ;
    const int64_t row_begin = (int64_t)strip * p.strip_rows;
    const int64_t row_end = min(p.n_rows, row_begin + p.strip_rows);
    uint2* my_scratch = p.scratch + (size_t)blockIdx.x * BQ * p.cap;

    for (int i = tid; i < BQ; i += NT) {
        cnt_s[i] = 0;
        tau_s[i] = q0 + i < p.n_queries
                       ? key_to_f32(__hip_atomic_load(&p.tau_shared[q0 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                       : -INFINITY;
    }

    const int n_tiles = row_begin < row_end ? (int)((row_end - row_begin + BN - 1) / BN) : 0;
    const int k_stages = p.dim / DMA_BK;
    const int n_stages = n_tiles * k_stages;
    const uint32_t stride = (uint32_t)p.dim * 4u;                                // bytes per image row
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem_raw;

    // a wave-instruction fills 1 KiB of a buffer, ROWS_PER_DMA rows.  64-byte rows: lane l -> row l >> 2, stored
    // piece l & 3 = source piece (l & 3) ^ ((row >> 1) & 3); 128-byte rows: row l >> 3, stored piece l & 7 = source
    // piece (l & 7) ^ (row & 7).  Wave w fills rows ROWS_PER_DMA w .. of every group of GROUP_ROWS rows.
    const uint32_t voff = NS == 1 ? (uint32_t)(lane >> 2) * stride + 16u * (uint32_t)((lane & 3) ^ ((lane >> 3) & 3))
                                  : (uint32_t)(lane >> 3) * stride + 16u * (uint32_t)((lane & 7) ^ ((lane >> 3) & 7));
    const unsigned char* rows_b =
        reinterpret_cast<const unsigned char*>(p.rows) + (size_t)(row_begin + ROWS_PER_DMA * wave) * stride;
    const unsigned char* qrs_b =
        reinterpret_cast<const unsigned char*>(p.queries) + (size_t)(q0 + ROWS_PER_DMA * wave) * stride;
    auto dma_stage = [&](int tile, int ks, int buf) {
        const unsigned char* rb = rows_b + (size_t)tile * BN * stride + ks * ROW_B;
        const unsigned char* qbp = qrs_b + ks * ROW_B;
        const uint32_t dst = lds0 + buf * DMA_STAGE_BYTES + 1024 * wave;
#pragma unroll
        for (int i = 0; i < BN / GROUP_ROWS; ++i) lds_dma16(dst + 8192 * i, rb + (size_t)i * GROUP_ROWS * stride, voff);
#pragma unroll
        for (int i = 0; i < BQ / GROUP_ROWS; ++i)
            lds_dma16(dst + BN * ROW_B + 8192 * i, qbp + (size_t)i * GROUP_ROWS * stride, voff);
    };
    static_assert(BN / GROUP_ROWS + BQ / GROUP_ROWS == DMA_PER_WAVE, "loads per wave and slice");

    f32x16 acc[TQ][TR];
#pragma unroll
    for (int a = 0; a < TQ; ++a)
#pragma unroll
        for (int b = 0; b < TR; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // the load cursor runs DMA_RING - 1 slices ahead; past the last slice it stays there (the same slice is
    // loaded again into a buffer nobody reads), so every iteration issues the same number of loads and the
    // count the waits name holds to the end
    int ltile = 0, lks = 0;
    auto advance = [&]() {
        if (lks + 1 < k_stages) ++lks;
        else if (ltile + 1 < n_tiles) { ++ltile; lks = 0; }
    };
    if (n_stages > 0) {
#pragma unroll
        for (int i = 0; i < DMA_RING - 1; ++i) {
            dma_stage(ltile, lks, i);
            advance();
        }
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DMA_RING - 2) * DMA_PER_WAVE) : "memory");
    __syncthreads();

    // lane (row, half) of sub-step s reads the high parts k = 16 s + 8 half .. + 7 (piece 4 s + half) and the low
    // parts (piece 4 s + 2 + half) of its row, each piece at its swizzled place
    const int lrow = lane & 31, half = lane >> 5;
    const int swz = NS == 1 ? (lrow >> 1) & 3 : lrow & 7;
    uint32_t a_h[NS], a_l[NS], q_h[NS], q_l[NS];                       // LDS addresses in buffer 0
#pragma unroll
    for (int sI = 0; sI < NS; ++sI) {
        const uint32_t ph = (uint32_t)lrow * ROW_B + 16u * (uint32_t)((4 * sI + half) ^ swz);
        const uint32_t pl = (uint32_t)lrow * ROW_B + 16u * (uint32_t)((4 * sI + 2 + half) ^ swz);
        a_h[sI] = lds0 + (wr * TR * 32) * ROW_B + ph;
        a_l[sI] = lds0 + (wr * TR * 32) * ROW_B + pl;
        q_h[sI] = lds0 + BN * ROW_B + (wq * TQ * 32) * ROW_B + ph;
        q_l[sI] = lds0 + BN * ROW_B + (wq * TQ * 32) * ROW_B + pl;
    }

    // A slice is multiplied in G = 2 NS groups of 12 MFMAs: group g = (sub-step g >> 1, row sub-tiles 2 (g & 1) and
    // + 1) x both query sub-tiles x the three products, product-major, so an accumulator comes round every fourth
    // MFMA.  The fragment reads of group g + 1 (four row fragments; with an odd g also the four query fragments of
    // the next sub-step) are issued before the MFMAs of group g.
    constexpr int G = 2 * NS;
    int tile = 0, ks = 0;
    for (int st = 0; st < n_stages; ++st) {
        const int buf = st & (DMA_RING - 1);
        dma_stage(ltile, lks, (st + DMA_RING - 1) & (DMA_RING - 1));
        advance();

        const uint32_t bo = (uint32_t)buf * DMA_STAGE_BYTES;
        u32x4 Bf[2][4], Af[2][4];
        auto read_b = [&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            lds_read16<0>(Bf[S & 1][0], q_h[S] + bo);
            lds_read16<32 * ROW_B>(Bf[S & 1][1], q_h[S] + bo);
            lds_read16<0>(Bf[S & 1][2], q_l[S] + bo);
            lds_read16<32 * ROW_B>(Bf[S & 1][3], q_l[S] + bo);
        };
        auto read_a = [&](auto g_c) {
            constexpr int GI = decltype(g_c)::value, S = GI >> 1, T0 = 2 * (GI & 1);
            lds_read16<T0 * 32 * ROW_B>(Af[GI & 1][0], a_h[S] + bo);
            lds_read16<T0 * 32 * ROW_B>(Af[GI & 1][1], a_l[S] + bo);
            lds_read16<(T0 + 1) * 32 * ROW_B>(Af[GI & 1][2], a_h[S] + bo);
            lds_read16<(T0 + 1) * 32 * ROW_B>(Af[GI & 1][3], a_l[S] + bo);
        };
        auto group = [&](auto g_c) {
            constexpr int GI = decltype(g_c)::value, S = GI >> 1, T0 = 2 * (GI & 1);
            if constexpr ((GI & 1) && GI + 1 < G) read_b(std::integral_constant<int, S + 1>{});
            if constexpr (GI + 1 < G) read_a(std::integral_constant<int, GI + 1>{});
            constexpr int newer = (GI + 1 < G ? 4 : 0) + ((GI & 1) && GI + 1 < G ? 4 : 0);
            lds_wait8<newer>(Bf[S & 1], Af[GI & 1]);
            const bf16x8 ah0 = __builtin_bit_cast(bf16x8, Af[GI & 1][0]), al0 = __builtin_bit_cast(bf16x8, Af[GI & 1][1]);
            const bf16x8 ah1 = __builtin_bit_cast(bf16x8, Af[GI & 1][2]), al1 = __builtin_bit_cast(bf16x8, Af[GI & 1][3]);
            const bf16x8 bh0 = __builtin_bit_cast(bf16x8, Bf[S & 1][0]), bh1 = __builtin_bit_cast(bf16x8, Bf[S & 1][1]);
            const bf16x8 bl0 = __builtin_bit_cast(bf16x8, Bf[S & 1][2]), bl1 = __builtin_bit_cast(bf16x8, Bf[S & 1][3]);
            acc[0][T0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh0, acc[0][T0], 0, 0, 0);
            acc[1][T0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al0, bh1, acc[1][T0], 0, 0, 0);
            acc[0][T0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh0, acc[0][T0 + 1], 0, 0, 0);
            acc[1][T0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al1, bh1, acc[1][T0 + 1], 0, 0, 0);
            acc[0][T0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl0, acc[0][T0], 0, 0, 0);
            acc[1][T0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bl1, acc[1][T0], 0, 0, 0);
            acc[0][T0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl0, acc[0][T0 + 1], 0, 0, 0);
            acc[1][T0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bl1, acc[1][T0 + 1], 0, 0, 0);
            acc[0][T0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh0, acc[0][T0], 0, 0, 0);
            acc[1][T0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0, bh1, acc[1][T0], 0, 0, 0);
            acc[0][T0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh0, acc[0][T0 + 1], 0, 0, 0);
            acc[1][T0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1, bh1, acc[1][T0 + 1], 0, 0, 0);
        };
        read_b(std::integral_constant<int, 0>{});
        read_a(std::integral_constant<int, 0>{});
        unrolled(std::make_integer_sequence<int, G>{}, group);

        if (ks == k_stages - 1) {
            tile_epilogue<Cfg, EUCLID, TQ, TR>(p, acc, tile, n_tiles, row_begin, row_end, q0, wq, wr, wave, lane, strip, cnt_s,
                                               tau_s, sel_s, my_scratch);
            ks = 0;
            ++tile;
        } else {
            ++ks;
        }
        // the next slice has landed once at most the loads of the two slices behind it are outstanding
        // (loads retire in issue order; younger memory operations of the epilogue only make this stricter)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DMA_RING - 2) * DMA_PER_WAVE) : "memory");
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    publish_strip<Cfg>(p, n_tiles, q0, wave, lane, strip, cnt_s, my_scratch);
}

// fp32 [n][dim] -> split image [n][dim / 16][hi x 16 | lo x 16] bf16 (dim a multiple of 32): one thread per
// 8 values.
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, int64_t n_rows, int dim,
                                                         bf16* __restrict__ out) {
    const int64_t n8 = n_rows * (dim / 8);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / (dim / 8);
        const int c = (int)(i - row * (dim / 8));          // 8-value group inside the row
        const float4 v0 = *reinterpret_cast<const float4*>(x + row * dim + c * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(x + row * dim + c * 8 + 4);
        const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hi[j] = (bf16)f[j];
            lo[j] = (bf16)(f[j] - (float)hi[j]);
        }
        bf16* o = out + row * 2 * dim + (c >> 1) * 32 + (c & 1) * 8;
        *reinterpret_cast<bf16x8*>(o) = hi;
        *reinterpret_cast<bf16x8*>(o + 16) = lo;
    }
}

// Lower bound of every query's final K-th score from a finished top-K of SOME of the rows: if that
// list is full, its last score is reached by K rows, so the final K-th score cannot be lower.
__global__ void tau_from_topk_kernel(const float* __restrict__ score, const int32_t* __restrict__ count, int n_queries,
                                     int k, uint32_t* __restrict__ tau) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_queries) return;
    tau[q] = count[q] >= k ? f32_to_key(score[(size_t)q * k + k - 1]) : 0x007FFFFFu;   // else the key of -inf
}

// Exact score of chosen (query, row) pairs: one wave per pair.
__global__ void dense_rescore_kernel(const float* __restrict__ rows, const float* __restrict__ queries,
                                     const int64_t* __restrict__ cand, float* __restrict__ out,
                                     int64_t n_pairs, int m, int dim, int64_t row_base, int64_t n_rows,
                                     int euclid) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pair >= n_pairs) return;
    const int64_t q = pair / m;
    const int64_t row = cand[pair] - row_base;
    float s = 0.f;
    if (row >= 0 && row < n_rows) {
        const float* e = rows + row * dim;
        const float* qv = queries + q * dim;
        for (int k = lane * 4; k < dim; k += 256) {
            float4 a = *reinterpret_cast<const float4*>(e + k);
            float4 b = *reinterpret_cast<const float4*>(qv + k);
            if (euclid) {
                float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
                s += dx * dx + dy * dy + dz * dz + dw * dw;
            } else {
                s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (euclid) s = sqrtf(s);
    }
    if (lane == 0) out[pair] = s;
}

__global__ void row_norm2_kernel(const float* __restrict__ rows, float* __restrict__ out, int64_t n_rows, int dim) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* e = rows + row * dim;
    float s = 0.f;
    for (int k = lane * 4; k < dim; k += 256) {
        float4 a = *reinterpret_cast<const float4*>(e + k);
        s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[row] = s;
}

// Euclidean finish: the scan ranks by 2 q.e - |e|^2; turn that into a distance.
__global__ void euclid_finish_kernel(float* score, const int32_t* count, const float* queries, int n_queries,
                                     int k, int dim) {
    const int q = blockIdx.x;
    __shared__ float qn_s;
    float s = 0.f;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) {
        float v = queries[(int64_t)q * dim + i];
        s += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) qn_s = s;   // blockDim.x == 64
    __syncthreads();
    for (int i = threadIdx.x; i < count[q]; i += blockDim.x) {
        float v = qn_s - score[(int64_t)q * k + i];
        score[(int64_t)q * k + i] = sqrtf(fmaxf(v, 0.f));
    }
}

}  // namespace

struct rag_dense_s {
    float* rows = nullptr;
    float* row_norm2 = nullptr;
    float* rows_split = nullptr;   // split image of the rows (rag_dense_enable_split), or null
    DevBuf queries_split, tmp_score, tmp_row, tmp_count;
    bool owns_rows = false;
    int64_t n_rows = 0;
    int dim = 0;
    int metric = 0;
    int64_t row_base = 0;
    int device = 0;
    DevBuf scratch, cand, cand_count, tau_shared;
    DevBuf grp_score, grp_row, grp_count;      // first level of the two-level strip merge: [n_queries][groups][k]
    // optional timing of the scan kernel: a ring of event pairs on the launch stream
    static constexpr int PROF_RING = 256;
    bool profiling = false;
    hipEvent_t ev0[PROF_RING], ev1[PROF_RING];
    int prof_n = 0;
    int last_strips = 0, last_qblocks = 0;
};

extern "C" int rag_dense_create(const float* rows, int64_t n_rows, int32_t dim, int32_t metric,
                                int32_t mem_kind, int64_t row_base, rag_dense_t* out) {
    RAG_REQUIRE(out != nullptr, RAG_EINVAL, "rag_dense_create: out is null");
    *out = nullptr;
    RAG_REQUIRE(rows != nullptr || n_rows == 0, RAG_EINVAL, "rag_dense_create: rows is null");
    RAG_REQUIRE(n_rows >= 0 && n_rows < (1ll << 32), RAG_ELIMIT, "rag_dense_create: n_rows %lld out of range",
                (long long)n_rows);
    RAG_REQUIRE(row_base >= 0 && row_base + n_rows < (1ll << 32), RAG_ELIMIT,
                "rag_dense_create: global rows must stay below 2^32");
    RAG_REQUIRE(dim > 0 && dim % 4 == 0, RAG_EINVAL, "rag_dense_create: dim %d must be a positive multiple of 4", dim);
    RAG_REQUIRE(metric >= RAG_METRIC_COSINE && metric <= RAG_METRIC_EUCLID, RAG_EINVAL,
                "rag_dense_create: unknown metric %d", metric);
    RAG_REQUIRE(mem_kind >= RAG_MEM_HOST_COPY && mem_kind <= RAG_MEM_DEVICE_BORROW, RAG_EINVAL,
                "rag_dense_create: unknown mem_kind %d", mem_kind);
    rag_dense_s* h = new rag_dense_s();
    h->n_rows = n_rows;
    h->dim = dim;
    h->metric = metric;
    h->row_base = row_base;
    RAG_HIP_TRY(hipGetDevice(&h->device));
    const size_t bytes = (size_t)n_rows * dim * sizeof(float);
    if (mem_kind == RAG_MEM_DEVICE_BORROW) {
        h->rows = const_cast<float*>(rows);
    } else if (n_rows > 0) {
        hipError_t e = hipMalloc((void**)&h->rows, bytes);
        if (e != hipSuccess) {
            delete h;
            rag_set_error("rag_dense_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            return RAG_ENOMEM;
        }
        h->owns_rows = true;
        e = hipMemcpy(h->rows, rows, bytes,
                      mem_kind == RAG_MEM_HOST_COPY ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            (void)hipFree(h->rows);
            delete h;
            rag_set_error("rag_dense_create: hipMemcpy failed: %s", hipGetErrorString(e));
            return RAG_EHIP;
        }
    }
    if (metric == RAG_METRIC_EUCLID && n_rows > 0) {
        hipError_t e = hipMalloc((void**)&h->row_norm2, (size_t)n_rows * sizeof(float));
        if (e != hipSuccess) {
            if (h->owns_rows) (void)hipFree(h->rows);
            delete h;
            rag_set_error("rag_dense_create: hipMalloc(norms) failed: %s", hipGetErrorString(e));
            return RAG_ENOMEM;
        }
        row_norm2_kernel<<<(unsigned)ceil_div64(n_rows, 4), 256>>>(h->rows, h->row_norm2, n_rows, dim);
        RAG_HIP_TRY(hipDeviceSynchronize());
    }
    *out = h;
    return RAG_OK;
}

extern "C" int rag_dense_destroy(rag_dense_t h) {
    if (!h) return RAG_OK;
    if (h->owns_rows && h->rows) (void)hipFree(h->rows);
    if (h->row_norm2) (void)hipFree(h->row_norm2);
    if (h->rows_split) (void)hipFree(h->rows_split);
    h->queries_split.release();
    h->tmp_score.release();
    h->tmp_row.release();
    h->tmp_count.release();
    if (h->profiling) (void)rag_dense_profile(h, 0);
    h->scratch.release();
    h->cand.release();
    h->cand_count.release();
    h->tau_shared.release();
    h->grp_score.release();
    h->grp_row.release();
    h->grp_count.release();
    delete h;
    return RAG_OK;
}

extern "C" int64_t rag_dense_rows(rag_dense_t h) { return h ? h->n_rows : -1; }

extern "C" int rag_dense_enable_split(rag_dense_t h, int32_t enable) {
    RAG_REQUIRE(h != nullptr, RAG_EINVAL, "rag_dense_enable_split: null handle");
    if (!enable) {
        if (h->rows_split) (void)hipFree(h->rows_split);
        h->rows_split = nullptr;
        return RAG_OK;
    }
    RAG_REQUIRE(h->dim % 32 == 0, RAG_EINVAL, "rag_dense_enable_split: dim %d is not a multiple of 32", h->dim);
    if (h->rows_split || h->n_rows == 0) return RAG_OK;
    // 256 rows of zeros behind the image: dense_scan_dma_kernel loads whole tiles
    const size_t bytes = (size_t)(h->n_rows + BN_MAX) * h->dim * sizeof(float);
    hipError_t e = hipMalloc((void**)&h->rows_split, bytes);
    if (e == hipSuccess) {
        e = hipMemset((char*)h->rows_split + (size_t)h->n_rows * h->dim * sizeof(float), 0, (size_t)BN_MAX * h->dim * sizeof(float));
        if (e != hipSuccess) (void)hipFree(h->rows_split);
    }
    if (e != hipSuccess) {
        h->rows_split = nullptr;
        rag_set_error("rag_dense_enable_split: allocating %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return RAG_ENOMEM;
    }
    const int64_t n8 = h->n_rows * (h->dim / 8);
    split_rows_kernel<<<(unsigned)(n8 / 256 + 1 > 65536 ? 65536 : n8 / 256 + 1), 256>>>(h->rows, h->n_rows, h->dim,
                                                                                         (bf16*)h->rows_split);
    RAG_HIP_TRY(hipGetLastError());
    RAG_HIP_TRY(hipDeviceSynchronize());
    return RAG_OK;
}

extern "C" int rag_dense_profile(rag_dense_t h, int32_t enable) {
    RAG_REQUIRE(h != nullptr, RAG_EINVAL, "rag_dense_profile: null handle");
    if (enable && !h->profiling) {
        for (int i = 0; i < rag_dense_s::PROF_RING; ++i) {
            RAG_HIP_TRY(hipEventCreate(&h->ev0[i]));
            RAG_HIP_TRY(hipEventCreate(&h->ev1[i]));
        }
        h->profiling = true;
    }
    if (!enable && h->profiling) {
        for (int i = 0; i < rag_dense_s::PROF_RING; ++i) {
            (void)hipEventDestroy(h->ev0[i]);
            (void)hipEventDestroy(h->ev1[i]);
        }
        h->profiling = false;
    }
    h->prof_n = 0;
    return RAG_OK;
}

extern "C" int rag_dense_profile_read(rag_dense_t h, float* out_ms, int32_t max_n, int32_t* out_n) {
    RAG_REQUIRE(h != nullptr && out_ms && out_n, RAG_EINVAL, "rag_dense_profile_read: null argument");
    int n = h->profiling ? h->prof_n : 0;
    if (n > max_n) n = max_n;
    for (int i = 0; i < n; ++i) {
        RAG_HIP_TRY(hipEventSynchronize(h->ev1[i]));
        RAG_HIP_TRY(hipEventElapsedTime(&out_ms[i], h->ev0[i], h->ev1[i]));
    }
    *out_n = n;
    h->prof_n = 0;
    return RAG_OK;
}

namespace {

// Strips per group of the two-level list merge, 0 = one level.  A small batch is scanned in about a thousand strips
// (every CU has to pull on HBM), each of which publishes its own k to 2 k best: one workgroup per query walking all
// 200 k places six times took longer than the scan itself (0.85 ms against 0.54 ms for one query over 1M rows).
// Groups of 16 strips are merged by a workgroup each, then the groups' lists by one: 0.85 -> 0.1 ms.
int merge_group(int n_strips) {
    if (getenv("RAG_DENSE_MERGE_FLAT") || n_strips < 64) return 0;
    return n_strips % 16 == 0 ? 16 : 8;                 // n_strips is a multiple of 8
}

// Geometry of a scan (query blocks, row strips) for p.n_queries / p.n_rows / p.k / p.cap, and the handle's work
// buffers grown to hold it.  No launch: rag_dense_reserve runs this alone.
template <int WQ, int TQ, int WR, int TR>
int plan_scan(rag_dense_t h, DenseScanParams& p) {
    using Cfg = DenseCfg<WQ, TQ, WR, TR>;
    p.n_qblocks = (p.n_queries + Cfg::BQ - 1) / Cfg::BQ;
    // one resident wave of workgroups (LDS decides how many fit a CU), strips a multiple of 8
    const int per_cu = (int)((160 * 1024) / Cfg::LDS_BYTES) > 0 ? (int)((160 * 1024) / Cfg::LDS_BYTES) : 1;
    const int resident = 256 * (per_cu > 4 ? 4 : per_cu);
    int64_t want_strips = (resident + p.n_qblocks - 1) / p.n_qblocks;
    constexpr int BN = Cfg::BN;
    int64_t max_strips = ceil_div64(p.n_rows, BN);
    if (want_strips > max_strips) want_strips = max_strips;
    if (want_strips < 1) want_strips = 1;
    p.strip_rows = ceil_div64(ceil_div64(p.n_rows, want_strips), BN) * BN;
    int64_t strips = ceil_div64(p.n_rows, p.strip_rows);
    p.n_strips = (int)(ceil_div64(strips, 8) * 8);
    p.kpub = p.maxima ? (int)(p.strip_rows / 16) : (2 * p.k < p.cap ? 2 * p.k : p.cap);
    const int64_t n_blocks = (int64_t)p.n_strips * p.n_qblocks;

    int rc = h->scratch.ensure((size_t)n_blocks * Cfg::BQ * p.cap * sizeof(uint2));
    if (rc) return rc;
    rc = h->cand.ensure((size_t)p.n_queries * p.n_strips * p.kpub * sizeof(uint2));
    if (rc) return rc;
    rc = h->cand_count.ensure((size_t)p.n_queries * p.n_strips * sizeof(int32_t));
    if (rc) return rc;
    if (const int g = merge_group(p.n_strips)) {
        const size_t n = (size_t)p.n_queries * (p.n_strips / g);
        if ((rc = h->grp_score.ensure(n * p.k * sizeof(float)))) return rc;
        if ((rc = h->grp_row.ensure(n * p.k * sizeof(int64_t)))) return rc;
        if ((rc = h->grp_count.ensure(n * sizeof(int32_t)))) return rc;
    }
    return h->tau_shared.ensure((size_t)p.n_queries * sizeof(uint32_t));
}

// which tile configuration a batch of n_queries runs
// (`split`: the index holds the split image, whose [256 x 256] direct-to-LDS kernel also beats the 128-query
// tiles for 129 to 255 queries: 1.36 against 1.84 ms per search at 129 queries over 1M rows)
template <typename F>
int dense_dispatch(int n_queries, bool split, F&& f) {
    if (const char* e = getenv("RAG_DENSE_CFG")) {          // tuning aid: force a tile configuration
        const int c = atoi(e);
        if (c == 0) return f(std::integral_constant<int, 0>{});
        if (c == 1) return f(std::integral_constant<int, 1>{});
        if (c == 2) return f(std::integral_constant<int, 2>{});
        if (c == 3) return f(std::integral_constant<int, 3>{});
    }
    if (n_queries <= 32) return f(std::integral_constant<int, 0>{});
    if (n_queries <= 64) return f(std::integral_constant<int, 1>{});
    if (getenv("RAG_DENSE_WIDE") || n_queries <= 128 || (!split && n_queries < 256)) return f(std::integral_constant<int, 2>{});
    return f(std::integral_constant<int, 3>{});
}

// rows of the threshold pre-pass: 1/32 of the shard up to 65536 rows, at least 64 K, in whole tiles of either height
// (the sample's K-th score has to sit well inside the final top K: 64 K rows make it the top 1.6 % of the sample)
int64_t prepass_rows(int k, int64_t n_rows) {
    int64_t n = n_rows / 32 < 65536 ? n_rows / 32 : 65536;
    if (const char* e = getenv("RAG_DENSE_SAMPLE")) n = atoll(e) > 0 ? atoll(e) : n;      // tuning aid
    if (n < 64 * (int64_t)k) n = 64 * (int64_t)k;
    return (n + BN_MAX - 1) / BN_MAX * BN_MAX;
}
// the pre-pass pays for itself from shards of eight samples on (100 k rows, k = 10, 1024 queries: 1.26 -> 0.66 ms)
bool prepass_wanted(int n_queries, int64_t n_rows, int64_t sample) {
    return n_queries >= 128 && n_rows >= 8 * sample && !getenv("RAG_DENSE_NO_PREPASS");
}

int dense_cap(int k) {                                   // holds for either tile height
    int cap = 2 * k + BN_MAX;
    if (cap < 3 * BN_MAX) cap = 3 * BN_MAX;
    if (const char* e = getenv("RAG_DENSE_CAP")) {      // tuning aid: candidate list capacity
        int v = atoi(e);
        if (v >= k + BN_MAX && v <= 8192) cap = v;
    }
    return (cap + 63) / 64 * 64;
}

template <int WQ, int TQ, int WR, int TR, bool DMA = false>
int launch_scan(rag_dense_t h, DenseScanParams& p, hipStream_t stream, bool preset_tau = false) {
    using Cfg = DenseCfg<WQ, TQ, WR, TR>;
    int rc = plan_scan<WQ, TQ, WR, TR>(h, p);
    if (rc) return rc;
    const int64_t n_blocks = (int64_t)p.n_strips * p.n_qblocks;
    p.tau_shared = (uint32_t*)h->tau_shared.p;
    // key of -inf: below every real score
    if (!preset_tau)
        RAG_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)p.tau_shared, 0x007FFFFF, (size_t)p.n_queries, stream));
    p.scratch = (uint2*)h->scratch.p;
    p.cand = (uint2*)h->cand.p;
    p.cand_count = (int32_t*)h->cand_count.p;

    if constexpr (DMA) {
        static_assert(std::is_same<Cfg, DmaCfg>::value, "the direct-to-LDS scan has one tile shape");
        static const bool ring4 = getenv("RAG_DENSE_RING4") != nullptr;     // tuning aid: 4 x 32 KiB buffers
        auto kern = ring4 ? (p.euclid ? dense_scan_dma_kernel<true, 1, 4> : dense_scan_dma_kernel<false, 1, 4>)
                          : (p.euclid ? dense_scan_dma_kernel<true, 2, 2> : dense_scan_dma_kernel<false, 2, 2>);
        RAG_GRANT_LDS(kern, DMA_LDS_BYTES);
        hipLaunchKernelGGL(kern, dim3((unsigned)n_blocks), dim3(Cfg::NT), DMA_LDS_BYTES, stream, p);
    } else {
        auto kern = p.split ? (p.euclid ? dense_scan_kernel<WQ, TQ, WR, TR, true, true> : dense_scan_kernel<WQ, TQ, WR, TR, false, true>)
                            : (p.euclid ? dense_scan_kernel<WQ, TQ, WR, TR, true, false> : dense_scan_kernel<WQ, TQ, WR, TR, false, false>);
        RAG_GRANT_LDS(kern, Cfg::LDS_BYTES);
        hipLaunchKernelGGL(kern, dim3((unsigned)n_blocks), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p);
    }
    RAG_HIP_TRY(hipGetLastError());
    h->last_strips = p.n_strips;
    h->last_qblocks = p.n_qblocks;
    return RAG_OK;
}

// The k best of every query's per-strip lists, sorted: in one level, or groups of strips first (merge_group).
int merge_strips(rag_dense_t h, const DenseScanParams& p, int n_queries, int k, float* out_score, int64_t* out_row,
                 int32_t* out_count, hipStream_t stream) {
    const int g = merge_group(p.n_strips);
    if (g == 0)
        return launch_merge_cand32(p.cand, p.cand_count, n_queries, p.n_strips, p.kpub, k, h->row_base, out_score, out_row,
                                   out_count, p.fault, stream);
    // (query, group) pairs are consecutive runs of g lists: the one-level kernel on n_queries * groups "queries"
    const int groups = p.n_strips / g;
    float* g_score = (float*)h->grp_score.p;
    int64_t* g_row = (int64_t*)h->grp_row.p;
    int32_t* g_count = (int32_t*)h->grp_count.p;
    int rc = launch_merge_cand32(p.cand, p.cand_count, n_queries * groups, g, p.kpub, k, h->row_base, g_score, g_row,
                                 g_count, p.fault, stream);
    if (rc) return rc;
    ragmerge::ListF32Src src{g_score, g_row, g_count, groups, k, 0};
    hipLaunchKernelGGL(ragmerge::merge_topk_kernel<ragmerge::ListF32Src>, dim3(n_queries), dim3(ragmerge::MERGE_NT), 0,
                       stream, src, k, (int64_t)0, 0, out_score, out_row, out_count, p.fault);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

}  // namespace

extern "C" int rag_dense_search_dev(rag_dense_t h, const float* d_queries, int32_t n_queries, int32_t k,
                                    const uint8_t* d_row_mask, float* d_out_score, int64_t* d_out_row,
                                    int32_t* d_out_count, void* stream_) {
    RAG_REQUIRE(h != nullptr, RAG_EINVAL, "rag_dense_search: null handle");
    RAG_REQUIRE(n_queries >= 0 && k >= 1 && k <= RAG_MAX_K, RAG_ELIMIT,
                "rag_dense_search: need n_queries >= 0 and 1 <= k <= %d (got %d, %d)", RAG_MAX_K, n_queries, k);
    RAG_REQUIRE(d_queries && d_out_score && d_out_row && d_out_count, RAG_EINVAL, "rag_dense_search: null pointer");
    if (n_queries == 0) return RAG_OK;
    hipStream_t stream = (hipStream_t)stream_;
    if (h->n_rows == 0) {
        RAG_HIP_TRY(hipMemsetAsync(d_out_count, 0, (size_t)n_queries * sizeof(int32_t), stream));
        RAG_HIP_TRY(hipMemsetAsync(d_out_row, 0xff, (size_t)n_queries * k * sizeof(int64_t), stream));
        RAG_HIP_TRY(hipMemsetAsync(d_out_score, 0, (size_t)n_queries * k * sizeof(float), stream));
        return RAG_OK;
    }
    DenseScanParams p;
    p.rows = h->rows;
    p.queries = d_queries;
    p.row_norm2 = h->row_norm2;
    p.row_mask = d_row_mask;
    p.n_rows = h->n_rows;
    p.dim = h->dim;
    p.n_queries = n_queries;
    p.k = k;
    p.kpub = k;
    p.cap = dense_cap(k);
    p.euclid = h->metric == RAG_METRIC_EUCLID;
    p.split = 0;
    if (h->rows_split && !getenv("RAG_DENSE_NO_SPLIT")) {
        // whole blocks of 256 queries: dense_scan_dma_kernel loads whole tiles (what lies past the batch is never used)
        int rc0 = h->queries_split.ensure(((size_t)n_queries + BN_MAX) * h->dim * sizeof(float));
        if (rc0) return rc0;
        const int64_t n8 = (int64_t)n_queries * (h->dim / 8);
        split_rows_kernel<<<(unsigned)(n8 / 256 + 1 > 65536 ? 65536 : n8 / 256 + 1), 256, 0, stream>>>(
            d_queries, n_queries, h->dim, (bf16*)h->queries_split.p);
        RAG_HIP_TRY(hipGetLastError());
        p.rows = h->rows_split;
        p.queries = (const float*)h->queries_split.p;
        p.split = 1;
    }
    p.debug_row_mod = 0;
    p.maxima = 0;
    p.fault = rag_fault_word();
    if (const char* e = getenv("RAG_DENSE_ROW_MOD")) p.debug_row_mod = atoi(e);
    int rc;
    auto scan = [&](DenseScanParams& pp, bool preset) -> int {
        return dense_dispatch(n_queries, pp.split != 0, [&](auto cfg) -> int {
            constexpr int C = decltype(cfg)::value;
            if constexpr (C == 0) return launch_scan<1, 1, 4, 1>(h, pp, stream, preset);
            else if constexpr (C == 1) return launch_scan<1, 2, 4, 1>(h, pp, stream, preset);
            else if constexpr (C == 2) return launch_scan<2, 2, 2, 2>(h, pp, stream, preset);
            else if (pp.split && !getenv("RAG_DENSE_NO_DMA")) return launch_scan<4, 2, 2, 4, true>(h, pp, stream, preset);
            else return launch_scan<4, 2, 2, 2>(h, pp, stream, preset);
        });
    };
    // Large batches over large shards: a pre-pass over the first rows (about 1.6 % of the scan)
    // gives every query a threshold before the main pass starts, so the main pass appends and
    // compacts candidates only for rows that beat the pre-pass's K-th score instead of letting
    // each of its strips find a threshold from -inf on its own.  The result is the same exact
    // top-K: the threshold is reached by K real rows.
    const int64_t sample = prepass_rows(k, h->n_rows);
    // timing aid: one event pair around everything the scan does (pre-pass included), not the final merge
    const bool prof = h->profiling && h->prof_n < rag_dense_s::PROF_RING && !rag_stream_capturing(stream);
    if (prof) RAG_HIP_TRY(hipEventRecord(h->ev0[h->prof_n], stream));
    const bool prepass = prepass_wanted(n_queries, h->n_rows, sample);
    if (prepass) {
        DenseScanParams p0 = p;
        p0.n_rows = sample;
        // without a row mask the pre-pass keeps no lists: the K-th largest of the 16-row group maxima is the bound
        p0.maxima = d_row_mask == nullptr && !getenv("RAG_DENSE_LIST_PREPASS");
        if ((rc = scan(p0, false))) return rc;
        if ((rc = launch_merge_cand32(p0.cand, p0.cand_count, n_queries, p0.n_strips, p0.kpub, k, h->row_base, d_out_score,
                                      d_out_row, d_out_count, p.fault, stream)))
            return rc;
        tau_from_topk_kernel<<<(n_queries + 255) / 256, 256, 0, stream>>>(d_out_score, d_out_count, n_queries, k,
                                                                          (uint32_t*)h->tau_shared.p);
        RAG_HIP_TRY(hipGetLastError());
    }
    rc = scan(p, prepass);
    if (rc) return rc;
    if (prof) {
        RAG_HIP_TRY(hipEventRecord(h->ev1[h->prof_n], stream));
        h->prof_n++;
    }
    if (p.split) {
        // The split products chose the rows; their scores (good to about 1e-6) are replaced by plain fp32
        // scores of the same rows and the list is put in order again, so callers that amplify score
        // differences (min-max normalisation in weighted fusion) see fp32 accuracy.
        const size_t nk = (size_t)n_queries * k;
        if ((rc = h->tmp_score.ensure(nk * sizeof(float)))) return rc;
        if ((rc = h->tmp_row.ensure(nk * sizeof(int64_t)))) return rc;
        if ((rc = h->tmp_count.ensure((size_t)n_queries * sizeof(int32_t)))) return rc;
        float* t_score = (float*)h->tmp_score.p;
        int64_t* t_row = (int64_t*)h->tmp_row.p;
        int32_t* t_count = (int32_t*)h->tmp_count.p;
        rc = merge_strips(h, p, n_queries, k, t_score, t_row, t_count, stream);
        if (rc) return rc;
        dense_rescore_kernel<<<(unsigned)ceil_div64((int64_t)nk, 4), 256, 0, stream>>>(
            h->rows, d_queries, t_row, t_score, (int64_t)nk, k, h->dim, h->row_base, h->n_rows, p.euclid);
        RAG_HIP_TRY(hipGetLastError());
        ragmerge::ListF32Src src{t_score, t_row, t_count, 1, k, p.euclid ? 1 : 0};
        hipLaunchKernelGGL(ragmerge::merge_topk_kernel<ragmerge::ListF32Src>, dim3(n_queries), dim3(ragmerge::MERGE_NT),
                           0, stream, src, k, (int64_t)0, p.euclid ? 1 : 0, d_out_score, d_out_row, d_out_count, p.fault);
        RAG_HIP_TRY(hipGetLastError());
        return RAG_OK;
    }
    rc = merge_strips(h, p, n_queries, k, d_out_score, d_out_row, d_out_count, stream);
    if (rc) return rc;
    if (p.euclid) {
        euclid_finish_kernel<<<n_queries, 64, 0, stream>>>(d_out_score, d_out_count, d_queries, n_queries, k, h->dim);
        RAG_HIP_TRY(hipGetLastError());
    }
    return RAG_OK;
}

extern "C" int rag_dense_reserve(rag_dense_t h, int32_t max_queries, int32_t max_k) {
    RAG_REQUIRE(h != nullptr, RAG_EINVAL, "rag_dense_reserve: null handle");
    RAG_REQUIRE(max_queries >= 1 && max_k >= 1 && max_k <= RAG_MAX_K, RAG_ELIMIT,
                "rag_dense_reserve: need max_queries >= 1 and 1 <= max_k <= %d (got %d, %d)", RAG_MAX_K, max_queries, max_k);
    if (h->n_rows == 0) return RAG_OK;
    // Every plan a batch of up to max_queries can run.  The work buffers are sized by the plan, and a plan's sizes are
    // not monotone in the batch size (strips are rounded per query-block count), so every query-block count of every
    // tile configuration is planned: batch sizes at both sides of every multiple of 32 (the smallest query block),
    // each as the main pass and as both forms of the pre-pass (group maxima; lists, which a row mask asks for).
    const int64_t sample = prepass_rows(max_k, h->n_rows);
    auto plan = [&](int nq, int64_t n_rows, int maxima) -> int {
        DenseScanParams p = {};
        p.n_rows = n_rows;
        p.maxima = maxima;
        p.n_queries = nq;
        p.k = max_k;
        p.kpub = max_k;
        p.cap = dense_cap(max_k);
        return dense_dispatch(nq, h->rows_split != nullptr, [&](auto cfg) -> int {
            constexpr int C = decltype(cfg)::value;
            if constexpr (C == 0) return plan_scan<1, 1, 4, 1>(h, p);
            else if constexpr (C == 1) return plan_scan<1, 2, 4, 1>(h, p);
            else if constexpr (C == 2) return plan_scan<2, 2, 2, 2>(h, p);
            else {
                int rc3 = h->rows_split ? plan_scan<4, 2, 2, 4>(h, p) : RAG_OK;
                return rc3 ? rc3 : plan_scan<4, 2, 2, 2>(h, p);
            }
        });
    };
    auto plan_batch = [&](int nq) -> int {
        if (nq < 1 || nq > max_queries) return RAG_OK;
        int rc = plan(nq, h->n_rows, 0);
        if (!rc && prepass_wanted(nq, h->n_rows, sample)) {
            rc = plan(nq, sample, 1);
            if (!rc) rc = plan(nq, sample, 0);
        }
        return rc;
    };
    for (int nq = 32; nq <= max_queries + 31; nq += 32) {
        int rc = plan_batch(nq < max_queries ? nq : max_queries);
        if (!rc) rc = plan_batch(nq - 31);           // first batch size of this query-block count
        if (!rc) rc = plan_batch(nq + 1);
        if (rc) return rc;
    }
    int rc;
    if (h->rows_split && (rc = h->queries_split.ensure(((size_t)max_queries + BN_MAX) * h->dim * sizeof(float)))) return rc;
    const size_t nk = (size_t)max_queries * max_k;
    if ((rc = h->tmp_score.ensure(nk * sizeof(float)))) return rc;
    if ((rc = h->tmp_row.ensure(nk * sizeof(int64_t)))) return rc;
    return h->tmp_count.ensure((size_t)max_queries * sizeof(int32_t));
}

extern "C" int rag_dense_search(rag_dense_t h, const float* queries, int32_t n_queries, int32_t k,
                                const uint8_t* row_mask, float* out_score, int64_t* out_row, int32_t* out_count) {
    RAG_REQUIRE(h != nullptr, RAG_EINVAL, "rag_dense_search: null handle");
    RAG_REQUIRE(n_queries >= 0 && k >= 1 && k <= RAG_MAX_K, RAG_ELIMIT,
                "rag_dense_search: need n_queries >= 0 and 1 <= k <= %d (got %d, %d)", RAG_MAX_K, n_queries, k);
    if (n_queries == 0) return RAG_OK;
    RAG_REQUIRE(queries && out_score && out_row && out_count, RAG_EINVAL, "rag_dense_search: null pointer");
    const size_t qb = (size_t)n_queries * h->dim * sizeof(float);
    const size_t nk = (size_t)n_queries * k;
    float* d_q = nullptr;
    uint8_t* d_mask = nullptr;
    float* d_s = nullptr;
    int64_t* d_r = nullptr;
    int32_t* d_c = nullptr;
    int rc = RAG_OK;
    auto cleanup = [&]() {
        if (d_q) (void)hipFree(d_q);
        if (d_mask) (void)hipFree(d_mask);
        if (d_s) (void)hipFree(d_s);
        if (d_r) (void)hipFree(d_r);
        if (d_c) (void)hipFree(d_c);
    };
#define TRY_OR_CLEAN(expr)                                                                       \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            rag_set_error("%s failed: %s", #expr, hipGetErrorString(_e));                        \
            cleanup();                                                                           \
            return RAG_EHIP;                                                                     \
        }                                                                                        \
    } while (0)
    TRY_OR_CLEAN(hipMalloc((void**)&d_q, qb));
    TRY_OR_CLEAN(hipMalloc((void**)&d_s, nk * sizeof(float)));
    TRY_OR_CLEAN(hipMalloc((void**)&d_r, nk * sizeof(int64_t)));
    TRY_OR_CLEAN(hipMalloc((void**)&d_c, (size_t)n_queries * sizeof(int32_t)));
    TRY_OR_CLEAN(hipMemcpy(d_q, queries, qb, hipMemcpyHostToDevice));
    if (row_mask && h->n_rows > 0) {
        TRY_OR_CLEAN(hipMalloc((void**)&d_mask, (size_t)h->n_rows));
        TRY_OR_CLEAN(hipMemcpy(d_mask, row_mask, (size_t)h->n_rows, hipMemcpyHostToDevice));
    }
    rc = rag_dense_search_dev(h, d_q, n_queries, k, d_mask, d_s, d_r, d_c, nullptr);
    if (rc) {
        cleanup();
        return rc;
    }
    TRY_OR_CLEAN(hipMemcpy(out_score, d_s, nk * sizeof(float), hipMemcpyDeviceToHost));
    TRY_OR_CLEAN(hipMemcpy(out_row, d_r, nk * sizeof(int64_t), hipMemcpyDeviceToHost));
    TRY_OR_CLEAN(hipMemcpy(out_count, d_c, (size_t)n_queries * sizeof(int32_t), hipMemcpyDeviceToHost));
    cleanup();
    return RAG_OK;
}

extern "C" int rag_dense_rescore_dev(rag_dense_t h, const float* d_queries, int32_t n_queries,
                                     const int64_t* d_cand_rows, int32_t m, float* d_out_score, void* stream_) {
    RAG_REQUIRE(h != nullptr, RAG_EINVAL, "rag_dense_rescore: null handle");
    RAG_REQUIRE(n_queries >= 0 && m >= 0, RAG_EINVAL, "rag_dense_rescore: negative size");
    const int64_t n_pairs = (int64_t)n_queries * m;
    if (n_pairs == 0) return RAG_OK;
    RAG_REQUIRE(d_queries && d_cand_rows && d_out_score, RAG_EINVAL, "rag_dense_rescore: null pointer");
    dense_rescore_kernel<<<(unsigned)ceil_div64(n_pairs, 4), 256, 0, (hipStream_t)stream_>>>(
        h->rows, d_queries, d_cand_rows, d_out_score, n_pairs, m, h->dim, h->row_base, h->n_rows,
        h->metric == RAG_METRIC_EUCLID);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

extern "C" int rag_dense_rescore(rag_dense_t h, const float* queries, int32_t n_queries, const int64_t* cand_rows,
                                 int32_t m, float* out_score) {
    RAG_REQUIRE(h != nullptr, RAG_EINVAL, "rag_dense_rescore: null handle");
    RAG_REQUIRE(n_queries >= 0 && m >= 0, RAG_EINVAL, "rag_dense_rescore: negative size");
    const size_t n_pairs = (size_t)n_queries * m;
    if (n_pairs == 0) return RAG_OK;
    RAG_REQUIRE(queries && cand_rows && out_score, RAG_EINVAL, "rag_dense_rescore: null pointer");
    float* d_q = nullptr;
    int64_t* d_c = nullptr;
    float* d_o = nullptr;
    auto cleanup = [&]() {
        if (d_q) (void)hipFree(d_q);
        if (d_c) (void)hipFree(d_c);
        if (d_o) (void)hipFree(d_o);
    };
    TRY_OR_CLEAN(hipMalloc((void**)&d_q, (size_t)n_queries * h->dim * sizeof(float)));
    TRY_OR_CLEAN(hipMalloc((void**)&d_c, n_pairs * sizeof(int64_t)));
    TRY_OR_CLEAN(hipMalloc((void**)&d_o, n_pairs * sizeof(float)));
    TRY_OR_CLEAN(hipMemcpy(d_q, queries, (size_t)n_queries * h->dim * sizeof(float), hipMemcpyHostToDevice));
    TRY_OR_CLEAN(hipMemcpy(d_c, cand_rows, n_pairs * sizeof(int64_t), hipMemcpyHostToDevice));
    int rc = rag_dense_rescore_dev(h, d_q, n_queries, d_c, m, d_o, nullptr);
    if (rc) {
        cleanup();
        return rc;
    }
    TRY_OR_CLEAN(hipMemcpy(out_score, d_o, n_pairs * sizeof(float), hipMemcpyDeviceToHost));
    cleanup();
    return RAG_OK;
#undef TRY_OR_CLEAN
}
