// btba_solve_mid.hpp -- k_solve_mid: the per-instance system solve of windows of 22 ... 31 frames in ONE launch (round 6).
//
// BASELINE configs[3] (K = 30 keyframes) ran rounds 1-4's three-launch path until now -- k_big_reduce + k_big_assemble + k_system_solve,
// 5.0 + 11.4 + 20.7 us per Gauss-Newton iteration at c4 x 32 (profiles/r05/bench_c4x32_r05_kernel_stats.csv) -- because k_solve_small
// (btba_solve_small.hpp) keeps the 6 (N - 1) square system matrix in LDS next to the reduced pair sums, which stops fitting one compute
// unit's 160 KB at 22 frames.  The same Gauss-Newton step
//   reduce the sweep partials -> A = w_s JsT Js + JdT Jd, b -> 5 x Jacobi-PCG -> x <- Log(Exp(delta) Exp(x)), T, T^-1
// (SolverBundling.cu:575-651, 692-818, 805-815, 890-897; SolverBundlingDenseUtil.h:349-385) with the matrix NEVER in LDS:
//
//   reduce     as k_solve_small: the pair sums (sparse [P][44], dense [Pd][28]: 125 KB at K = 30) land in LDS, one fabric round trip
//   frame sums as k_solve_small: a frame's 20 sparse and 27 dense sums over its pairs, formed first, expanded once
//   assemble   GATHER, into registers: four lanes own a matrix row, 48 columns (8 frame blocks) each, and form their 48 entries from
//              the pair sums in LDS -- cross blocks from the pair's record through the 36-entry descriptor table (transposed below the
//              diagonal), the diagonal block from the frame sums.  k_solve_small scatters the same values into an LDS matrix and then
//              loads its rows into registers for the PCG; here the rows are born there.
//   PCG        a step = 24 packed FMAs per lane against p (12 broadcast 16-byte LDS reads), two DPP adds, A p through LDS, a barrier,
//              the dot products and vector updates on wave 0 alone (three entries per lane: <= 192 unknowns), a barrier
//   update     as k_solve_small: one lane per frame for Exp / Log, sixteen lanes per frame for the generic cofactor inverse
//
// The same sums as k_system_solve / k_solve_small, in another order (the gate is the parity suite: c4 per iterate against the oracle and
// against the reference's own solver, tests/test_gpu_fullsize.py, test_gpu_vs_reference.py; this kernel against k_system_solve,
// tests/test_gpu_parity.py).  Phases 1, 2b and 4 are k_solve_small's code with this kernel's layout constants -- duplicated on purpose:
// that kernel's instruction schedule was tuned phase by phase (profiles/r05/solve_small.json) and is left alone.
#pragma once
#include "btba_solve_small.hpp"

namespace btba {

constexpr int kMidMaxFrames = 31;        // BTBA_MAX_FRAMES_LDS: 6 (N - 1) <= 180 unknowns, P <= 465 pairs: pair sums + tables <= 160 KB of LDS
constexpr int kMidCPL = 48;              // matrix columns per lane, four lanes per row: 192 columns
constexpr int kMidNA = 4 * kMidCPL;

struct MidLayout {
    static constexpr int NF = kMidMaxFrames, NP = NF * (NF - 1) / 2;
    static constexpr int lv = kMidNA + 4;                                    // a vector: 192 entries + a 16-byte pad
    static constexpr int op = 0, oAp = op + lv, ob = oAp + lv, oM = ob + lv, od = oM + lv;
    static constexpr int oT = od + lv;                                       // this iterate's T [N][16]
    static constexpr int oE = oT + 16 * NF;                                  // the next iterate's T
    static constexpr int ox = oE + 16 * NF;                                  // this iterate's x
    static constexpr int oF = ox + round4(6 * NF);                           // frame sums [(N - 1)][48]
    static constexpr int olut = oF + (NF - 1) * kFrameSums;                  // ints: the 72 entry descriptors (36 diagonal-block, 36 cross-block)
    static constexpr int opij = olut + 288;                                  // ints: canonical pair -> (i << 8 | j)
    static constexpr int ocross = opij + round4(NP);                         // ints: canonical pair -> dense pair with its cross block
    static constexpr int oadjoff = ocross + round4(NP);                      // ints: N + 1
    static constexpr int oadj = oadjoff + round4(NF + 1);                    // ints: 2 Pd <= 2 NP (the host sends explicit lists with more dense pairs to k_system_solve)
    static constexpr int ops = oadj + round4(2 * NP);                        // reduced sparse pair sums [P][44], behind them the dense ones [Pd][28]
    static constexpr int fixed = ops;
};
__host__ inline size_t mid_solve_lds_floats(int N, int Pd)
{
    return (size_t)MidLayout::fixed + (size_t)(N * (N - 1) / 2) * kSparseVals + (size_t)Pd * kDenseVals + 8 * kSparseVals + 16;      // (+ slack: the frame sums' unconditional loads run past the last pair)
}

__global__ void __launch_bounds__(kSmallBlock) k_solve_mid(const SmallSolveArgs S)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using L = MidLayout;
    constexpr unsigned nthr = kSmallBlock;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef btba_f4v v4;
    const unsigned tid = threadIdx.x, b = blockIdx.x;
    const int N = S.n_frames, P = S.n_pairs, Pd = S.n_dense_pairs, na = 6 * (N - 1);
    float *vp = lds + L::op, *vAp = lds + L::oAp, *vb = lds + L::ob, *vM = lds + L::oM, *vd = lds + L::od;
    float *vT = lds + L::oT, *vE = lds + L::oE, *x_l = lds + L::ox, *F = lds + L::oF;
    int *lut_l = reinterpret_cast<int *>(lds + L::olut);
    int *pair_ij_l = reinterpret_cast<int *>(lds + L::opij), *cross_l = reinterpret_cast<int *>(lds + L::ocross);
    int *adj_off_l = reinterpret_cast<int *>(lds + L::oadjoff), *adj_l = reinterpret_cast<int *>(lds + L::oadj);
    float *ps = lds + L::ops, *pd = ps + __umul24(P, kSparseVals);

    float *tr = S.trace ? S.trace + (size_t)b * (size_t)S.trace_instance + (size_t)S.iter * S.trace_record : nullptr;
    const long long clk0 = tr ? (long long)clock64() : 0;
#define BTBA_MSTAMP(slot) do { if (tr && tid == 0) tr[S.tr_clk + (slot)] = (float)((long long)clock64() - clk0); } while (0)

    // ---- phase 1: everything this solve reads from global memory, issued at once; loads and stores without conditions around them (a dead slot
    // repeats the last live one) -- see k_solve_small
    const float *T_in = S.T + __umul24(b, (unsigned)S.pose_stride), *x_in = S.x + __umul24(b, (unsigned)S.x_stride);
    const v4 *sp4 = reinterpret_cast<const v4 *>(S.sparse_partials + (size_t)b * S.sp_stride);
    const v4 *dp4 = reinterpret_cast<const v4 *>(S.dense_partials + (size_t)b * S.dp_stride);
    const float st_T = T_in[min(tid, 16u * N - 1u)];
    const float st_x = x_in[min(tid, 6u * N - 1u)];
    const int st_pij = S.pair_ij[min(tid, (unsigned)P - 1u)];
    const int st_cross = Pd ? S.cross[min(tid, (unsigned)P - 1u)] : -1;
    const int st_ao = Pd ? S.adj_off[min(tid, (unsigned)N)] : 0;
    const int st_adj = Pd ? S.adj[min(tid, 2u * Pd - 1u)] : 0;
    const int st_lut = S.entry_lut[min(tid, 287u)];
    constexpr int kS4 = kSparseVals / 4, kD4 = kDenseVals / 4;
    const int ns4 = S.use_sparse ? P * kS4 : 0, nd4 = Pd * kD4, n4 = ns4 + nd4;
    v4 *ps4 = reinterpret_cast<v4 *>(ps), *pd4 = reinterpret_cast<v4 *>(pd);
    if (S.sparse_chunks == 1 && S.dense_tiles == 1) {
        // one partial per sum (every chip-filling batch): the reduced arrays ARE the partial arrays.  Four slots per lane in flight (K = 30: 7 830 slots)
        for (int e0 = (int)tid; e0 < n4; e0 += 4 * nthr) {
            int e_[4]; v4 f_[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { e_[u] = min(e0 + u * (int)nthr, n4 - 1); f_[u] = *(e_[u] < ns4 ? sp4 + e_[u] : dp4 + (e_[u] - ns4)); }
#pragma unroll
            for (int u = 0; u < 4; u++) *(e_[u] < ns4 ? ps4 + e_[u] : pd4 + (e_[u] - ns4)) = f_[u];
        }
    } else {
        // several partials per sum (small batches): two slots per lane, four or eight partials of each in flight; sums in partial order
        const int max_parts = max(S.sparse_chunks, S.dense_tiles);
        for (int e0 = (int)tid; e0 < n4; e0 += 2 * nthr) {
            const v4 *src[2]; int per[2], parts[2], e_[2];
            v4 acc[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int e = min(e0 + u * (int)nthr, n4 - 1);
                const bool sp = e < ns4;
                const int q = sp ? e : e - ns4, rec = sp ? q / kS4 : q / kD4;
                e_[u] = e; per[u] = sp ? kS4 : kD4; parts[u] = sp ? S.sparse_chunks : S.dense_tiles;
                src[u] = (sp ? sp4 : dp4) + __umul24(__umul24(rec, parts[u]), per[u]) + (q - rec * per[u]);
                acc[u] = (v4){ 0.f, 0.f, 0.f, 0.f };
            }
            auto round = [&](auto width_c, int c0) {
                constexpr int kW = decltype(width_c)::value;
                v4 g[2][kW];
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int c = 0; c < kW; c++) g[u][c] = src[u][__umul24(c0 + c < parts[u] ? c0 + c : 0, per[u])];
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int c = 0; c < kW; c++) { const bool live = c0 + c < parts[u]; acc[u] += (v4){ live ? g[u][c].x : 0.0f, live ? g[u][c].y : 0.0f, live ? g[u][c].z : 0.0f, live ? g[u][c].w : 0.0f }; }
            };
            if (max_parts <= 4) round(std::integral_constant<int, 4>{}, 0);
            else for (int c0 = 0; c0 < max_parts; c0 += 8) round(std::integral_constant<int, 8>{}, c0);
#pragma unroll
            for (int u = 0; u < 2; u++) *(e_[u] < ns4 ? ps4 + e_[u] : pd4 + (e_[u] - ns4)) = acc[u];
        }
    }
    if (!S.use_sparse) for (int e = (int)tid; e < P * kS4; e += nthr) ps4[e] = (v4){ 0.f, 0.f, 0.f, 0.f };      // (read with weight 0 below)
    vT[min(tid, 16u * N - 1u)] = st_T;
    x_l[min(tid, 6u * N - 1u)] = st_x;
    pair_ij_l[min(tid, (unsigned)P - 1u)] = st_pij; cross_l[min(tid, (unsigned)P - 1u)] = st_cross;
    if (Pd) { adj_off_l[min(tid, (unsigned)N)] = st_ao; adj_l[min(tid, 2u * Pd - 1u)] = st_adj; }
    lut_l[min(tid, 287u)] = st_lut;
    if (tid < (unsigned)(L::lv - na)) vp[na + tid] = 0.0f;                   // p beyond na: the padded columns multiply zeros
    BTBA_MSTAMP(0);
    __syncthreads();
    BTBA_MSTAMP(1);

    const float w_s = S.use_sparse ? S.w_sparse : 0.0f;
    // ---- phase 2b: frame sums (k_solve_small's; sparse sums on lanes 0 .. 20 (N - 1) - 1, dense ones -- four per lane -- on lanes 640 .. 640 + 7 (N - 1) - 1)
    if (tid < 20u * (N - 1)) {
        const unsigned fk1 = tid / 20u, fv = tid - 20u * fk1;                // frame fk1 + 1
        const int fk = (int)fk1 + 1;
        unsigned off_i, off_j;
        float sg_i = 1.0f;
        if (fv == 0) { off_i = off_j = 0; }
        else if (fv < 4) { off_i = fv; off_j = fv + 3; }
        else if (fv < 10) { off_i = fv + 3; off_j = fv + 9; }
        else if (fv < 13) { off_i = off_j = fv + 18; sg_i = -1.0f; }
        else if (fv < 16) { off_i = fv + 18; off_j = fv + 21; sg_i = -1.0f; }
        else if (fv == 16) { off_i = off_j = 37; }
        else { off_i = fv + 21; off_j = fv + 24; }
        const unsigned lim = __umul24((unsigned)P - 1u, kSparseVals) + off_j;
        unsigned aj = __umul24((unsigned)fk - 1u, kSparseVals) + off_j;      // pair (0, k)
        const float *pi = ps + __umul24((unsigned)(fk * N - fk * (fk + 1) / 2 - fk), kSparseVals) + off_i;      // pair (k, q + 1) at + 44 q
        int stride = kSparseVals * (N - 2);
        float acc_j = 0.0f, acc_i = 0.0f;
        for (int q0 = 0; q0 < N - 1; q0 += 8) {
            float vj[8], vi[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { vj[u] = ps[min(aj, lim)]; vi[u] = pi[kSparseVals * (q0 + u)]; aj += stride; stride -= kSparseVals; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const int q = q0 + u; acc_j += (q < fk) ? vj[u] : 0.0f; acc_i += (q >= fk && q < N - 1) ? vi[u] : 0.0f; }
        }
        F[__umul24(fk1, kFrameSums) + fv] = acc_j + sg_i * acc_i;
    } else if (tid >= 640u && tid < 640u + 7u * (N - 1)) {
        const unsigned t = tid - 640u, fk1 = t / 7u, sl = t - 7u * fk1;      // slot sl of frame fk1 + 1: record floats 4 sl .. 4 sl + 3 (S: 0 .. 20, g: 21 .. 26, count: 27)
        v4 acc = (v4){ 0.f, 0.f, 0.f, 0.f };
        if (Pd) {
            const int qa = adj_off_l[fk1 + 1], qb = adj_off_l[fk1 + 2];
            const v4 *pd4s = reinterpret_cast<const v4 *>(pd) + sl;
            for (int q0 = qa; q0 < qb; q0 += 8) {
                int a[8];
                v4 v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) a[u] = adj_l[min(q0 + u, qb - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = pd4s[__umul24(a[u] >> 1, kDenseVals / 4)];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    // g: + for the source frame (row_j = a), - for the target frame (row_i = -a); S: + for both.  (a dead slot repeats the last entry with weight 0)
                    const float live = q0 + u < qb ? 1.0f : 0.0f, sg = (a[u] & 1) ? live : -live;
                    const float wx = sl == 6u ? sg : live, wr = sl >= 5u ? sg : live;
                    acc.x += wx * v[u].x; acc.y += wr * v[u].y; acc.z += wr * v[u].z; acc.w += wr * v[u].w;
                }
            }
        }
        *reinterpret_cast<v4 *>(F + __umul24(fk1, kFrameSums) + 20u + 4u * sl) = acc;
    }
    __syncthreads();
    BTBA_MSTAMP(2);

    // ---- phase 2c: the matrix rows, gathered into registers.  Lane (row a_row, quarter h) owns columns 48 h .. 48 h + 47 = the 6 x 6 blocks of column
    // frames 8 h .. 8 h + 7 (frame indices without the fixed frame 0).  Entry (r, c) of the block (row frame i, column frame j):
    //   i <  j   -(w_s (J_i^T J_j)[r][c] + S_dense[r][c])        from the canonical pair (i, j)'s record, descriptor (r, c)
    //   i >  j   the transpose: pair (j, i)'s entry (c, r)        (SolverBundlingDenseUtil.h:349-385; FlipJtJ mirrors the kept triangle)
    //   i == j   w_s (diagonal block of the frame's sparse sums) + the frame's dense S                                  (frame sums)
    // Right-hand side, Jacobi diagonal and p_0 on the lanes behind the row lanes.
    const int wave = (int)(tid >> 6), lane = (int)(tid & 63u);
    const int n_pw = (na + 15) >> 4;                                         // waves that own matrix rows (16 rows each)
    const bool pw = wave < n_pw;
    const int a_row = 16 * wave + (lane >> 2), h = lane & 3;
    const bool row_live = pw && a_row < na;
    f2 Ar[kMidCPL / 2];
    if (pw) {
        const int a_c = min(a_row, na - 1);
        const int i1 = a_c / 6, r = a_c - 6 * i1;
        unsigned t21r[6];
#pragma unroll
        for (int c = 0; c < 6; c++) t21r[c] = (unsigned)tri21(r, c);
        const int4 *lut4 = reinterpret_cast<const int4 *>(lut_l);
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {
            const int jf = 8 * h + jj;
            const bool col_live = jf < N - 1;
            const int jc = min(jf, N - 2);
            float v[6];
            if (jc == i1) {
                const float *Fk = F + __umul24((unsigned)i1, kFrameSums);
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    const int4 d = lut4[6 * r + c];
                    const unsigned q1 = d.x & 255, q2 = d.y;                // indices into (n, s[3], -, M[6]) of endpoint i -> frame sums 0, 1..3, 4..9
                    const unsigned di1 = q1 < 4 ? q1 : q1 - 3, di2 = q2 < 4 ? q2 : q2 - 3;
                    float e = w_s * (__int_as_float(d.z) * Fk[di1] + __int_as_float(d.w) * Fk[di2]);
                    if (Pd) e += Fk[20 + t21r[c]];
                    v[c] = e;
                }
            } else {
                const bool upper = i1 < jc;
                const int lo = upper ? i1 : jc, hi = upper ? jc : i1;       // canonical pair (lo + 1, hi + 1)
                const int p = (lo + 1) * N - ((lo + 1) * (lo + 2)) / 2 + (hi - lo - 1);
                const int dq = cross_l[p];
                const float *rec = ps + __umul24((unsigned)p, kSparseVals);
                const float *sdp = pd + __umul24((unsigned)max(dq, 0), kDenseVals);
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    const int4 d = lut4[36 + (upper ? 6 * r + c : 6 * c + r)];
                    const float m1 = rec[d.x & 255], m2 = rec[d.y];
                    const float sd = Pd ? sdp[t21r[c]] : 0.0f;
                    v[c] = -(w_s * (__int_as_float(d.z) * m1 + __int_as_float(d.w) * m2)) - (dq >= 0 ? sd : 0.0f);
                }
            }
#pragma unroll
            for (int c = 0; c < 6; c += 2) Ar[3 * jj + c / 2] = col_live ? (f2){ v[c], v[c + 1] } : (f2){ 0.f, 0.f };
        }
    }
    {
        const unsigned t0 = 16u * (unsigned)n_pw * 4u;                       // first lane behind the row lanes (a wave boundary: n_pw waves of 64)
        if (tid >= t0 && tid < t0 + (unsigned)na) {
            const unsigned a = tid - t0, k1 = a / 6u, r = a - 6u * k1;
            const float *Fk = F + __umul24(k1, kFrameSums);
            // b = -J^T r: sparse part weighted (SolverBundlingEquationsLie.h:60-137), dense part from the sweep's g;  M^-1 = 1 / diag of the UNWEIGHTED sparse J^T J (Lie.h:107-108)
            const float rhs = w_s * Fk[10 + r] - (Pd ? Fk[41 + r] : 0.0f);
            const float md = Fk[r < 3 ? 16 : 14 + r];
            const float minv = (md > kEps) ? 1.0f / md : 1.0f;
            vb[a] = rhs;
            vM[a] = minv;
            vp[a] = minv * rhs;                                              // p_0 = M^-1 r_0
        }
    }
    __syncthreads();
    BTBA_MSTAMP(3);
    if (tr) {
        // trace order (rot, trans) per frame; internal [trans, rot]; frame 0's entries are zero
        const int n = 6 * N;
        for (int e = (int)tid; e < n; e += nthr) {
            const int k = e / 6, r = e % 6, o = k * 6 + (r < 3 ? r + 3 : r - 3);
            tr[S.tr_rhs + o] = k ? vb[e - 6] : 0.0f;
            tr[S.tr_prec + o] = k ? vM[e - 6] : 0.0f;
        }
        for (int e = (int)tid; e < n * n; e += nthr) { const int row = e / n, col = e - row * n; if (row < 6 || col < 6) tr[S.tr_A + e] = 0.0f; }
        if (row_live) {
#pragma unroll
            for (int k = 0; k < kMidCPL / 2; k++) {
                const int col = kMidCPL * h + 2 * k;
                if (col < na) tr[S.tr_A + (size_t)(a_row + 6) * n + col + 6] = Ar[k].x;
                if (col + 1 < na) tr[S.tr_A + (size_t)(a_row + 6) * n + col + 7] = Ar[k].y;
            }
        }
        for (int e = (int)tid; e < Pd * kDenseVals; e += nthr) tr[S.tr_dpair + e] = pd[e];
    }
    BTBA_MSTAMP(4);

    // ---- phase 3: Jacobi-preconditioned CG (SolverBundling.cu:575-818; the absolute epsilon guards of :746-818 as they stand)
    {
        float r_[3], m_[3], p_[3], d_[3];
        float rz = 0.0f;
        if (wave == 0) {
            float part = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int idx = lane + 64 * j;
                const bool live = idx < na;
                r_[j] = live ? vb[idx] : 0.0f; m_[j] = live ? vM[idx] : 0.0f; d_[j] = 0.0f;
                p_[j] = m_[j] * r_[j];                                      // (= what phase 2c stored in vp)
                part += r_[j] * p_[j];
            }
            rz = wave_sum_all(part);
        }
        for (int li = 0; li < S.n_pcg; li++) {
            if (pw) {
                const float4 *p4 = reinterpret_cast<const float4 *>(vp + h * kMidCPL);
                f2 qa = (f2){ 0.f, 0.f }, qb = qa, qc = qa, qd = qa;
#pragma unroll
                for (int k = 0; k < kMidCPL / 4; k++) {
                    const float4 pc = p4[k];
                    if (k & 1) { qc = __builtin_elementwise_fma(Ar[2 * k], (f2){ pc.x, pc.y }, qc); qd = __builtin_elementwise_fma(Ar[2 * k + 1], (f2){ pc.z, pc.w }, qd); }
                    else { qa = __builtin_elementwise_fma(Ar[2 * k], (f2){ pc.x, pc.y }, qa); qb = __builtin_elementwise_fma(Ar[2 * k + 1], (f2){ pc.z, pc.w }, qb); }
                }
                qa += qc; qb += qd;
                float s = (qa.x + qa.y) + (qb.x + qb.y);
                s = dpp_add<0xB1, 0xf>(s);                                  // + the row's other three lanes: quad_perm [1,0,3,2],
                s = dpp_add<0x4E, 0xf>(s);                                  //   quad_perm [2,3,0,1]
                if (h == 0 && row_live) vAp[a_row] = s;
            }
            __syncthreads();
            if (wave == 0) {
                float ap_[3], z_[3];
                float part = 0.0f;
#pragma unroll
                for (int j = 0; j < 3; j++) { const int idx = lane + 64 * j; ap_[j] = idx < na ? vAp[idx] : 0.0f; part += p_[j] * ap_[j]; }
                const float pAp = wave_sum_all(part);
                const float alpha = (pAp > kEps) ? rz * __builtin_amdgcn_rcpf(pAp) : 0.0f;
                part = 0.0f;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    d_[j] = d_[j] + alpha * p_[j];
                    r_[j] = r_[j] - alpha * ap_[j];
                    z_[j] = m_[j] * r_[j];
                    part += z_[j] * r_[j];
                }
                const float rz_new = wave_sum_all(part);
                const float beta = (rz > kEps) ? rz_new * __builtin_amdgcn_rcpf(rz) : 0.0f;
                if (tr && tid == 0) { float *sc = tr + S.tr_pcg + 4 * li; sc[0] = pAp; sc[1] = alpha; sc[2] = rz_new; sc[3] = beta; }
                rz = rz_new;
#pragma unroll
                for (int j = 0; j < 3; j++) { p_[j] = z_[j] + beta * p_[j]; if (lane + 64 * j < na) vp[lane + 64 * j] = p_[j]; }
            }
            __syncthreads();
        }
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < 3; j++) if (lane + 64 * j < na) vd[lane + 64 * j] = d_[j];
        }
    }
    __syncthreads();
    BTBA_MSTAMP(5);

    // ---- phase 4: x_k <- Log(Exp(delta_k) Exp(x_k)), the next iterate's T (SolverBundling.cu:805-815, 890-897), one lane per frame ...
    if (tid < (unsigned)N) {
        const int k = (int)tid;
        float *xk = S.x + __umul24(b, (unsigned)S.x_stride) + 6 * k;
        const float *xl = x_l + 6 * k;
        float rot[3] = { xl[0], xl[1], xl[2] }, trans[3] = { xl[3], xl[4], xl[5] };
        if (k > 0) {
            const float *dk = vd + 6 * (k - 1);
            const float dW[3] = { dk[3], dk[4], dk[5] }, dT[3] = { dk[0], dk[1], dk[2] };
            const Mat4 U = pose_to_matrix<BTBA_SOLVE_FAST_SE3>(dW, dT);
            Mat4 C = load_mat4(vT + 16 * k);                                // = Exp(x_k), from the previous launch: a pose_to_matrix result,
            C.m[12] = 0.0f; C.m[13] = 0.0f; C.m[14] = 0.0f; C.m[15] = 1.0f;   // whose last row is these constants
            matrix_to_pose<BTBA_SOLVE_FAST_SE3>(mat_mul(U, C), rot, trans);
            xk[0] = rot[0]; xk[1] = rot[1]; xk[2] = rot[2]; xk[3] = trans[0]; xk[4] = trans[1]; xk[5] = trans[2];
        }
        const Mat4 E = pose_to_matrix<BTBA_SOLVE_FAST_SE3>(rot, trans);
        store_mat4(S.T + __umul24(b, (unsigned)S.pose_stride) + 16 * k, E);
        if (S.poses_out) store_mat4(S.poses_out + 16 * (b * N + k), E);      // last iterate: convertPosesToMatricesCU (SBA.cpp:115)
        store_mat4(vE + 16 * k, E);
        if (tr) {
            for (int q = 0; q < 3; q++) { tr[S.tr_x + 6 * k + q] = rot[q]; tr[S.tr_x + 6 * k + 3 + q] = trans[q]; }
            for (int q = 0; q < 16; q++) tr[S.tr_T + 16 * k + q] = E.m[q];
            for (int q = 0; q < 3; q++) { tr[S.tr_delta + 6 * k + q] = k ? vd[6 * (k - 1) + 3 + q] : 0.0f; tr[S.tr_delta + 6 * k + 3 + q] = k ? vd[6 * (k - 1) + q] : 0.0f; }
        }
    }
    BTBA_MSTAMP(6);
    __syncthreads();
    // ... and its generic cofactor inverse (float4x4::getInverse, cuda_SimpleMatrixUtil.h:978-1104) on sixteen lanes per frame, as in k_solve_small
    if (tid < 16u * N) {
        const unsigned f = tid >> 4, e = tid & 15u, R = e >> 2, Cc = e & 3u;
        const float *m = vE + 16 * f;
        const int r0 = (Cc == 0) ? 1 : 0, r1 = (Cc <= 1) ? 2 : 1, r2 = (Cc <= 2) ? 3 : 2;
        const int c0 = (R == 0) ? 1 : 0, c1 = (R <= 1) ? 2 : 1, c2 = (R <= 2) ? 3 : 2;
        const float m00 = m[4 * r0 + c0], m01 = m[4 * r0 + c1], m02 = m[4 * r0 + c2];
        const float m10 = m[4 * r1 + c0], m11 = m[4 * r1 + c1], m12 = m[4 * r1 + c2];
        const float m20 = m[4 * r2 + c0], m21 = m[4 * r2 + c1], m22 = m[4 * r2 + c2];
        const float t1 = m00 * m11 * m22, t2 = m00 * m12 * m21, t3 = m10 * m01 * m22, t4 = m10 * m02 * m21, t5 = m20 * m01 * m12, t6 = m20 * m02 * m11;
        const float even = ((((t1 - t2) - t3) + t4) + t5) - t6;
        const float adj = ((R + Cc) & 1) ? -even : even;
        const int g0 = lane & ~15;
        const float a0 = __shfl(adj, g0, 64), a4 = __shfl(adj, g0 + 4, 64), a8 = __shfl(adj, g0 + 8, 64), a12 = __shfl(adj, g0 + 12, 64);
        const float det = m[0] * a0 + m[1] * a4 + m[2] * a8 + m[3] * a12;
        const float rdet = se3_div<BTBA_SOLVE_FAST_SE3>(1.0f, det);
        S.Tinv[__umul24(b, (unsigned)S.pose_stride) + tid] = adj * rdet;
    }
    BTBA_MSTAMP(7);
#undef BTBA_MSTAMP
}

}  // namespace btba
