// From-scratch rebuild of one slot's derived state (reference gaussian_components.py:319-331:
// slogdet + inv of the predictive covariance), blocked for the matrix pipe:
//     S_N = S - k_N mu mu'  =  L L',   Winv = L^-1,   logdet S_N = sum log of the pivots
// in 16 x 16 blocks.  Per block column J: the diagonal block is factored (L D L', no square root on
// the pivot chain) and inverted in REGISTERS by one wavefront, lane = row, the pivot row broadcast
// with v_readlane; the panel below it and the trailing matrix are v_mfma_f64_16x16x4_f64 products
// shared by the block's four wavefronts.  The inverse of the whole factor is then built block
// column by block column, each wavefront on its own columns with its partial results staying in
// registers: a 16 x 16 block in the accumulator layout (lane (lk, lr), register r <-> row lk + 4 r,
// column lr) IS the B operand of the next product (k = 4 kk + lk <-> register kk), so no data moves
// between the products of a chain.
// 256 threads.  LDS: A[Dp][Dp + 2] (row stride = 2 mod 32 doubles: fragment reads are conflict free),
// Wd[Dp / 16][16][18] inverse diagonal blocks, mu[Dp], row[Dp], 8 scalars.
#pragma once
#include "slot_math.h"
#include "fast_math.h"
#include "wave_ops.h"

typedef double rb_v4d __attribute__((ext_vector_type(4)));

__host__ __device__ inline int refresh_blocked_lds_doubles(int Dp) {
    return Dp * (Dp + 2) + (Dp / 16) * 16 * 18 + 2 * Dp + 8;
}

// acc (+)= sign * Ablk * Bblk' with both 16 x 16 blocks read row-wise from LDS:
//   A operand: lane (lr, lk) <- Ab[lr * lda + 4 kk + lk];  B[k][j] = Bb[j * ldb + k] <- Bb[lr * ldb + 4 kk + lk]
__device__ __forceinline__ rb_v4d rb_mma_abt(rb_v4d acc, const double *Ab, int lda, const double *Bb, int ldb,
                                             int lr, int lk, bool negate) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double a = Ab[lr * lda + 4 * kk + lk];
        const double b = Bb[lr * ldb + 4 * kk + lk];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(negate ? -a : a, b, acc, 0, 0, 0);
    }
    return acc;
}

// acc += Ablk * B with A read row-wise from LDS and B held in the accumulator layout (register kk = rows 4 kk + lk)
__device__ __forceinline__ rb_v4d rb_mma_areg(rb_v4d acc, const double *Ab, int lda, rb_v4d breg, int lr, int lk,
                                              bool negate) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double a = Ab[lr * lda + 4 * kk + lk];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(negate ? -a : a, breg[kk], acc, 0, 0, 0);
    }
    return acc;
}

// Diagonal block J (one wavefront; lanes 0 .. 15 = rows): unit-lower L and pivots d of A_JJ = L D L',
// then Wd[J] = D^-1/2 L^-1 (the inverse Cholesky factor of the block).  Returns the sum of log d
// (every lane) and raises *bad on a pivot that is not positive.
__device__ __forceinline__ double rb_diag_block(double *A, int LD, double *WdJ, int J, int lane, bool &bad) {
    const int i = lane & 15;
    double a[16];
    const double *Ar = A + (16 * J + i) * LD + 16 * J;
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = (k <= i && lane < 16) ? Ar[k] : 0.0;
    double dmine = 1.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double pj = wv_readlane(a[j], j);
        bad = bad || !(pj > 0.0);
        dmine = i == j ? pj : dmine;
        const double lcol = a[j] * fm_div(1.0, pj);
#pragma unroll
        for (int k = j + 1; k < 16; ++k) {
            const double akj = wv_readlane(a[j], k);          // a_kj before the scaling = l_kj d_j
            a[k] = fma(-lcol, akj, a[k]);
        }
        a[j] = lcol;
    }
    // X = L^-1 (unit lower), lane = column c
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        double sacc = r == i ? 1.0 : 0.0;
#pragma unroll
        for (int t = 0; t < r; ++t) {
            const double lrt = wv_readlane(a[t], r);
            sacc = fma(-lrt, x[t], sacc);
        }
        x[r] = sacc;
    }
    const double rs = fm_div(1.0, sqrt(dmine));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const double rsr = wv_readlane(rs, r);
        if (lane < 16) WdJ[r * 18 + i] = r >= i ? x[r] * rsr : 0.0;
    }
    double lg = lane < 16 ? log(dmine) : 0.0;
    lg += wv_dpp<0xB1>(lg);
    lg += wv_dpp<0x4E>(lg);
    lg += wv_dpp<0x141>(lg);
    lg += wv_dpp<0x140>(lg);
    return wv_readlane(lg, 0);
}

// sm: refresh_blocked_lds_doubles(Dp) doubles.  All 256 threads of the block call it.
__device__ inline void refresh_slot_blocked(const Dev &d, int s, double *sm) {
    const int D = d.D, Dp = d.Dp, nJ = Dp >> 4, LD = Dp + 2;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lk = lane >> 4;
    double *A = sm;
    double *Wd = A + Dp * LD;
    double *mu = Wd + nJ * 288, *row = mu + Dp;
    double *scal = row + Dp;                       // [0] logdet, [1] bad flag, [2] Gershgorin bound
#ifdef BGMM_PROFILE
    long long rb_t0 = clock64(), rb_t1;
#define RBPROF(i) do { if (tid == 0 && blockIdx.x == 0) { rb_t1 = clock64(); d.ctrl->prof[i] += rb_t1 - rb_t0; rb_t0 = rb_t1; } } while (0)
#else
#define RBPROF(i) do { } while (0)
#endif
    const double k_N = d.k0 + (double)d.n[s];
    const double *m = d.m + (long long)s * D;
    const double *S = d.S + (long long)s * D * D;
    for (int a = tid; a < Dp; a += 256) mu[a] = a < D ? m[a] / k_N : 0.0;
    __syncthreads();
    for (int e = tid; e < Dp * Dp; e += 256) {
        const int a = e / Dp, b = e - a * Dp;
        double v = 0.0;
        if (a < D && b <= a) v = S[a * D + b] - k_N * (mu[a] * mu[b]);
        else if (a >= D && a == b) v = 1.0;           // padding: identity (pivot 1, log 0)
        A[a * LD + b] = v;
    }
    __syncthreads();
    RBPROF(10);
    gershgorin_bound<256>(A, LD, D, row, &scal[2], tid, true);
    RBPROF(11);
    double logdet = 0.0;
    bool bad = false;
    for (int J = 0; J < nJ; ++J) {
        __syncthreads();
        if (w == 0) logdet += rb_diag_block(A, LD, Wd + J * 288, J, lane, bad);
        __syncthreads();
        // panel: A_IJ <- A_IJ W_JJ'   (I > J)
        for (int I = J + 1 + w; I < nJ; I += 4) {
            rb_v4d acc = (rb_v4d){0.0, 0.0, 0.0, 0.0};
            acc = rb_mma_abt(acc, A + (16 * I) * LD + 16 * J, LD, Wd + J * 288, 18, lr, lk, false);
#pragma unroll
            for (int r = 0; r < 4; ++r) A[(16 * I + lk + 4 * r) * LD + 16 * J + lr] = acc[r];
        }
        __syncthreads();
        // trailing matrix: A_IK -= A_IJ A_KJ'   (J < K <= I)
        int q = 0;
        for (int I = J + 1; I < nJ; ++I)
            for (int K = J + 1; K <= I; ++K, ++q) {
                if ((q & 3) != w) continue;
                rb_v4d acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = A[(16 * I + lk + 4 * r) * LD + 16 * K + lr];
                acc = rb_mma_abt(acc, A + (16 * I) * LD + 16 * J, LD, A + (16 * K) * LD + 16 * J, LD, lr, lk, true);
#pragma unroll
                for (int r = 0; r < 4; ++r) A[(16 * I + lk + 4 * r) * LD + 16 * K + lr] = acc[r];
            }
    }
    __syncthreads();
    RBPROF(12);
    if (tid == 0) {
        scal[0] = logdet;
        *(int *)&scal[1] = (bad || !(logdet == logdet)) ? 1 : 0;
    }
    // inverse of the factor, block column by block column: W_JJ = Wd[J];
    // W_IJ = -W_II (sum_{K = J}^{I-1} L_IK W_KJ), the W_KJ of the chain in registers.
    // Wavefront w owns block columns w and 7 - w (nJ <= 8).
    rb_v4d col[2][8];                                    // [pass][K - J]: W_KJ of this wavefront's two block columns
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int J = pass == 0 ? w : 7 - w;
#pragma unroll
        for (int di = 0; di < 8; ++di) col[pass][di] = (rb_v4d){0.0, 0.0, 0.0, 0.0};
        if (J < nJ) {
#pragma unroll
            for (int r = 0; r < 4; ++r) col[pass][0][r] = Wd[J * 288 + (lk + 4 * r) * 18 + lr];
#pragma unroll
            for (int di = 1; di < 8; ++di) {
                const int I = J + di;
                if (I < nJ) {
                    rb_v4d sacc = (rb_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int dk = 0; dk < 8; ++dk)
                        if (dk < di) sacc = rb_mma_areg(sacc, A + (16 * I) * LD + 16 * (J + dk), LD, col[pass][dk], lr, lk, false);
                    rb_v4d wij = (rb_v4d){0.0, 0.0, 0.0, 0.0};
                    wij = rb_mma_areg(wij, Wd + I * 288, 18, sacc, lr, lk, true);
                    col[pass][di] = wij;
                }
            }
        }
    }
    __syncthreads();                                     // every L block has been read
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int J = pass == 0 ? w : 7 - w;
#pragma unroll
        for (int di = 0; di < 8; ++di) {
            const int I = J + di;
            if (J < nJ && I < nJ) {
#pragma unroll
                for (int r = 0; r < 4; ++r) A[(16 * I + lk + 4 * r) * LD + 16 * J + lr] = col[pass][di][r];
            }
        }
    }
    __syncthreads();
    RBPROF(13);
    if (tid == 0 && *(int *)&scal[1]) atomicCAS(&d.ctrl->error, 0, -4);
    write_slot_wide(d, s, A, LD, mu, scal[0], scal[2], tid);
    if (tid == 0) d.nupd[s] = 0;
    RBPROF(14);
}
