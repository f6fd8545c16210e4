// poseidon_precomp.h — host-side precomputation for the grouped partial rounds of
// hash_kernels.cuh (glp_partial_group).  Pure C++ (shared with tests/emu).
#pragma once
#include <vector>
#include "hash_kernels.cuh"

// consts384 = rc[360] | circ[12] | diag[12].  On success fills coef (GLP_PG_GROUPS * GLP_PG_COEF)
// and cst (GLP_PG_GROUPS * GLP_PG_CST) and returns true; returns false when the integer
// coefficients would exceed the bounds the fast dot product relies on (then the kernels run the
// plain partial rounds).
static inline bool glp_poseidon_group_tables(const u64* consts384, std::vector<u32>& coef, std::vector<u64>& cst) {
    const u64* rc = consts384; const u64* circ = consts384 + 360; const u64* diag = consts384 + 372;
    unsigned long long M[12][12];
    for (int r = 0; r < 12; r++)
        for (int c = 0; c < 12; c++) {
            if (circ[(c - r + 12) % 12] >> 24 || diag[r] >> 24) return false;
            M[r][c] = circ[(c - r + 12) % 12] + (r == c ? diag[r] : 0);
        }
    coef.assign((size_t)GLP_PG_GROUPS * GLP_PG_COEF, 0);
    cst.assign((size_t)GLP_PG_GROUPS * GLP_PG_CST, 0);
    for (int g = 0; g < GLP_PG_GROUPS; g++) {
        const int r0 = GLP_POS_FULL_HALF + GLP_PG_K * g;           // absolute round index of the group's first round
        // state before round r0+j:  A[j] * that(r0) + sum_i f_i * B[j][i] + D[j]
        unsigned long long A[GLP_PG_K + 1][12][11] = {};
        unsigned long long B[GLP_PG_K + 1][GLP_PG_K][12] = {};
        u64 D[GLP_PG_K + 1][12] = {};
        for (int q = 1; q < 12; q++) A[0][q][q - 1] = 1;
        for (int j = 0; j < GLP_PG_K; j++) {
            for (int r = 0; r < 12; r++) {
                for (int c = 0; c < 11; c++) {
                    unsigned long long acc = 0;
                    for (int q = 1; q < 12; q++) acc += M[r][q] * A[j][q][c];
                    if (acc >> 24) return false;
                    A[j + 1][r][c] = acc;
                }
                for (int i = 0; i < j; i++) {
                    unsigned long long acc = 0;
                    for (int q = 1; q < 12; q++) acc += M[r][q] * B[j][i][q];
                    if (acc >> 24) return false;
                    B[j + 1][i][r] = acc;
                }
                B[j + 1][j][r] = M[r][0];
                u64 d = rc[(r0 + j + 1) * 12 + r];
                for (int q = 1; q < 12; q++) d = gl_add(d, gl_mul(M[r][q] % GL_P, D[j][q]));
                D[j + 1][r] = d;
            }
        }
        // row sums bound (accumulators stay < 2^58)
        for (int j = 1; j <= GLP_PG_K; j++)
            for (int r = 0; r < 12; r++) {
                unsigned long long sum = 0;
                for (int c = 0; c < 11; c++) sum += A[j][r][c];
                for (int i = 0; i < j; i++) sum += B[j][i][r];
                if (sum >> 26) return false;
            }
        u32* cf = coef.data() + (size_t)g * GLP_PG_COEF;
        u64* cs = cst.data() + (size_t)g * GLP_PG_CST;
        // x_1: 11 + 1, x_2: 11 + 2, end: 12 rows of 11 + 3
        for (int c = 0; c < 11; c++) cf[c] = (u32)A[1][0][c];
        cf[11] = (u32)B[1][0][0];
        cs[0] = D[1][0];
        for (int c = 0; c < 11; c++) cf[12 + c] = (u32)A[2][0][c];
        cf[12 + 11] = (u32)B[2][0][0]; cf[12 + 12] = (u32)B[2][1][0];
        cs[1] = D[2][0];
        for (int r = 0; r < 12; r++) {
            u32* row = cf + 25 + 14 * r;
            for (int c = 0; c < 11; c++) row[c] = (u32)A[3][r][c];
            for (int i = 0; i < 3; i++) row[11 + i] = (u32)B[3][i][r];
            cs[2 + r] = D[3][r];
        }
    }
    return true;
}
