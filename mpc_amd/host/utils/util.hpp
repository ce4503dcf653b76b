// Types, constants and wire codecs of the engines -- same names and byte layouts as the reference's
// utils/util.hpp:7-200 (raw little-endian words; matrices row-major; Z_67 bit shares as 64 uint16 per
// element; booleans one byte each; no headers).
#pragma once
#include "../eigen-3.4.0/Eigen/Dense"
#include <array>
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
using MatrixXl = Eigen::Matrix<int64_t, Eigen::Dynamic, Eigen::Dynamic>;
using VecI64 = Eigen::Array<int64_t, Eigen::Dynamic, 1>;
using zpshare = std::array<uint16_t, 64>;
using vecbool = std::vector<int8_t>;
static const uint16_t P = 67;
static const uint64_t L_1 = -1;
using std::string;
using std::vector;

// The wire order is row-major, the matrices are stored column-major: both directions are a transposition, done
// in 32 x 32 tiles so that neither side is walked with a large stride.  Large matrices go to the GPU library's
// host threads when the program is linked with it (a 4096 x 4096 matrix takes one core 70-100 ms).
extern "C" int mpcx_host_transpose64(void *dst, const void *src, uint64_t rows, uint64_t cols) __attribute__((weak));
// dst (C x R, row-major) = transpose of src (R x C, row-major)
static inline void transpose_words(const int64_t *src, int64_t *dst, size_t R, size_t C) {
    if (mpcx_host_transpose64 && R * C >= (size_t)1 << 16 && mpcx_host_transpose64(dst, src, R, C) == 0) return;
    const size_t T = 32;
    for (size_t c0 = 0; c0 < C; c0 += T)
        for (size_t r0 = 0; r0 < R; r0 += T) {
            const size_t c1 = c0 + T < C ? c0 + T : C, r1 = r0 + T < R ? r0 + T : R;
            for (size_t c = c0; c < c1; c++)
                for (size_t r = r0; r < r1; r++) dst[c * R + r] = src[r * C + c];
        }
}
// payload copies: the library's multi-threaded memcpy from 4 MiB on when the program is linked with it
extern "C" int mpcx_host_copy(void *dst, const void *src, uint64_t bytes) __attribute__((weak));
static inline void copy_bytes(void *dst, const void *src, size_t bytes) {
    if (!bytes) return;
    if (mpcx_host_copy && bytes >= ((size_t)4 << 20) && mpcx_host_copy(dst, src, bytes) == 0) return;
    memcpy(dst, src, bytes);
}
// a column-major rows x cols matrix is the row-major cols x rows matrix of its transpose
static inline void cm_to_rm(const int64_t *cm, int64_t *rm, size_t rows, size_t cols) { transpose_words(cm, rm, cols, rows); }
static inline void rm_to_cm(const int64_t *rm, int64_t *cm, size_t rows, size_t cols) { transpose_words(rm, cm, rows, cols); }
static inline void matrix2string(const MatrixXl &input, string &output) {
    const size_t rows = input.rows(), cols = input.cols();
    output.resize(rows * cols * sizeof(int64_t));
    if (rows && cols) cm_to_rm(input.data(), (int64_t *)&output[0], rows, cols);
}
static inline void string2matrix(const string &input, MatrixXl &output) {
    const size_t rows = output.rows(), cols = output.cols();
    assert(input.size() == rows * cols * sizeof(int64_t));
    if (rows && cols) rm_to_cm((const int64_t *)input.data(), output.data(), rows, cols);
}
static inline void vector2string(const vector<int64_t> &input, string &output) {
    output.assign((const char *)input.data(), input.size() * sizeof(int64_t));
}
static inline void string2vector(const string &input, vector<int64_t> &output) {
    assert(input.size() % sizeof(int64_t) == 0);
    output.resize(input.size() / sizeof(int64_t));
    copy_bytes(output.data(), input.data(), input.size());
}
static inline void vecZpshare2string(const vector<zpshare> &input, string &output) {
    output.assign((const char *)input.data(), input.size() * sizeof(zpshare));
}
static inline void string2vecZpshare(const string &input, vector<zpshare> &output) {
    assert(input.size() % sizeof(zpshare) == 0);
    output.resize(input.size() / sizeof(zpshare));
    copy_bytes(output.data(), input.data(), input.size());
}
static inline void matrix2vector(const MatrixXl &input, vector<int64_t> &output) {
    const size_t rows = input.rows(), cols = input.cols();
    output.resize(rows * cols);
    if (rows && cols) cm_to_rm(input.data(), output.data(), rows, cols);
}
static inline void vector2matrix(const vector<int64_t> &input, MatrixXl &output) {
    const size_t rows = output.rows(), cols = output.cols();
    assert(input.size() == rows * cols);
    if (rows && cols) rm_to_cm(input.data(), output.data(), rows, cols);
}
static inline void vecI642string(const VecI64 &input, string &output) {
    output.resize(input.size() * sizeof(int64_t));
    if (input.size()) copy_bytes(&output[0], input.data(), output.size());
}
static inline void string2vecI64(const string &input, VecI64 &output) {
    assert(input.size() % sizeof(int64_t) == 0);
    output.resize(input.size() / sizeof(int64_t));
    copy_bytes(output.data(), input.data(), input.size());
}
static inline void vecI642matrix(const VecI64 &input, MatrixXl &output) {
    const size_t rows = output.rows(), cols = output.cols();
    assert((size_t)input.size() == rows * cols);
    if (rows && cols) rm_to_cm(input.data(), output.data(), rows, cols);
}
static inline void matrix2vecI64(const MatrixXl &input, VecI64 &output) {
    const size_t rows = input.rows(), cols = input.cols();
    output.resize(rows * cols);
    if (rows && cols) cm_to_rm(input.data(), output.data(), rows, cols);
}
static inline void vecbool2string(const vector<int8_t> &input, string &output) {
    output.assign((const char *)input.data(), input.size());
}
static inline void string2vecbool(const string &input, vector<int8_t> &output) {
    output.assign((const int8_t *)input.data(), (const int8_t *)input.data() + input.size());
}

// debug printers
static inline void print_vec_int64(const vector<int64_t> &input) {
    for (int64_t v : input) printf("%016lX ", (unsigned long)v);
    printf("\n");
}
static inline void printVecI64(const VecI64 &input) {
    for (size_t i = 0; i < (size_t)input.size(); i++) printf("%016lX ", (unsigned long)input[i]);
    printf("\n");
}
static inline void printVecbool(const vecbool &input) {
    for (int8_t v : input) printf("%d ", v);
    printf("\n");
}
static inline void print_vec_zpshare(const vector<zpshare> &input) {
    for (const zpshare &z : input) {
        for (size_t k = 0; k < 64; k++) printf("%2d%s", z[k], k % 32 == 31 ? "\n" : " ");
        printf("\n");
    }
}
static inline void print_plus() { printf("++++++++++++++++++++++++++++++++++++++++++++++++++\n"); }
static inline void print_minus() { printf("--------------------------------------------------\n"); }
