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
// in 32 x 32 tiles so that neither side is walked with a large stride.
static inline void cm_to_rm(const int64_t *cm, int64_t *rm, size_t rows, size_t cols) {
    const size_t T = 32;
    for (size_t j0 = 0; j0 < cols; j0 += T)
        for (size_t i0 = 0; i0 < rows; i0 += T) {
            const size_t j1 = j0 + T < cols ? j0 + T : cols, i1 = i0 + T < rows ? i0 + T : rows;
            for (size_t j = j0; j < j1; j++)
                for (size_t i = i0; i < i1; i++) rm[i * cols + j] = cm[j * rows + i];
        }
}
static inline void rm_to_cm(const int64_t *rm, int64_t *cm, size_t rows, size_t cols) {
    const size_t T = 32;
    for (size_t j0 = 0; j0 < cols; j0 += T)
        for (size_t i0 = 0; i0 < rows; i0 += T) {
            const size_t j1 = j0 + T < cols ? j0 + T : cols, i1 = i0 + T < rows ? i0 + T : rows;
            for (size_t j = j0; j < j1; j++)
                for (size_t i = i0; i < i1; i++) cm[j * rows + i] = rm[i * cols + j];
        }
}
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
    if (!input.empty()) memcpy(output.data(), input.data(), input.size());
}
static inline void vecZpshare2string(const vector<zpshare> &input, string &output) {
    output.assign((const char *)input.data(), input.size() * sizeof(zpshare));
}
static inline void string2vecZpshare(const string &input, vector<zpshare> &output) {
    assert(input.size() % sizeof(zpshare) == 0);
    output.resize(input.size() / sizeof(zpshare));
    if (!input.empty()) memcpy(output.data(), input.data(), input.size());
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
    output.assign((const char *)input.data(), input.size() * sizeof(int64_t));
}
static inline void string2vecI64(const string &input, VecI64 &output) {
    assert(input.size() % sizeof(int64_t) == 0);
    output.resize(input.size() / sizeof(int64_t));
    if (!input.empty()) memcpy(output.data(), input.data(), input.size());
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
