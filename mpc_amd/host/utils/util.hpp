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

static inline void matrix2string(const MatrixXl &input, string &output) {
    const size_t rows = input.rows(), cols = input.cols();
    output.resize(rows * cols * sizeof(int64_t));
    int64_t *dst = (int64_t *)&output[0];
    for (size_t j = 0; j < cols; j++)
        for (size_t i = 0; i < rows; i++) dst[i * cols + j] = input(i, j);
}
static inline void string2matrix(const string &input, MatrixXl &output) {
    const size_t rows = output.rows(), cols = output.cols();
    assert(input.size() == rows * cols * sizeof(int64_t));
    const int64_t *src = (const int64_t *)input.data();
    for (size_t j = 0; j < cols; j++)
        for (size_t i = 0; i < rows; i++) output(i, j) = src[i * cols + j];
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
    for (size_t j = 0; j < cols; j++)
        for (size_t i = 0; i < rows; i++) output[i * cols + j] = input(i, j);
}
static inline void vector2matrix(const vector<int64_t> &input, MatrixXl &output) {
    const size_t rows = output.rows(), cols = output.cols();
    assert(input.size() == rows * cols);
    for (size_t j = 0; j < cols; j++)
        for (size_t i = 0; i < rows; i++) output(i, j) = input[i * cols + j];
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
    for (size_t j = 0; j < cols; j++)
        for (size_t i = 0; i < rows; i++) output(i, j) = input[i * cols + j];
}
static inline void matrix2vecI64(const MatrixXl &input, VecI64 &output) {
    const size_t rows = input.rows(), cols = input.cols();
    output.resize(rows * cols);
    for (size_t j = 0; j < cols; j++)
        for (size_t i = 0; i < rows; i++) output[i * cols + j] = input(i, j);
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
