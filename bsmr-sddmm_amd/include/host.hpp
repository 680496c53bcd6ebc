#pragma once
// Host (OpenMP) products used as the acceptance oracle and CPU baseline of the
// engine itself: interface of the reference's include/host.hpp:7-23.

#include "Matrix.hpp"

template <typename T>
void dmm_cpu(const Matrix<T>& matrixA, const Matrix<T>& matrixB, Matrix<T>& matrixC);

// P[e] = sum_k A[row(e),k] * B[k,col(e)] for every stored entry e of S; S's
// values are not multiplied in (reference src/host.cpp:62-73).
template <typename T>
void sddmm_cpu(const Matrix<T>& matrixA, const Matrix<T>& matrixB,
               const sparseMatrix::CSR<T>& matrixS, sparseMatrix::CSR<T>& matrixP);

template <typename T>
void sddmm_cpu(const Matrix<T>& matrixA, const Matrix<T>& matrixB,
               const sparseMatrix::COO<T>& matrixS, sparseMatrix::COO<T>& matrixP);
