// CPU products (reference src/host.cpp:5-124).  Strictly sequential k loop per
// output so the rounding sequence is the reference's; rows are distributed
// over OpenMP threads.
#include <omp.h>
#include "util.hpp"

#include "host.hpp"

#include <iostream>

namespace {
template <typename T>
bool shapesAgree(UIN aRow, UIN aCol, UIN bRow, UIN bCol, UIN pRow, UIN pCol) {
    if (aCol != bRow || aRow != pRow || bCol != pCol) {
        std::cerr << "The storage of the three matrices does not match" << std::endl;
        return false;
    }
    return true;
}

template <typename T>
inline T dotK(const Matrix<T>& A, const Matrix<T>& B, UIN row, UIN col, UIN K) {
    T acc = T(0);
    for (UIN k = 0; k < K; ++k)
        acc += A.getOneValueForMultiplication(left_multiplication, row, col, k) *
               B.getOneValueForMultiplication(right_multiplication, row, col, k);
    return acc;
}
}  // namespace

template <typename T>
void dmm_cpu(const Matrix<T>& A, const Matrix<T>& B, Matrix<T>& C) {
    if (!shapesAgree<T>(A.row(), A.col(), B.row(), B.col(), C.row(), C.col())) return;
    const UIN K = A.col();
    const long long total = static_cast<long long>(C.row()) * C.col();
#pragma omp parallel for num_threads(util::hostThreads(omp_get_max_threads()))
    for (long long i = 0; i < total; ++i)
        C[static_cast<size_t>(i)] = dotK(A, B, C.rowOfValueIndex(static_cast<UIN>(i)),
                                         C.colOfValueIndex(static_cast<UIN>(i)), K);
}

template <typename T>
void sddmm_cpu(const Matrix<T>& A, const Matrix<T>& B, const sparseMatrix::CSR<T>& S,
               sparseMatrix::CSR<T>& P) {
    if (!shapesAgree<T>(A.row(), A.col(), B.row(), B.col(), P.row(), P.col())) return;
    const UIN K = A.col();
    std::vector<T>& out = P.setValues();
#pragma omp parallel for schedule(dynamic, 64) num_threads(util::hostThreads(omp_get_max_threads()))
    for (long long row = 0; row < static_cast<long long>(S.row()); ++row)
        for (UIN e = S.rowOffsets()[row]; e < S.rowOffsets()[row + 1]; ++e)
            out[e] = dotK(A, B, static_cast<UIN>(row), S.colIndices()[e], K);
}

template <typename T>
void sddmm_cpu(const Matrix<T>& A, const Matrix<T>& B, const sparseMatrix::COO<T>& S,
               sparseMatrix::COO<T>& P) {
    if (!shapesAgree<T>(A.row(), A.col(), B.row(), B.col(), P.row(), P.col())) return;
    const UIN K = A.col();
    std::vector<T>& out = P.setValues();
#pragma omp parallel for num_threads(util::hostThreads(omp_get_max_threads()))
    for (long long e = 0; e < static_cast<long long>(S.nnz()); ++e)
        out[e] = dotK(A, B, S.rowIndices()[e], S.colIndices()[e], K);
}

#define BSMR_INSTANTIATE(T)                                                                    \
    template void dmm_cpu<T>(const Matrix<T>&, const Matrix<T>&, Matrix<T>&);                  \
    template void sddmm_cpu<T>(const Matrix<T>&, const Matrix<T>&, const sparseMatrix::CSR<T>&, \
                               sparseMatrix::CSR<T>&);                                         \
    template void sddmm_cpu<T>(const Matrix<T>&, const Matrix<T>&, const sparseMatrix::COO<T>&, \
                               sparseMatrix::COO<T>&);
BSMR_INSTANTIATE(int)
BSMR_INSTANTIATE(float)
BSMR_INSTANTIATE(double)
