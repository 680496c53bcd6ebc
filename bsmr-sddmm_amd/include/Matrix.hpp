#pragma once
// Dense operand container and sparse containers (CSR / COO / BELL) with the
// public interface of the reference's include/Matrix.hpp:40-399, re-implemented.
//
// Differences that matter:
//  * makeData() is deterministic (seeded, drawn sequentially); the reference
//    shares one std::mt19937 between OpenMP threads (src/Matrix.cpp:131-137).
//  * All offsets inside the implementation are size_t; the public index type
//    stays UIN (uint32_t) like the reference.

#include <cstddef>
#include <iostream>
#include <string>
#include <tuple>
#include <vector>

#include "MfmaConfig.hpp"

enum MatrixStorageOrder { row_major, col_major };

enum MatrixMultiplicationOrder { left_multiplication, right_multiplication };

namespace sparseMatrix {
template <typename T> class CSR;
template <typename T> class COO;
}  // namespace sparseMatrix

template <typename T>
class Matrix {
public:
    Matrix() = delete;

    Matrix(UIN row, UIN col, MatrixStorageOrder order)
        : row_(row), col_(col), storageOrder_(order),
          leadingDimension_(order == row_major ? col : row),
          values_(static_cast<size_t>(row) * col) {}

    Matrix(UIN row, UIN col, MatrixStorageOrder order, const std::vector<T>& values)
        : row_(row), col_(col), storageOrder_(order),
          leadingDimension_(order == row_major ? col : row), values_(values) {
        if (static_cast<size_t>(row) * col != values.size())
            std::cout << "Warning! Matrix initialization mismatch" << std::endl;
    }

    Matrix(UIN row, UIN col, MatrixStorageOrder order, const T* values)
        : row_(row), col_(col), storageOrder_(order),
          leadingDimension_(order == row_major ? col : row),
          values_(values, values + static_cast<size_t>(row) * col) {}

    // Densify a COO matrix (row-major).
    explicit Matrix(const sparseMatrix::COO<T>& matrixS);

    bool initializeValue(const std::vector<T>& src);

    void changeStorageOrder();

    UIN rowOfValueIndex(UIN idx) const;
    UIN colOfValueIndex(UIN idx) const;

    T getOneValue(UIN row, UIN col) const;

    // Element used at step `k` of C[rowMtxC, colMtxC] += A[rowMtxC,k] * B[k,colMtxC];
    // `order` says whether *this is the left (A) or right (B) operand.
    T getOneValueForMultiplication(MatrixMultiplicationOrder order,
                                   UIN rowMtxC, UIN colMtxC, UIN k) const;

    // U[0,2) like the reference (src/Matrix.cpp:131-137), but reproducible:
    // x = 2 * (u >> 8) * 2^-24, u drawn sequentially from std::mt19937(seed).
    void makeData();
    void makeData(UIN numRow, UIN numCol);
    void makeDataSeeded(uint32_t seed);

    void print() const;

    std::vector<T> getRowVector(UIN row) const;
    std::vector<T> getColVector(UIN col) const;

    UIN size() const { return static_cast<UIN>(values_.size()); }
    MatrixStorageOrder storageOrder() const { return storageOrder_; }
    UIN leadingDimension() const { return leadingDimension_; }
    UIN row() const { return row_; }
    UIN col() const { return col_; }
    const std::vector<T>& values() const { return values_; }
    std::vector<T>& setValues() { return values_; }
    const T* data() const { return values_.data(); }

    const T& operator[](size_t idx) const { return values_[idx]; }
    T& operator[](size_t idx) { return values_[idx]; }

private:
    UIN row_;
    UIN col_;
    MatrixStorageOrder storageOrder_ = row_major;
    UIN leadingDimension_;
    std::vector<T> values_;
};

template <typename T>
inline std::ostream& operator<<(std::ostream& os, const Matrix<T>& m) {
    return os << " [row : " << m.row() << ", col : " << m.col() << "]";
}

namespace sparseMatrix {

class DataBase {
public:
    DataBase() = default;
    UIN row() const { return row_; }
    UIN col() const { return col_; }
    UIN nnz() const { return nnz_; }
    float getSparsity() const {
        const uint64_t total = static_cast<uint64_t>(row_) * col_;
        return total == 0 ? 0.0f : 1.0f - static_cast<float>(nnz_) / static_cast<float>(total);
    }

protected:
    UIN row_ = 0;
    UIN col_ = 0;
    UIN nnz_ = 0;
};

template <typename T>
class CSR : public DataBase {
public:
    CSR() = default;

    CSR(UIN row, UIN col, UIN nnz, const std::vector<UIN>& rowOffsets,
        const std::vector<UIN>& colIndices, const std::vector<T>& values)
        : rowOffsets_(rowOffsets), colIndices_(colIndices), values_(values) { setDims(row, col, nnz); }

    CSR(UIN row, UIN col, UIN nnz, const UIN* rowOffsets, const UIN* colIndices, const T* values)
        : rowOffsets_(rowOffsets, rowOffsets + row + 1), colIndices_(colIndices, colIndices + nnz),
          values_(values, values + nnz) { setDims(row, col, nnz); }

    CSR(UIN row, UIN col, UIN nnz, const int* rowOffsets, const int* colIndices, const T* values)
        : rowOffsets_(rowOffsets, rowOffsets + row + 1), colIndices_(colIndices, colIndices + nnz),
          values_(values, values + nnz) { setDims(row, col, nnz); }

    CSR(UIN row, UIN col, UIN nnz, const std::vector<UIN>& rowOffsets,
        const std::vector<UIN>& colIndices)
        : rowOffsets_(rowOffsets), colIndices_(colIndices), values_(nnz, T(0)) { setDims(row, col, nnz); }

    // Dispatch on the file suffix: .mtx/.mmio, .smtx (DLMC), .txt (SNAP edge list).
    bool initializeFromMatrixFile(const std::string& file);
    bool initializeFromMtxFile(const std::string& file);
    bool initializeFromSmtxFile(const std::string& file);
    bool initializeFromGraphDataset(const std::string& file);
    bool initializeFromNpzFile(const std::string& file);   // graph archives of scripts/convert_mtx_to_npz.py

    bool outputToMarketMatrixFile(const std::string& fileName) const;
    bool outputToMarketMatrixFile() const;

    const std::vector<UIN>& rowOffsets() const { return rowOffsets_; }
    const std::vector<UIN>& colIndices() const { return colIndices_; }
    const std::vector<T>& values() const { return values_; }
    std::vector<T>& setValues() { return values_; }

private:
    void setDims(UIN r, UIN c, UIN n) { row_ = r; col_ = c; nnz_ = n; }
    std::vector<UIN> rowOffsets_;
    std::vector<UIN> colIndices_;
    std::vector<T> values_;
};

template <typename T>
class COO : public DataBase {
public:
    COO() = default;

    COO(UIN row, UIN col, UIN nnz, const std::vector<UIN>& rowIndices,
        const std::vector<UIN>& colIndices, const std::vector<T>& values)
        : rowIndices_(rowIndices), colIndices_(colIndices), values_(values) {
        row_ = row; col_ = col; nnz_ = nnz;
    }

    explicit COO(const CSR<T>& csr);

    bool initializeFromMatrixMarketFile(const std::string& file);
    bool outputToMarketMatrixFile(const std::string& fileName) const;
    bool outputToMarketMatrixFile() const;

    CSR<T> getCsrData() const;

    const std::vector<UIN>& rowIndices() const { return rowIndices_; }
    const std::vector<UIN>& colIndices() const { return colIndices_; }
    const std::vector<T>& values() const { return values_; }
    std::vector<T>& setValues() { return values_; }

    std::tuple<UIN, UIN, T> getSpareMatrixOneData(UIN idx) const {
        return std::make_tuple(rowIndices_[idx], colIndices_[idx], values_[idx]);
    }
    std::tuple<UIN, UIN, T> operator[](UIN idx) const { return getSpareMatrixOneData(idx); }

private:
    std::vector<UIN> rowIndices_;
    std::vector<UIN> colIndices_;
    std::vector<T> values_;
};

// Blocked-ELL view (reference include/Matrix.hpp:371-397); plain data holder.
template <typename T>
class BELL : public DataBase {
public:
    BELL() = default;
    BELL(UIN row, UIN col, UIN nnz, const std::vector<UIN>& blockRowOffsets,
         const std::vector<UIN>& blockColIndices, const std::vector<T>& blockValues)
        : blockRowOffsets_(blockRowOffsets), blockColIndices_(blockColIndices),
          blockValues_(blockValues) { row_ = row; col_ = col; nnz_ = nnz; }
    const std::vector<UIN>& blockRowOffsets() const { return blockRowOffsets_; }
    const std::vector<UIN>& blockColIndices() const { return blockColIndices_; }
    const std::vector<T>& blockValues() const { return blockValues_; }

private:
    std::vector<UIN> blockRowOffsets_;
    std::vector<UIN> blockColIndices_;
    std::vector<T> blockValues_;
};

}  // namespace sparseMatrix

// rowOffsets from row ids sorted ascending (reference src/Matrix.cpp:236-250).
void getCsrRowOffsets(UIN row, const std::vector<UIN>& rowIndices, std::vector<UIN>& rowOffsets);

// Structural sanity of a CSR matrix: monotone offsets, in-range columns, no
// duplicate (row, col).
template <typename T>
bool checkMatrixData(const sparseMatrix::CSR<T>& csr);
